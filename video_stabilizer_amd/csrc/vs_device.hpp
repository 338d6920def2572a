// vs_device.hpp -- device-side building blocks shared by the gfx950 kernels.
//
// Numerics contract (DESIGN.md "Numerics"): this translation unit is compiled with
// -ffp-contract=off, so every fp32/fp64 expression below rounds exactly as written -- the
// evaluation order is the one in the reference's generators.cpp (cited per function).
// Where a fused multiply-add is wanted for speed it is spelled __builtin_fmaf explicitly.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

// ---- debug build with bounds-checked LDS / scratch indexing (-DVS_DEBUG_BOUNDS, tools/build_variant.sh bounds) ----------------
// GPU AddressSanitizer is not available on this pool, so the index arithmetic of the LDS-tiled kernels has no other net.  In the
// debug build the arrays the kernels carve out of LDS (and the per-pair scratch in global memory) are wrapped in vsd::Span:
// every operator[] checks its index against the extent the carving code derived, records the FIRST violation of the
// translation unit in a device word (site id, index, limit, workgroup, thread) and performs the access on element 0 instead --
// a violation is reported, never executed: nothing faults, the kernel finishes, and vs_debug_bounds_check() (include/vs_amd.h)
// hands the record to the host.  (No trap: on this pool a trapping wave can take the whole node down.)  In the regular build
// VS_SPAN declares the plain restrict pointer the code had before, so the shipped kernels are unchanged instruction for instruction.
// Site ids: 1xx vs_align_kernels.inc (selection, gather, exchange, staging), 2xx vs_warp.hip, 3xx vs_phase.hip.
#ifdef VS_DEBUG_BOUNDS
namespace vsd {
static __device__ unsigned int g_bounds_rec[8];      // [0] violations, [1] site, [2] index (low 32 bits), [3] limit, [4] blockIdx.x, [5] threadIdx.x
__device__ __forceinline__ bool bounds_ok(long long i, long long n, int site) {
    if (i >= 0 && i < n) return true;
    if (atomicAdd(&g_bounds_rec[0], 1u) == 0u) {
        g_bounds_rec[1] = (unsigned)site; g_bounds_rec[2] = (unsigned)i; g_bounds_rec[3] = (unsigned)n;
        g_bounds_rec[4] = blockIdx.x; g_bounds_rec[5] = threadIdx.x;
    }
    return false;
}
template <typename P>
struct Span {
    P p; long long n; int site;
    __device__ __forceinline__ decltype(auto) operator[](long long i) const { return p[bounds_ok(i, n, site) ? i : 0]; }
    __device__ __forceinline__ P raw() const { return p; }
};
}  // namespace vsd
#define VS_SPAN(ptr_type, name, expr, extent, site) const vsd::Span<ptr_type> name{(ptr_type)(expr), (long long)(extent), (site)}
#define VS_SPAN_RAW(name) ((name).raw())
#define VS_ARR(ptr_type) const vsd::Span<ptr_type>&          /* an array parameter */
#define VS_BOUNDS_CHECK(i, extent, site) ((void)vsd::bounds_ok((long long)(i), (long long)(extent), (site)))
#define VS_DEBUG_CLAMP(i, extent) (min(max((i), 0), (extent) - 1))      /* a reported index is not used as it stands */
/* a byte offset that must lie in [0, extent): checked, reported under `site`, clamped */
#define VS_DEBUG_CLAMP_BYTES(off, extent, site) (vsd::bounds_ok((off), (extent), (site)) ? (off) : 0)
// the host-side reader of this translation unit's record (defined once per .hip file: the record is per translation unit)
#define VS_BOUNDS_TU(fn)                                                                                                \
    extern "C" int fn(unsigned out[8], int reset) {                                                                     \
        if (hipMemcpyFromSymbol(out, HIP_SYMBOL(vsd::g_bounds_rec), sizeof(unsigned) * 8) != hipSuccess) return -1;      \
        if (reset) { const unsigned z[8] = {0, 0, 0, 0, 0, 0, 0, 0}; (void)hipMemcpyToSymbol(HIP_SYMBOL(vsd::g_bounds_rec), z, sizeof(z)); } \
        return 0;                                                                                                       \
    }
#else
#define VS_SPAN(ptr_type, name, expr, extent, site) ptr_type __restrict__ name = (ptr_type)(expr)
#define VS_SPAN_RAW(name) (name)
#define VS_ARR(ptr_type) ptr_type __restrict__
#define VS_BOUNDS_CHECK(i, extent, site) ((void)0)
#define VS_DEBUG_CLAMP(i, extent) (i)
#define VS_DEBUG_CLAMP_BYTES(off, extent, site) (off)
#define VS_BOUNDS_TU(fn)
#endif

namespace vsd {

constexpr int kWave = 64;

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return min(max(v, lo), hi); }
// floor(position) -> integer pixel index for clamp-to-edge (or constant-border) sampling of an image n pixels wide / high.  v_cvt_i32_f32
// saturates (NaN -> 0), but the window arithmetic that follows (ix - 1, ix + 2, ...) would then overflow: undefined behaviour, which the compiler
// had turned into a row index of -1 -- a diverged Gauss-Newton run (|T| ~ 1e14) read the 344 bytes in FRONT of its image (found by the random
// sweeps + allocation poisoning, round 4).  An index 4 or more outside the image puts every tap of a 4- or 5-tap window on the same border
// pixel (or outside, for the constant border), so the index is pulled into [-8, n + 8) first: the same taps for every position, no overflow.
// The clamp happens in the float domain, BEFORE the conversion (a float -> int conversion of a value outside the int range, of an infinity or of
// a NaN is undefined in C++ whatever the hardware instruction does): fl is a floor() result, i.e. an integer, and n + 7 < 2^24 is exact in fp32, so
// for every finite fl this is clampi(int(fl), -8, n + 7); NaN -> 0 as in the oracle's sat_int().
__device__ __forceinline__ int sample_index(float fl, int n) { return (fl == fl) ? (int)fminf(fmaxf(fl, -8.0f), (float)(n + 7)) : 0; }

// generators.cpp:31-47: 6th-order even polynomial, Horner in x*x, zero outside |x| < 2.
__device__ __forceinline__ float lanczos2(float x) {
    float x2 = x * x;
    float v = 0.000858519f;
    v = -0.0158853f + v * x2;
    v = 0.128693f + v * x2;
    v = -0.583468f + v * x2;
    v = 1.52229f + v * x2;
    v = -2.05238f + v * x2;
    v = 0.999861f + v * x2;
    return fabsf(x) >= 2.0f ? 0.0f : v;
}

// The four live 1-D weights of the 5-tap window (generators.cpp:482-484 / 684-685).
// Tap u = 0 sits at distance -2 - frac <= -2, so its weight is exactly 0 for every frac in
// [0,1]; adding 0*v to the accumulators does not change them, so it is dropped.  Taps 1..4 are
// kept (tap 4 is 0 only when frac == 0, tap 1 only when frac rounds to 1).
__device__ __forceinline__ void lanczos_weights4(float frac, float w[4]) {
    w[0] = lanczos2(-1.0f - frac);
    w[1] = lanczos2(0.0f - frac);
    w[2] = lanczos2(1.0f - frac);
    w[3] = lanczos2(2.0f - frac);
}

// ---- VS_WARP_LANCZOS2_FAST: the contracted form of the sampler ------------------------------------------------------
// The reference builds its generators for a target with FMA (CMakeLists.txt:151 "...-fma-...") and without strict_float,
// so on the reference's own machine LLVM may fuse a multiply into the add that consumes it.  This mode is the sampler of
// generators.cpp:31-47 / :684-697 with exactly those fusions, same tap order, same accumulators, same sampling position:
//   w(t)   : "c + val * x2" -> fma(val, x2, c), six times (x2 = t * t and the |t| >= 2 select unchanged)
//   w2d    = wx[rx] * wy[ry]                        a rounded product (generators.cpp:687)
//   num    = fma(w2d, val, num)                     rx inner, ry outer, from 0 (:694)
//   den    = den + w2d                              (:695: nothing of its own to fuse)
//   out    = num / den                              one IEEE divide (:697)
// Its CPU twin is the checker's VSO_WARP_LANCZOS2_CONTRACTED (oracle/, std::fmaf), pinned by a literal restatement with an
// exact rational fma (tests/test_oracle_known_answers.py); every kernel that offers the mode is bit-identical to it
// (tests/test_warp_fast_gpu.py: np.array_equal on float and integer outputs).
__device__ __forceinline__ float lanczos2_fma(float x) {
    const float x2 = x * x;
    float v = 0.000858519f;
    v = __builtin_fmaf(v, x2, -0.0158853f);
    v = __builtin_fmaf(v, x2, 0.128693f);
    v = __builtin_fmaf(v, x2, -0.583468f);
    v = __builtin_fmaf(v, x2, 1.52229f);
    v = __builtin_fmaf(v, x2, -2.05238f);
    v = __builtin_fmaf(v, x2, 0.999861f);
    return fabsf(x) >= 2.0f ? 0.0f : v;
}
// taps 1..4 of the 5-tap window: tap 0 has weight exactly 0 (|-2 - frac| >= 2), and fma(+-0, val, num) == num, den + +-0 == den
__device__ __forceinline__ void lanczos_weights4_fma(float frac, float w[4]) {
    w[0] = lanczos2_fma(-1.0f - frac);
    w[1] = lanczos2_fma(0.0f - frac);
    w[2] = lanczos2_fma(1.0f - frac);
    w[3] = lanczos2_fma(2.0f - frac);
}
// one channel: v[ry][rx] = the live 4x4 window as floats
__device__ __forceinline__ float lanczos_contracted_combine(const float v[4][4], const float wx[4], const float wy[4]) {
    float num = 0.0f, den = 0.0f;
#pragma unroll
    for (int ry = 0; ry < 4; ry++)
#pragma unroll
        for (int rx = 0; rx < 4; rx++) {
            const float w2d = wx[rx] * wy[ry];
            num = __builtin_fmaf(w2d, v[ry][rx], num);
            den = den + w2d;
        }
    return num / den;
}

// ---- VS_WARP_LANCZOS2_SEP: the separable member of the sampler family ------------------------------------------------
// Same weights (lanczos2_fma), sampling position and 4 x 4 live window as the contracted form; the window sum is taken rows first,
// then columns, the denominator is the product of the two 1-D weight sums, and ONE correctly rounded reciprocal serves every channel:
//   h[ry] = fma(wx4, v4, fma(wx3, v3, fma(wx2, v2, wx1 * v1)))      num = fma(wy4, h4, fma(wy3, h3, fma(wy2, h2, wy1 * h1)))
//   den   = ((wx1 + wx2) + (wx3 + wx4)) * ((wy1 + wy2) + (wy3 + wy4))      out = num * RN(1 / den)
// A reassociation of generators.cpp:687-697 (equal in real arithmetic), inside the slack the reference's own non-strict_float build has;
// it enters a benchmark only through SURVEY 8(d)'s integer gate against the UN-contracted order (tests/test_warp_gate_gpu.py).  CPU
// twin: the checker's VSO_WARP_LANCZOS2_SEPARABLE, bit for bit (tests/test_warp_fast_gpu.py).
__device__ __forceinline__ float lanczos_separable_den(const float wx[4], const float wy[4]) {
    return ((wx[0] + wx[1]) + (wx[2] + wx[3])) * ((wy[0] + wy[1]) + (wy[2] + wy[3]));
}
// one channel: v[ry][rx] = the live 4x4 window as floats; rden = RN(1 / den)
__device__ __forceinline__ float lanczos_separable_combine(const float v[4][4], const float wx[4], const float wy[4], float rden) {
    float h[4];
#pragma unroll
    for (int ry = 0; ry < 4; ry++)
        h[ry] = __builtin_fmaf(wx[3], v[ry][3], __builtin_fmaf(wx[2], v[ry][2], __builtin_fmaf(wx[1], v[ry][1], wx[0] * v[ry][0])));
    const float num = __builtin_fmaf(wy[3], h[3], __builtin_fmaf(wy[2], h[2], __builtin_fmaf(wy[1], h[1], wy[0] * h[0])));
    return num * rden;
}

// ---- VS_WARP_BILINEAR_CV: cv::warpAffine's fixed-point coordinates (OpenCV 4.x imgwarp.cpp, AB_BITS = 10, INTER_BITS = 5) -------------------
// cvRound = round half to even; a value outside the int range saturates (as the oracle's cv_round_sat; no frame gets near it)
__device__ __forceinline__ int cv_round_sat(double v) {
    v = fmin(fmax(rint(v), -2147483648.0), 2147483647.0);
    return (v == v) ? (int)v : 0;
}
// adelta[x] / bdelta[x]: cvRound(m * x * 1024) -- the product m * x rounds to double before the scaling (exact), as in the source
__device__ __forceinline__ int cv_delta(double m, int x) { return cv_round_sat(m * (double)x * 1024.0); }
// X0 / Y0 of a row: cvRound((m_y * y + m_t) * 1024) + round_delta (16), two's-complement wrap like the int arithmetic it restates
__device__ __forceinline__ int cv_row_origin(double my, double mt, int y) {
    return (int)((unsigned)cv_round_sat((my * (double)y + mt) * 1024.0) + 16u);
}

// Lanczos2 sample of a single-channel u8 image with clamp-to-edge addressing:
// generators.cpp:672-697 (sparse_warpdiff) == :469-498 (sparse_ica).  rx inner, ry outer,
// separate num / den accumulators from 0, one IEEE divide.
__device__ __forceinline__ float lanczos_sample_u8(const uint8_t* __restrict__ img, int w, int h, int stride,
                                                   float Wx, float Wy) {
    float flx = floorf(Wx), fly = floorf(Wy);
    float frx = Wx - flx, fry = Wy - fly;
    float wx[4], wy[4];
    lanczos_weights4(frx, wx);
    lanczos_weights4(fry, wy);
    int ix = sample_index(flx, w), iy = sample_index(fly, h);
    int xs[4];
#pragma unroll
    for (int r = 0; r < 4; r++) xs[r] = clampi(ix + r - 1, 0, w - 1);
    float num = 0.0f, den = 0.0f;
#pragma unroll
    for (int ry = 0; ry < 4; ry++) {
        const uint8_t* row = img + (size_t)clampi(iy + ry - 1, 0, h - 1) * stride;
#pragma unroll
        for (int rx = 0; rx < 4; rx++) {
            float w2d = wx[rx] * wy[ry];
            float val = (float)row[xs[rx]];
            num = num + w2d * val;
            den = den + w2d;
        }
    }
    return num / den;
}

// ---- the same sampler, arranged for the Gauss-Newton loop -------------------------------------
// Identical fp32 operations per value (so identical results), but: the eight weights run as four
// packed {x,y} Horner chains, and the 4x4 window is fetched as four unaligned dword loads (the
// hardware handles unaligned global loads) from a window origin clamped into the image.  Lanes
// whose window touches the border re-select bytes with v_perm_b32 so that they see exactly the
// clamp-to-edge pixels; that fix-up runs only in waves that contain such a lane.
typedef float f2v __attribute__((ext_vector_type(2)));

template <bool EDGE>
__device__ __forceinline__ f2v lanczos2_pk(f2v x) {
    f2v x2 = x * x;
    f2v v = 0.000858519f;
    v = -0.0158853f + v * x2;
    v = 0.128693f + v * x2;
    v = -0.583468f + v * x2;
    v = 1.52229f + v * x2;
    v = -2.05238f + v * x2;
    v = 0.999861f + v * x2;
    if (EDGE) {   // only taps 1 and 4 (-1-frac, 2-frac) can reach |x| >= 2
        v.x = fabsf(x.x) >= 2.0f ? 0.0f : v.x;
        v.y = fabsf(x.y) >= 2.0f ? 0.0f : v.y;
    }
    return v;
}

__device__ __forceinline__ uint32_t load_u32_unaligned(const uint8_t* p) {
    uint32_t r;
    __builtin_memcpy(&r, p, 4);
    return r;
}
// The same load through a pointer known to address global memory.  Inside a function that is not inlined into its kernel
// a plain pointer is generic and every access becomes a flat_load (aperture check, both wait counters); the cast keeps
// the gathers on global_load.
#define VS_GLOBAL_AS __attribute__((address_space(1)))
typedef const VS_GLOBAL_AS uint8_t* vs_gbytes;
struct __attribute__((packed)) vs_packed_u32 { uint32_t v; };
__device__ __forceinline__ uint32_t load_u32_unaligned(vs_gbytes p) { return ((const VS_GLOBAL_AS vs_packed_u32*)p)->v; }
// ... and through a pointer into LDS (a level image staged there): the two aligned dwords around the address + v_alignbyte.
// The staged image is padded so that the second dword always exists.
#define VS_LDS_BYTES_AS __attribute__((address_space(3)))
typedef const VS_LDS_BYTES_AS uint8_t* vs_lbytes;
__device__ __forceinline__ uint32_t load_u32_unaligned(vs_lbytes p) {
    const uint32_t addr = (uint32_t)(uintptr_t)p;
    const VS_LDS_BYTES_AS uint32_t* q = (const VS_LDS_BYTES_AS uint32_t*)(uintptr_t)(addr & ~3u);
    return __builtin_amdgcn_alignbyte(q[1], q[0], addr & 3u);
}

// requires w >= 4.  P: const uint8_t* (generic), vs_gbytes (global) or vs_lbytes (LDS).
// Split in two so that a caller can put the gathers of several points in flight before it consumes the first of them
// (one workgroup owns a frame pair: nothing else hides the memory latency): lanczos_fetch issues the four row loads,
// lanczos_finish does the arithmetic.  lanczos_sample_u8_fast = finish(fetch), the same fp32 operations either way.
struct LanczosFetch {
    uint32_t r[4];
    float frx, fry;
    int ix;
};
template <typename P>
__device__ __forceinline__ LanczosFetch lanczos_fetch(P img, int w, int h, int stride, float Wx, float Wy) {
    LanczosFetch f;
    const float flx = floorf(Wx), fly = floorf(Wy);
    const int ix = sample_index(flx, w), iy = sample_index(fly, h);
    f.frx = Wx - flx; f.fry = Wy - fly; f.ix = ix;
    const int xb = clampi(ix - 1, 0, w - 4);                  // window origin, always inside the row
#pragma unroll
    for (int ry = 0; ry < 4; ry++) {
        const int row = clampi(iy + ry - 1, 0, h - 1);
        const int off = __mul24(row, stride) + xb;     // (a level image is far below 2^31 bytes, rows and stride below 2^24: no 64-bit multiply)
        VS_BOUNDS_CHECK(off + 3, h * stride, 130);
        f.r[ry] = load_u32_unaligned(img + off);
    }
    return f;
}
// the scalar form of lanczos2_pk (same operations per value)
template <bool EDGE>
__device__ __forceinline__ float lanczos2_s(float x) {
    const float x2 = x * x;
    float v = 0.000858519f;
    v = -0.0158853f + v * x2;
    v = 0.128693f + v * x2;
    v = -0.583468f + v * x2;
    v = 1.52229f + v * x2;
    v = -2.05238f + v * x2;
    v = 0.999861f + v * x2;
    if (EDGE) v = fabsf(x) >= 2.0f ? 0.0f : v;
    return v;
}
#ifndef VS_SAMPLE_SCALAR_WEIGHTS
#define VS_SAMPLE_SCALAR_WEIGHTS 0
#endif
__device__ __forceinline__ float lanczos_finish(const LanczosFetch& f, int w) {
    const int ix = f.ix;
#if VS_SAMPLE_SCALAR_WEIGHTS
    const float wx[4] = {lanczos2_s<true>(-1.0f - f.frx), lanczos2_s<false>(0.0f - f.frx), lanczos2_s<false>(1.0f - f.frx),
                         lanczos2_s<true>(2.0f - f.frx)};
    const float wy[4] = {lanczos2_s<true>(-1.0f - f.fry), lanczos2_s<false>(0.0f - f.fry), lanczos2_s<false>(1.0f - f.fry),
                         lanczos2_s<true>(2.0f - f.fry)};
#else
    const f2v fr = {f.frx, f.fry};
    const f2v w0 = lanczos2_pk<true>(f2v{-1.0f, -1.0f} - fr), w1 = lanczos2_pk<false>(f2v{0.0f, 0.0f} - fr),
              w2 = lanczos2_pk<false>(f2v{1.0f, 1.0f} - fr), w3 = lanczos2_pk<true>(f2v{2.0f, 2.0f} - fr);
    const float wx[4] = {w0.x, w1.x, w2.x, w3.x}, wy[4] = {w0.y, w1.y, w2.y, w3.y};
#endif
    uint32_t r[4] = {f.r[0], f.r[1], f.r[2], f.r[3]};
    const bool edge = ix < 1 || ix + 2 >= w;
    if (__any(edge)) {                                        // wave-uniform branch
        const int xb = clampi(ix - 1, 0, w - 4);
        uint32_t sel = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) sel |= (uint32_t)(clampi(ix - 1 + k, 0, w - 1) - xb) << (8 * k);
#pragma unroll
        for (int ry = 0; ry < 4; ry++) r[ry] = __builtin_amdgcn_perm(r[ry], r[ry], sel);
    }
    float num = 0.0f, den = 0.0f;
#pragma unroll
    for (int ry = 0; ry < 4; ry++) {
#pragma unroll
        for (int rx = 0; rx < 4; rx++) {
            const float w2d = wx[rx] * wy[ry];
            const float val = (float)((r[ry] >> (8 * rx)) & 0xffu);
            num = num + w2d * val;
            den = den + w2d;
        }
    }
    return num / den;
}
template <typename P>
__device__ __forceinline__ float lanczos_sample_u8_fast(P img, int w, int h, int stride, float Wx, float Wy) {
    return lanczos_finish(lanczos_fetch(img, w, h, stride, Wx, Wy), w);
}

// imgproc.cpp:69-75 / 98-103: centre-based double transform -> float kernel arguments.
// `w*0.5f` is a float in the reference, promoted to double inside the expression.
__device__ __forceinline__ void ul_params_sparse(const double T[4], int w, int h, float p[4]) {
    double hw = (double)((float)w * 0.5f), hh = (double)((float)h * 0.5f);
    p[0] = (float)T[0];
    p[1] = (float)T[1];
    p[2] = (float)(T[2] - T[0] * hw + T[1] * hh);
    p[3] = (float)(T[3] - T[1] * hw - T[0] * hh);
}

// ---- wave / block reductions ---------------------------------------------------------------
// Cross-lane data movement uses DPP (row_shr / row_bcast: one VALU move each) instead of __shfl
// (ds_bpermute through the LDS crossbar, ~100 cycles of latency per step): a 64-lane inclusive scan is
// six dependent move+op pairs.  Lanes with no source lane receive 0.
template <int CTRL, int ROW_MASK = 0xf, int BANK_MASK = 0xf>
__device__ __forceinline__ int dpp_mov0(int v) {
    return __builtin_amdgcn_update_dpp(0, v, CTRL, ROW_MASK, BANK_MASK, true);
}
template <int CTRL, int ROW_MASK = 0xf>
__device__ __forceinline__ double dpp_mov0(double v) {
    const int lo = dpp_mov0<CTRL, ROW_MASK>(__double2loint(v)), hi = dpp_mov0<CTRL, ROW_MASK>(__double2hiint(v));
    return __hiloint2double(hi, lo);
}
// Kogge-Stone inside each row of 16 (row_shr 1,2,4,8), then row_bcast15 into rows 1,3 and row_bcast31 into rows 2,3
__device__ __forceinline__ int wave_incl_scan(int v) {
    v += dpp_mov0<0x111>(v);
    v += dpp_mov0<0x112>(v);
    v += dpp_mov0<0x114>(v);
    v += dpp_mov0<0x118>(v);
    v += dpp_mov0<0x142, 0xa>(v);
    v += dpp_mov0<0x143, 0xc>(v);
    return v;
}
__device__ __forceinline__ int wave_total(int incl_scan) { return __builtin_amdgcn_readlane(incl_scan, 63); }
// sum over the wave; the result is valid in every lane (read back from lane 63 of the scan)
__device__ __forceinline__ double wave_sum(double v) {
    v += dpp_mov0<0x111>(v);
    v += dpp_mov0<0x112>(v);
    v += dpp_mov0<0x114>(v);
    v += dpp_mov0<0x118>(v);
    v += dpp_mov0<0x142, 0xa>(v);
    v += dpp_mov0<0x143, 0xc>(v);
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), 63), __builtin_amdgcn_readlane(__double2loint(v), 63));
}
// min over the wave of non-negative ints (lanes without a source contribute INT_MAX)
__device__ __forceinline__ int wave_min_nonneg(int v) {
    // work on the complement so that the DPP fill value 0 is the identity of max
    int c = 0x7fffffff - v;
    c = max(c, dpp_mov0<0x111>(c));
    c = max(c, dpp_mov0<0x112>(c));
    c = max(c, dpp_mov0<0x114>(c));
    c = max(c, dpp_mov0<0x118>(c));
    c = max(c, dpp_mov0<0x142, 0xa>(c));
    c = max(c, dpp_mov0<0x143, 0xc>(c));
    return 0x7fffffff - __builtin_amdgcn_readlane(c, 63);
}
__device__ __forceinline__ unsigned long long wave_max_u64(unsigned long long v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        unsigned long long o = __shfl_down(v, off, kWave);
        v = o > v ? o : v;
    }
    return v;
}
__device__ __forceinline__ unsigned wave_max_u32(unsigned v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        unsigned o = __shfl_down(v, off, kWave);
        v = o > v ? o : v;
    }
    return v;
}

// Sum K doubles over the whole block.  Every thread returns the same total: per-wave shuffle
// tree, lane 0 of each wave parks its partials in LDS, one barrier, then every thread adds the
// per-wave partials in wave order.  `lds` needs (blockDim.x/64)*K doubles and must not be in
// use by anyone else until the NEXT call with a different buffer (callers ping-pong two).
template <int K>
__device__ __forceinline__ void block_sum(double v[K], double* lds) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
#pragma unroll
    for (int k = 0; k < K; k++) {
        double s = wave_sum(v[k]);
        if (lane == 0) lds[wave * K + k] = s;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < K; k++) {
        double s = 0.0;
        for (int wv = 0; wv < nw; wv++) s += lds[wv * K + k];
        v[k] = s;
    }
}

// first half of block_sum only: per-wave sums parked in LDS (lds[wave*K + k]); the caller puts a barrier behind it
template <int K, typename LP = double*>
__device__ __forceinline__ void wave_partials(const double v[K], LP lds) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < K; k++) {
        double s = wave_sum(v[k]);
        if (lane == 0) lds[wave * K + k] = s;
    }
}

// the same with the slot given by the caller (a hardware wave standing in for virtual wave `slot`, vs_align_kernels.inc)
template <int K, typename LP = double*>
__device__ __forceinline__ void wave_partials_at(const double v[K], LP lds, int slot) {
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int k = 0; k < K; k++) {
        double s = wave_sum(v[k]);
        if (lane == 0) lds[slot * K + k] = s;
    }
}

// Eight sums over the wave at once: a butterfly that halves the number of values a lane carries at each of the first three
// steps (lane pairs at distance 1, 2, 4 exchange the half the partner keeps), then three steps on the one value that is
// left -- 10 fp64 additions per lane instead of the 48 of eight separate wave_sum trees.  Lanes 0..7 end up holding the
// totals (lane j the value with index bit-reversed j) and write lds[wave_slot * 8 + index].  The tree is fixed, so the
// result does not depend on anything but the inputs.
template <int CTRL, int ROW_MASK = 0xf, int BANK_MASK = 0xf>
__device__ __forceinline__ double dpp_mov_keep(double old, double v) {
    const int lo = __builtin_amdgcn_update_dpp(__double2loint(old), __double2loint(v), CTRL, ROW_MASK, BANK_MASK, false);
    const int hi = __builtin_amdgcn_update_dpp(__double2hiint(old), __double2hiint(v), CTRL, ROW_MASK, BANK_MASK, false);
    return __hiloint2double(hi, lo);
}
template <typename LP>
__device__ __forceinline__ void wave_partials8_butterfly(const double v[8], LP lds, int wave_slot) {
    const int lane = threadIdx.x & 63;
    const bool b0 = lane & 1, b1 = lane & 2, b2 = lane & 4;
    double k4[4], k2[2], k1;
#pragma unroll
    for (int i = 0; i < 4; i++) {                               // distance 1: quad_perm [1,0,3,2]
        const double keep = b0 ? v[4 + i] : v[i], send = b0 ? v[i] : v[4 + i];
        k4[i] = keep + dpp_mov_keep<0xB1>(send, send);
    }
#pragma unroll
    for (int i = 0; i < 2; i++) {                               // distance 2: quad_perm [2,3,0,1]
        const double keep = b1 ? k4[2 + i] : k4[i], send = b1 ? k4[i] : k4[2 + i];
        k2[i] = keep + dpp_mov_keep<0x4E>(send, send);
    }
    {                                                           // distance 4: row_shl:4 into banks 0,2; row_shr:4 into banks 1,3
        const double keep = b2 ? k2[1] : k2[0], send = b2 ? k2[0] : k2[1];
        double recv = dpp_mov_keep<0x104, 0xf, 0x5>(send, send);
        recv = dpp_mov_keep<0x114, 0xf, 0xa>(recv, send);
        k1 = keep + recv;
    }
    k1 = k1 + dpp_mov_keep<0x128>(k1, k1);                      // distance 8: row_ror:8
    {                                                           // distance 16: swizzle, xor 0x10 inside each half
        const int lo = __builtin_amdgcn_ds_swizzle(__double2loint(k1), (0x10 << 10) | 0x1f);
        const int hi = __builtin_amdgcn_ds_swizzle(__double2hiint(k1), (0x10 << 10) | 0x1f);
        k1 = k1 + __hiloint2double(hi, lo);
    }
    {                                                           // distance 32
        const int src = (lane ^ 32) << 2;
        const int lo = __builtin_amdgcn_ds_bpermute(src, __double2loint(k1));
        const int hi = __builtin_amdgcn_ds_bpermute(src, __double2hiint(k1));
        k1 = k1 + __hiloint2double(hi, lo);
    }
    if (lane < 8) lds[wave_slot * 8 + ((lane & 1) << 2 | (lane & 2) | (lane >> 2))] = k1;
}

// ---- 4x4 symmetric eigen-solver + conditioned pseudo-inverse --------------------------------
// alignment.cpp:555-583 uses cv::SVD and Mat::inv(DECOMP_SVD) on the symmetric PSD Hessian.
// Singular values of such a matrix are its eigenvalues; cyclic Jacobi sweeps deliver them to ~1e-16 relative.
// One wave runs this alone, so the time is the length of the dependent fp64 chain: the six rotations of a sweep are taken in
// the order (0,1)(2,3) | (0,2)(1,3) | (0,3)(1,2) -- the two rotations of a stage touch disjoint rows and columns, their
// angles do not depend on each other, and their sqrt / divide / rsqrt chains run interleaved.  (The CPU oracle sweeps
// (0,1),(0,2),(0,3),(1,2),(1,3),(2,3); both converge to the same eigenvalues, the results differ in the last bits.)
// Fully unrolled with compile-time indices so that a[][] and V[] live in registers (behind a pointer they would
// be scratch memory: ~500 cycles per element access).
struct JacobiRot { double c, s, t; };
__device__ __forceinline__ JacobiRot jacobi_rot(double app, double aqq, double apq) {
    // t = tan of the rotation angle = sgn(theta) / (|theta| + sqrt(theta^2 + 1)), theta = (aqq-app)/(2 apq),
    // written with d = aqq - app so that it costs one sqrt and one divide; c = 1/sqrt(t^2+1) by rsqrt.
    // apq == 0: the identity (c = 1, s = 0), without a branch.
    const double d = aqq - app;
    const double r = sqrt(d * d + 4.0 * apq * apq);
    // (the divide runs unconditionally, on 0 / 1 when apq == 0: a select around it would become a branch and keep the two
    // rotations of a stage from interleaving)
    const double den = apq == 0.0 ? 1.0 : fabs(d) + r;
    JacobiRot R;
    R.t = (d >= 0.0 ? 2.0 * apq : -2.0 * apq) / den;
    R.c = rsqrt(R.t * R.t + 1.0);
    R.s = R.t * R.c;
    return R;
}
// Rotations (P1,Q1) and (P2,Q2), {P1,Q1} and {P2,Q2} disjoint, on the symmetric matrix a (both halves kept): the 2x2
// diagonal blocks by the closed form (app - t apq, aqq + t apq, 0), the 2x2 cross block as R1^T B R2, V's four columns --
// 78 multiply-adds where two full two-sided products take 144.
template <int P1, int Q1, int P2, int Q2>
__device__ __forceinline__ void jacobi_stage(double a[4][4], double* V) {
    const JacobiRot R1 = jacobi_rot(a[P1][P1], a[Q1][Q1], a[P1][Q1]), R2 = jacobi_rot(a[P2][P2], a[Q2][Q2], a[P2][Q2]);
    {
        const double x1 = R1.t * a[P1][Q1], x2 = R2.t * a[P2][Q2];
        a[P1][P1] = a[P1][P1] - x1; a[Q1][Q1] = a[Q1][Q1] + x1; a[P1][Q1] = 0.0; a[Q1][P1] = 0.0;
        a[P2][P2] = a[P2][P2] - x2; a[Q2][Q2] = a[Q2][Q2] + x2; a[P2][Q2] = 0.0; a[Q2][P2] = 0.0;
    }
    // cross block: rows P1,Q1 x columns P2,Q2
    double bpp = a[P1][P2], bpq = a[P1][Q2], bqp = a[Q1][P2], bqq = a[Q1][Q2];
    {   // rows by R1
        const double n_pp = R1.c * bpp - R1.s * bqp, n_qp = R1.s * bpp + R1.c * bqp;
        const double n_pq = R1.c * bpq - R1.s * bqq, n_qq = R1.s * bpq + R1.c * bqq;
        bpp = n_pp; bqp = n_qp; bpq = n_pq; bqq = n_qq;
    }
    {   // columns by R2
        const double n_pp = R2.c * bpp - R2.s * bpq, n_pq = R2.s * bpp + R2.c * bpq;
        const double n_qp = R2.c * bqp - R2.s * bqq, n_qq = R2.s * bqp + R2.c * bqq;
        bpp = n_pp; bpq = n_pq; bqp = n_qp; bqq = n_qq;
    }
    a[P1][P2] = bpp; a[P2][P1] = bpp; a[P1][Q2] = bpq; a[Q2][P1] = bpq;
    a[Q1][P2] = bqp; a[P2][Q1] = bqp; a[Q1][Q2] = bqq; a[Q2][Q1] = bqq;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const double vkp = V[k * 4 + P1], vkq = V[k * 4 + Q1];
        V[k * 4 + P1] = R1.c * vkp - R1.s * vkq; V[k * 4 + Q1] = R1.s * vkp + R1.c * vkq;
        const double wkp = V[k * 4 + P2], wkq = V[k * 4 + Q2];
        V[k * 4 + P2] = R2.c * wkp - R2.s * wkq; V[k * 4 + Q2] = R2.s * wkp + R2.c * wkq;
    }
}
__device__ __forceinline__ void jacobi_eig4(const double* Hin, double* eval, double* V) {
    double a[4][4];
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) { a[i][j] = Hin[i * 4 + j]; V[i * 4 + j] = (i == j) ? 1.0 : 0.0; }
    for (int sweep = 0; sweep < 32; sweep++) {
        double off = 0.0, diag = 0.0;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            diag += a[i][i] * a[i][i];
#pragma unroll
            for (int j = i + 1; j < 4; j++) off += a[i][j] * a[i][j];
        }
        if (off <= 1e-300 || off <= 1e-32 * diag) break;   // off-diagonal mass below 1e-16 of the diagonal's
        jacobi_stage<0, 1, 2, 3>(a, V);
        jacobi_stage<0, 2, 1, 3>(a, V);
        jacobi_stage<0, 3, 1, 2>(a, V);
    }
#pragma unroll
    for (int i = 0; i < 4; i++) eval[i] = a[i][i];
}

#ifndef VS_COND_INLINE
#define VS_COND_INLINE __forceinline__   // (out of line its 32 doubles travel through scratch memory)
#endif
// cond = smax/(smin+1e-10); cond > 1e6 => H += 1e-6*smax*I (alignment.cpp:561-572);
// Hinv = V diag(1/w) V^T dropping w <= 2*eps*sum(w) (OpenCV's DECOMP_SVD back-substitution).
__device__ VS_COND_INLINE double condition_and_invert(double* Hio, double* Hinv_out) {
    double H[16], Hinv[16];
#pragma unroll
    for (int i = 0; i < 16; i++) H[i] = Hio[i];
    double ev[4], V[16];
    jacobi_eig4(H, ev, V);
    double max_sv = 0.0, min_sv = 1e300;
#pragma unroll
    for (int i = 0; i < 4; i++) { double a = fabs(ev[i]); max_sv = fmax(max_sv, a); min_sv = fmin(min_sv, a); }
    double cond = max_sv / (min_sv + 1e-10);
    if (cond > 1e6) {
        double lambda = 1e-6 * max_sv;
#pragma unroll
        for (int i = 0; i < 4; i++) H[i * 4 + i] += lambda;
        jacobi_eig4(H, ev, V);
    }
    double sum = 0.0;
#pragma unroll
    for (int i = 0; i < 4; i++) sum += fabs(ev[i]);
    double thresh = 2.0 * 2.220446049250313e-16 * sum;
    // one reciprocal per eigenvalue (as OpenCV's own back-substitution does: wi = 1 / wi) and the upper triangle only --
    // sixty-four fp64 divisions on one wave were a third of this function's time
    double inv[4];
#pragma unroll
    for (int k = 0; k < 4; k++) inv[k] = fabs(ev[k]) > thresh ? 1.0 / ev[k] : 0.0;
#pragma unroll
    for (int r = 0; r < 4; r++)
#pragma unroll
        for (int c = r; c < 4; c++) {
            double s = 0.0;
#pragma unroll
            for (int k = 0; k < 4; k++) s += V[r * 4 + k] * V[c * 4 + k] * inv[k];
            Hinv[r * 4 + c] = s;
            Hinv[c * 4 + r] = s;
        }
#pragma unroll
    for (int i = 0; i < 16; i++) { Hio[i] = H[i]; Hinv_out[i] = Hinv[i]; }
    return cond;
}

// ---- similarity-transform algebra in fp64 (imgproc.cpp:361-411) ------------------------------
// compose: apply d first, then t.
__device__ __forceinline__ void compose(const double d[4], const double t[4], double out[4]) {
    double p1 = 1.0 + d[0], q1 = d[1], p2 = 1.0 + t[0], q2 = t[1];
    double A3 = (p2 * p1 - q2 * q1) - 1.0;
    double B3 = (p2 * q1 + q2 * p1);
    double TX3 = p2 * d[2] - q2 * d[3] + t[2];
    double TY3 = q2 * d[2] + p2 * d[3] + t[3];
    out[0] = A3; out[1] = B3; out[2] = TX3; out[3] = TY3;
}
__device__ __forceinline__ void inverse(const double t[4], double out[4]) {
    double p = 1.0 + t[0], q = t[1];
    double denom = p * p + q * q;
    out[0] = (p / denom) - 1.0;
    out[1] = -q / denom;
    out[2] = (-p * t[2] - q * t[3]) / denom;
    out[3] = (q * t[2] - p * t[3]) / denom;
}
__device__ __forceinline__ void warp_center(const double t[4], double x, double y, double cx, double cy,
                                            double& ox, double& oy) {
    double px = x - cx, py = y - cy;
    ox = (1 + t[0]) * px - t[1] * py + cx + t[2];
    oy = t[1] * px + (1 + t[0]) * py + cy + t[3];
}
// the four corners (0,0),(w-1,0),(0,h-1),(w-1,h-1) about (w/2,h/2): alignment.cpp:587-593
__device__ __forceinline__ void warp_corners(const double t[4], int w, int h, double c[8]) {
    double cx = w * 0.5, cy = h * 0.5;
    double x1 = (double)((float)w - 1.f), y1 = (double)((float)h - 1.f);
    warp_center(t, 0.0, 0.0, cx, cy, c[0], c[1]);
    warp_center(t, x1, 0.0, cx, cy, c[2], c[3]);
    warp_center(t, 0.0, y1, cx, cy, c[4], c[5]);
    warp_center(t, x1, y1, cx, cy, c[6], c[7]);
}
// max over the four corners of Point::distance, nested as the reference nests it (alignment.cpp:647-649 / 670-672): std::max(std::max(ul, ur),
// std::max(ll, lr)) with std::max(x, y) = (x < y) ? y : x -- NOT fmax: when the update has gone NaN (a singular level: Hinv and then T are NaN)
// every distance is NaN, the reference's displacement is NaN, `NaN < threshold` is false and the level runs out of iterations and is REFUSED.
// fmax(0, NaN) = 0 called that "converged" and the frame came back aligned with a NaN transform (found by the random sweeps, round 4).
// sqrt is correctly rounded, hence monotone: max_k sqrt(d_k) == sqrt(max_k d_k) bit for bit, so one square root instead of four.
__device__ __forceinline__ double corner_move(const double a[8], const double b[8]) {
    double d[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const double dx = a[2 * k] - b[2 * k], dy = a[2 * k + 1] - b[2 * k + 1];
        d[k] = dx * dx + dy * dy;
    }
    const double u = d[0] < d[1] ? d[1] : d[0], l = d[2] < d[3] ? d[3] : d[2];
    return sqrt(u < l ? l : u);
}

}  // namespace vsd
