// vs_warp.hip -- tuned bgr_image_warp for interleaved 3-channel u8 / u16 frames (the 1080p / 4K roofline kernel).
//
// Sampler semantics = vs_k_bgr_warp_generic (vs_kernels.hip), i.e. the reference's Lanczos2 sampler
// (generators.cpp:672-697: 5x5 window, polynomial weights, rx inner / ry outer, num and den summed
// separately, one divide) or image_warp's bilinear (generators.cpp:148-163), evaluated per channel at
// image_warp's coordinates (generators.cpp:141-142).  Results are bit-identical to the generic kernel
// and to the CPU oracle: same fp32 operations in the same order, no FMA contraction.
//
// Structure (one 256-thread workgroup = one 64x16 output tile of one frame):
//   1. The similarity is affine, so the tile's source footprint is the bounding box of its four
//      corners (fp32 rounding is monotonic, so the corners bound every pixel exactly).  For the
//      near-identity transforms of stabilisation that is ~67x19 pixels.
//   2. The footprint (+ the Lanczos halo) is read from HBM once with aligned 12-byte loads (4 pixels),
//      converted to float ONCE, and parked in LDS as one float4 {B,G,R,1} per source pixel.  Clamp /
//      constant-0 borders are resolved here, so the inner loop has no clamps and no global loads.
//   3. Lane = output column, each wave owns 4 output rows.  One tap = one ds_read_b128 at an immediate
//      offset from a single address register; neighbouring lanes read neighbouring 16-byte slots, so
//      the reads are conflict-free.  The arithmetic is packed fp32 (v_pk_mul_f32 / v_pk_add_f32 on
//      {B,G} and {R,den} pairs -- the trailing 1.0 makes den += w fall out of the same instruction,
//      and w*1.0 is exact) and the eight polynomial weights run as four {x,y} packed Horner chains.
//      Packed instructions issue at the scalar-VALU rate on gfx950, so this halves the VALU time;
//      each component still sees exactly the reference's sequence of roundings.
//   4. Results are transposed through a 3 KB LDS tile so that the stores leave as aligned dwords,
//      192 contiguous bytes per output row.
// Tiles whose footprint does not fit the LDS window (large rotation / zoom) take the generic
// global-memory path inside the same kernel, so every transform is supported.
//
// Cost model (DESIGN.md "bgr_image_warp roofline"): measured with rocprofv3 PMC the exact-order
// Lanczos2 form needs a few hundred VALU instructions per output pixel against 6 bytes of HBM
// traffic, and a wave64 VALU instruction occupies its SIMD for 4 cycles: the kernel is VALU-bound,
// not HBM-bound, on gfx950 (bilinear is ~4x lighter).
#include "vs_kernels.hpp"
#include "vs_device.hpp"

using namespace vsd;

namespace {

constexpr int WT_W = 64, WT_H = 16;      // output tile
constexpr int WS_W = 80;                 // staged source pixels per row (multiple of 4)
constexpr int WS_H = 24;                 // staged source rows

typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x3 __attribute__((ext_vector_type(3)));

__device__ __forceinline__ float lerpf(float a, float b, float t) { return a * (1.0f - t) + b * t; }

__device__ __forceinline__ float ub(uint32_t q, int k) { return (float)((q >> (8 * k)) & 0xffu); }

// build rule: round half up, saturate = clamp(floor(v + 0.5), 0, maxv).  maxv is an integer, so clamping first and
// then truncating (float -> u32 conversion truncates, = floor for non-negative values) gives the same integer.
__device__ __forceinline__ uint32_t store_u(float v, float maxv) {
    return (uint32_t)__builtin_amdgcn_fmed3f(v + 0.5f, 0.0f, maxv);
}

// generators.cpp:31-47 on an {x,y} pair: identical roundings per component, packed instructions.
// EDGE = the argument can reach |x| >= 2 (taps 1 and 4: -1-frac, 2-frac with frac in [0,1]); for taps 2
// and 3 (|x| <= 1) the select of generators.cpp:46 can never fire and is dropped.
// Three correctly rounded quotients over one denominator.  This is hipcc's own fp32 division expansion
// (v_div_scale, v_rcp, two Newton steps on the reciprocal-quotient pair, v_div_fmas, v_div_fixup) with the
// parts that are no-ops here removed: den is a sum of Lanczos weights (0.99..1.05) and the numerators are
// bounded byte sums, so no operand scaling and no special-case fix-up ever applies; what is left is the
// same fma sequence, and the reciprocal refinement is shared by the three channels (18 instructions
// instead of 33).  Outside the safe range it falls back to operator/.
__device__ __forceinline__ void div3_exact(float n0, float n1, float n2, float den, float& q0, float& q1, float& q2) {
    if (den > 0.5f && den < 2.0f) {
        float r = __builtin_amdgcn_rcpf(den);
        const float e = __builtin_fmaf(-den, r, 1.0f);
        r = __builtin_fmaf(e, r, r);
        float q, t;
        q = n0 * r; t = __builtin_fmaf(-den, q, n0); q = __builtin_fmaf(t, r, q); t = __builtin_fmaf(-den, q, n0); q0 = __builtin_fmaf(t, r, q);
        q = n1 * r; t = __builtin_fmaf(-den, q, n1); q = __builtin_fmaf(t, r, q); t = __builtin_fmaf(-den, q, n1); q1 = __builtin_fmaf(t, r, q);
        q = n2 * r; t = __builtin_fmaf(-den, q, n2); q = __builtin_fmaf(t, r, q); t = __builtin_fmaf(-den, q, n2); q2 = __builtin_fmaf(t, r, q);
    } else {
        q0 = n0 / den; q1 = n1 / den; q2 = n2 / den;
    }
}

// u16 pixels: a pair of lanes owns 2 pixels = 12 bytes = 3 dwords {B0|G0, R0|B1, G1|R1}
__device__ __forceinline__ void pair_pack_bgr16(const uint32_t o[3], int odd, uint32_t& d0, uint32_t& d1) {
    const uint32_t bg = o[0] | (o[1] << 16);
    const uint32_t nbg = (uint32_t)dpp_mov0<0x101>((int)bg);   // row_shl:1 (only even lanes use it)
    d0 = odd ? (o[1] | (o[2] << 16)) : bg;            // odd lane: G1|R1 ; even lane: B0|G0
    d1 = o[2] | (nbg << 16);                           // even lane only: R0|B1
}

// EDGE_X / EDGE_Y: that component's argument can reach |x| >= 2 (taps 1 and 4: -1-frac, 2-frac with frac in [0,1]);
// for taps 2 and 3 (|x| <= 1) the select of generators.cpp:46 can never fire and is dropped.
template <bool EDGE_X, bool EDGE_Y>
__device__ __forceinline__ f2 lanczos2_pk(f2 x) {
    f2 x2 = x * x;
    f2 v = 0.000858519f;
    v = -0.0158853f + v * x2;
    v = 0.128693f + v * x2;
    v = -0.583468f + v * x2;
    v = 1.52229f + v * x2;
    v = -2.05238f + v * x2;
    v = 0.999861f + v * x2;
    if (EDGE_X) v.x = fabsf(x.x) >= 2.0f ? 0.0f : v.x;
    if (EDGE_Y) v.y = fabsf(x.y) >= 2.0f ? 0.0f : v.y;
    return v;
}

// VS_WARP_LANCZOS2_FAST: the same polynomial with every multiply-add fused (v_pk_fma_f32): one rounding per Horner step
// instead of two.  Not the reference's sequence of roundings any more -- results stay within the north star's 1 ULP /
// 1 LSB gate of the exact kernel (tests/test_kernels_gpu.py), at ~0.6x the VALU work.
template <bool EDGE_X, bool EDGE_Y>
__device__ __forceinline__ f2 lanczos2_pk_fma(f2 x) {
    const f2 x2 = x * x;
    f2 v = 0.000858519f;
    v = __builtin_elementwise_fma(v, x2, (f2)(-0.0158853f));
    v = __builtin_elementwise_fma(v, x2, (f2)(0.128693f));
    v = __builtin_elementwise_fma(v, x2, (f2)(-0.583468f));
    v = __builtin_elementwise_fma(v, x2, (f2)(1.52229f));
    v = __builtin_elementwise_fma(v, x2, (f2)(-2.05238f));
    v = __builtin_elementwise_fma(v, x2, (f2)(0.999861f));
    if (EDGE_X) v.x = fabsf(x.x) >= 2.0f ? 0.0f : v.x;
    if (EDGE_Y) v.y = fabsf(x.y) >= 2.0f ? 0.0f : v.y;
    return v;
}

// 4 adjacent lanes hold one BGR pixel each (p = B | G<<8 | R<<16); lanes 0..2 of the quad assemble the three
// dwords of the 12-byte group from their own pixel and their right neighbour's: bytes m..m+3 of {own, next}.
__device__ __forceinline__ uint32_t quad_pack_bgr(uint32_t p, int m) {
    const uint32_t q = (uint32_t)dpp_mov0<0x101>((int)p);   // row_shl:1 = the right neighbour's pixel (lane 3 of a quad ignores it)
    const uint32_t lo = p | (q << 24), hi = q >> 8;
    return __builtin_amdgcn_alignbyte(hi, lo, (uint32_t)m);
}

template <typename T, int MODE, int BORDER>
__device__ __forceinline__ void warp_pixel_global(const T* __restrict__ src, int w, int h, int stride, float Wx,
                                                  float Wy, float maxv, uint32_t out[3]) {
    float flx = floorf(Wx), fly = floorf(Wy);
    int ix = (int)flx, iy = (int)fly;
    float frx = Wx - flx, fry = Wy - fly;
    auto fetch = [&](int sx, int sy, int c) -> float {
        if (BORDER == 1) return (sx < 0 || sy < 0 || sx >= w || sy >= h) ? 0.0f : (float)src[(size_t)sy * stride + (size_t)sx * 3 + c];
        return (float)src[(size_t)clampi(sy, 0, h - 1) * stride + (size_t)clampi(sx, 0, w - 1) * 3 + c];
    };
    if (MODE == 0 || MODE == 2) {   // the rare tiles on this path keep the exact arithmetic in the fast mode too
        float wx[4], wy[4];
        lanczos_weights4(frx, wx);
        lanczos_weights4(fry, wy);
        float num[3] = {0.f, 0.f, 0.f}, den = 0.f;
#pragma unroll
        for (int ry = 0; ry < 4; ry++)
#pragma unroll
            for (int rx = 0; rx < 4; rx++) {
                float w2d = wx[rx] * wy[ry];
#pragma unroll
                for (int c = 0; c < 3; c++) num[c] = num[c] + w2d * fetch(ix + rx - 1, iy + ry - 1, c);
                den = den + w2d;
            }
#pragma unroll
        for (int c = 0; c < 3; c++) out[c] = store_u(num[c] / den, maxv);
    } else {
#pragma unroll
        for (int c = 0; c < 3; c++) {
            float top = lerpf(fetch(ix, iy, c), fetch(ix + 1, iy, c), frx);
            float bottom = lerpf(fetch(ix, iy + 1, c), fetch(ix + 1, iy + 1, c), frx);
            out[c] = store_u(lerpf(top, bottom, fry), maxv);
        }
    }
}

template <typename T, int MODE, int BORDER>
__global__ __launch_bounds__(256) void vs_k_bgr_warp_c3(const T* __restrict__ src, int w, int h, int src_stride,
                                                        const float4* __restrict__ params, T* __restrict__ dst,
                                                        int dst_stride, size_t src_fs, size_t dst_fs, int tiles_x,
                                                        int tiles_per_frame, int total_tiles, int chunk, float maxv,
                                                        vsk::Roi roi) {
    __shared__ f4 tile[WS_H * WS_W];                       // {B,G,R,1} per staged source pixel
    // XCD-aware tile order: workgroups are dealt round-robin over the 8 XCDs (each with its own L2), so
    // workgroup b works on logical tile (b % 8) * chunk + b / 8: every XCD walks one contiguous run of
    // tiles in raster order and the halo rows / columns shared by neighbouring tiles hit in its L2.
    const int logical = (int)(blockIdx.x & 7) * chunk + (int)(blockIdx.x >> 3);
    if (logical >= total_tiles) return;
    const int frame = logical / tiles_per_frame, tl = logical - frame * tiles_per_frame;
    const int tyi = tl / tiles_x, txi = tl - tyi * tiles_x;
    const float4 P = params[frame];
    src += (size_t)frame * src_fs;
    dst += (size_t)frame * dst_fs;
    const float A1 = 1.0f + P.x, B = P.y, TX = P.z, TY = P.w;
    // output pixel (x, y) of the window is pixel (x + roi.x, y + roi.y) of the full frame: the sampling position is
    // computed from the full-frame coordinate, so a window equals the same rows / columns cut out of the whole warp
    const int x0 = txi * WT_W, y0 = tyi * WT_H;
    const int x1 = min(x0 + WT_W, roi.w) - 1, y1 = min(y0 + WT_H, roi.h) - 1;

    // source footprint of the tile.  Wx = fl(fl(A1*x) - fl(B*y)) + TX is monotone in x and in y (rounding is
    // monotone), so its extremes over the tile sit at corners chosen by the signs of A1 and B: 4 evaluations.
    const float fx0 = (float)(x0 + roi.x), fx1 = (float)(x1 + roi.x), fy0 = (float)(y0 + roi.y), fy1 = (float)(y1 + roi.y);
    const float xa = A1 >= 0.f ? fx0 : fx1, xb = A1 >= 0.f ? fx1 : fx0;     // x minimising / maximising A1*x
    const float ya = B >= 0.f ? fy0 : fy1, yb = B >= 0.f ? fy1 : fy0;       // y minimising / maximising B*y
    const float mnx = A1 * xa - B * yb + TX, mxx = A1 * xb - B * ya + TX;
    const float xc = B >= 0.f ? fx0 : fx1, xd = B >= 0.f ? fx1 : fx0;       // x minimising / maximising B*x
    const float yc = A1 >= 0.f ? fy0 : fy1, yd = A1 >= 0.f ? fy1 : fy0;
    const float mny = B * xc + A1 * yc + TY, mxy = B * xd + A1 * yd + TY;
    bool fits = fabsf(mnx) < 1.0e6f && fabsf(mxx) < 1.0e6f && fabsf(mny) < 1.0e6f && fabsf(mxy) < 1.0e6f;
    int sx_lo = 0, sy_lo = 0;
    if (fits) {
        sx_lo = ((int)floorf(mnx) - 1) & ~3;               // first staged column: a multiple of 4 pixels (12 bytes)
        const int sx_hi = (int)floorf(mxx) + 2;
        sy_lo = (int)floorf(mny) - 1;
        const int sy_hi = (int)floorf(mxy) + 2;
        fits = (sx_hi - sx_lo + 1) <= WS_W && (sy_hi - sy_lo + 1) <= WS_H;
        if (fits) {
            // 4 source pixels (12 bytes, one aligned load) per work item, converted once, written as 4 float4
            const int rows = sy_hi - sy_lo + 1;
            const int groups = (sx_hi - sx_lo + 4) >> 2;              // <= 20
            const uint32_t inv_groups = 65536u / (uint32_t)groups + 1u;  // i / groups == (i * inv) >> 16 for i < 24*20
            for (int i = threadIdx.x; i < rows * groups; i += 256) {
                const int r = (int)(((uint32_t)i * inv_groups) >> 16), g = i - r * groups;
                const int sy = sy_lo + r, sx = sx_lo + 4 * g;
                f4* t = tile + r * WS_W + 4 * g;
                const bool row_in = sy >= 0 && sy < h;
                if (BORDER == 1 && !row_in) {
                    const f4 z = {0.f, 0.f, 0.f, 1.f};
                    t[0] = z; t[1] = z; t[2] = z; t[3] = z;
                    continue;
                }
                const T* row = src + (size_t)clampi(sy, 0, h - 1) * src_stride;
                if (sx >= 0 && sx + 3 < w && ((((uintptr_t)(row + sx * 3)) & 3) == 0)) {
                    if (sizeof(T) == 1) {
                        const u32x3 q = *(const u32x3*)(row + sx * 3);   // B0 G0 R0 B1 | G1 R1 B2 G2 | R2 B3 G3 R3
                        t[0] = f4{ub(q.x, 0), ub(q.x, 1), ub(q.x, 2), 1.f};
                        t[1] = f4{ub(q.x, 3), ub(q.y, 0), ub(q.y, 1), 1.f};
                        t[2] = f4{ub(q.y, 2), ub(q.y, 3), ub(q.z, 0), 1.f};
                        t[3] = f4{ub(q.z, 1), ub(q.z, 2), ub(q.z, 3), 1.f};
                    } else {
                        const u32x3 q0 = *(const u32x3*)(row + sx * 3), q1 = *(const u32x3*)(row + sx * 3 + 6);
                        // q0 = B0G0 R0B1 G1R1 ; q1 = B2G2 R2B3 G3R3 (16 bits each)
                        t[0] = f4{(float)(q0.x & 0xffffu), (float)(q0.x >> 16), (float)(q0.y & 0xffffu), 1.f};
                        t[1] = f4{(float)(q0.y >> 16), (float)(q0.z & 0xffffu), (float)(q0.z >> 16), 1.f};
                        t[2] = f4{(float)(q1.x & 0xffffu), (float)(q1.x >> 16), (float)(q1.y & 0xffffu), 1.f};
                        t[3] = f4{(float)(q1.y >> 16), (float)(q1.z & 0xffffu), (float)(q1.z >> 16), 1.f};
                    }
                } else {
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        const int px = sx + k;
                        if (BORDER == 1 && (px < 0 || px >= w)) {
                            t[k] = f4{0.f, 0.f, 0.f, 1.f};
                        } else {
                            const T* q = row + clampi(px, 0, w - 1) * 3;
                            t[k] = f4{(float)q[0], (float)q[1], (float)q[2], 1.f};
                        }
                    }
                }
            }
        }
    }
    __syncthreads();

    const int lx = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int x = x0 + lx;
    const float fx = (float)(x + roi.x);
    const float A1x = A1 * fx, Bx = B * fx;
    // a quad of lanes owns 4 pixels = 12 output bytes; it stores them as 3 aligned dwords when the whole
    // quad is inside the row and the row is dword aligned, else byte by byte
    const int m = lx & 3;
    const bool quad_in = (x | 3) < roi.w;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int y = y0 + wv * 4 + k;
        if (y >= roi.h) break;                           // wave-uniform
        uint32_t o[3] = {0, 0, 0};
        if (x < roi.w) {
            const float fy = (float)(y + roi.y);
            const float Wx = A1x - B * fy + TX;          // generators.cpp:141
            const float Wy = Bx + A1 * fy + TY;          // generators.cpp:142
            if (!fits) {
                warp_pixel_global<T, MODE, BORDER>(src, w, h, src_stride, Wx, Wy, maxv, o);
            } else {
                const float flx = floorf(Wx), fly = floorf(Wy);
                const int ix = (int)flx, iy = (int)fly;
                const f2 fr = {Wx - flx, Wy - fly};
                if (MODE == 0) {
                    // the four live taps of the 5-tap window (tap 0 has weight exactly 0): four packed Horner chains,
                    // each holding two adjacent taps of one axis, so the tap products below are packed too
                    const f2 frx = {fr.x, fr.x}, fry = {fr.y, fr.y};
                    const f2 wx01 = lanczos2_pk<true, false>(f2{-1.0f, 0.0f} - frx), wx23 = lanczos2_pk<false, true>(f2{1.0f, 2.0f} - frx);
                    const f2 wy01 = lanczos2_pk<true, false>(f2{-1.0f, 0.0f} - fry), wy23 = lanczos2_pk<false, true>(f2{1.0f, 2.0f} - fry);
                    const float wy[4] = {wy01.x, wy01.y, wy23.x, wy23.y};
                    const f4* t = tile + (iy - 1 - sy_lo) * WS_W + (ix - 1 - sx_lo);
                    f2 nbg = {0.f, 0.f}, nrd = {0.f, 0.f};
#pragma unroll
                    for (int ry = 0; ry < 4; ry++) {
                        const f2 wyy = {wy[ry], wy[ry]};
                        const f2 p01 = wx01 * wyy, p23 = wx23 * wyy;              // w2d = wx[rx] * wy[ry]
                        const float w2d[4] = {p01.x, p01.y, p23.x, p23.y};
#pragma unroll
                        for (int rx = 0; rx < 4; rx++) {
                            const f4 v = t[ry * WS_W + rx];
                            const f2 ww = {w2d[rx], w2d[rx]};
                            nbg = nbg + ww * f2{v.x, v.y};       // num_B, num_G
                            nrd = nrd + ww * f2{v.z, v.w};       // num_R, den (v.w == 1: den + w2d*1 == den + w2d)
                        }
                    }
                    float qb, qg, qr;
                    div3_exact(nbg.x, nbg.y, nrd.x, nrd.y, qb, qg, qr);
                    o[0] = store_u(qb, maxv);
                    o[1] = store_u(qg, maxv);
                    o[2] = store_u(qr, maxv);
                } else if (MODE == 2) {
                    // VS_WARP_LANCZOS2_FAST: fused multiply-adds for the weights and the tap sums, one refined reciprocal
                    const f2 frx = {fr.x, fr.x}, fry = {fr.y, fr.y};
                    const f2 wx01 = lanczos2_pk_fma<true, false>(f2{-1.0f, 0.0f} - frx), wx23 = lanczos2_pk_fma<false, true>(f2{1.0f, 2.0f} - frx);
                    const f2 wy01 = lanczos2_pk_fma<true, false>(f2{-1.0f, 0.0f} - fry), wy23 = lanczos2_pk_fma<false, true>(f2{1.0f, 2.0f} - fry);
                    const float wy[4] = {wy01.x, wy01.y, wy23.x, wy23.y};
                    const f4* t = tile + (iy - 1 - sy_lo) * WS_W + (ix - 1 - sx_lo);
                    f2 nbg = {0.f, 0.f}, nrd = {0.f, 0.f};
#pragma unroll
                    for (int ry = 0; ry < 4; ry++) {
                        const f2 wyy = {wy[ry], wy[ry]};
                        const f2 p01 = wx01 * wyy, p23 = wx23 * wyy;
                        const float w2d[4] = {p01.x, p01.y, p23.x, p23.y};
#pragma unroll
                        for (int rx = 0; rx < 4; rx++) {
                            const f4 v = t[ry * WS_W + rx];
                            const f2 ww = {w2d[rx], w2d[rx]};
                            nbg = __builtin_elementwise_fma(ww, f2{v.x, v.y}, nbg);
                            nrd = __builtin_elementwise_fma(ww, f2{v.z, v.w}, nrd);
                        }
                    }
                    float r = __builtin_amdgcn_rcpf(nrd.y);
                    r = __builtin_fmaf(__builtin_fmaf(-nrd.y, r, 1.0f), r, r);       // one Newton step: ~0.5 ULP reciprocal
                    o[0] = store_u(nbg.x * r, maxv);
                    o[1] = store_u(nbg.y * r, maxv);
                    o[2] = store_u(nrd.x * r, maxv);
                } else {
                    const f4* t = tile + (iy - sy_lo) * WS_W + (ix - sx_lo);
                    const f4 a0 = t[0], a1 = t[1], b0 = t[WS_W], b1 = t[WS_W + 1];
                    const f2 tx = {fr.x, fr.x}, ty = {fr.y, fr.y};
                    const f2 otx = 1.0f - tx, oty = 1.0f - ty;
                    // lerp(a,b,t) = a*(1-t) + b*t (generators.cpp:161-163), {B,G} and {R,-} pairs
                    const f2 top_bg = f2{a0.x, a0.y} * otx + f2{a1.x, a1.y} * tx;
                    const f2 bot_bg = f2{b0.x, b0.y} * otx + f2{b1.x, b1.y} * tx;
                    const f2 top_r = f2{a0.z, a0.z} * otx + f2{a1.z, a1.z} * tx;
                    const f2 bot_r = f2{b0.z, b0.z} * otx + f2{b1.z, b1.z} * tx;
                    const f2 bg = top_bg * oty + bot_bg * ty;
                    const f2 rr = top_r * oty + bot_r * ty;
                    o[0] = store_u(bg.x, maxv);
                    o[1] = store_u(bg.y, maxv);
                    o[2] = store_u(rr.x, maxv);
                }
            }
        }
        T* orow = dst + (size_t)y * dst_stride;
        if (sizeof(T) == 1) {
            const uint32_t p = o[0] | (o[1] << 8) | (o[2] << 16);
            const uint32_t d = quad_pack_bgr(p, m);          // every lane of the wave takes part in the shuffle
            if (quad_in && ((((uintptr_t)orow) & 3) == 0)) {
                if (m < 3) *(uint32_t*)((uint8_t*)orow + (size_t)(x & ~3) * 3 + 4 * m) = d;
            } else if (x < roi.w) {
                orow[(size_t)x * 3] = (T)o[0];
                orow[(size_t)x * 3 + 1] = (T)o[1];
                orow[(size_t)x * 3 + 2] = (T)o[2];
            }
        } else {
            uint32_t d0, d1;
            pair_pack_bgr16(o, x & 1, d0, d1);
            const bool pair_in = (x | 1) < roi.w;
            if (pair_in && ((((uintptr_t)orow) & 3) == 0)) {
                uint32_t* q = (uint32_t*)(orow + (size_t)(x & ~1) * 3);   // 12 bytes per pixel pair
                if (x & 1) q[2] = d0;
                else { q[0] = d0; q[1] = d1; }
            } else if (x < roi.w) {
                orow[(size_t)x * 3] = (T)o[0];
                orow[(size_t)x * 3 + 1] = (T)o[1];
                orow[(size_t)x * 3 + 2] = (T)o[2];
            }
        }
    }
}

}  // namespace

namespace vsk {

template <typename T>
static hipError_t launch_c3(const T* src, int w, int h, int src_stride, const float4* params_dev, int mode, int border, T* dst,
                            int dst_stride, int n_frames, size_t src_fs, size_t dst_fs, float maxv, Roi roi, hipStream_t s) {
    const int tiles_x = (roi.w + WT_W - 1) / WT_W, tiles_y = (roi.h + WT_H - 1) / WT_H;
    const long long total = (long long)tiles_x * tiles_y * n_frames;
    if (total > 0x3fffffffLL) return hipErrorNotSupported;
    const int chunk = (int)((total + 7) / 8);
    dim3 grid((unsigned)(chunk * 8)), block(256);
#define VS_LAUNCH(M, Bd) \
    hipLaunchKernelGGL((vs_k_bgr_warp_c3<T, M, Bd>), grid, block, 0, s, src, w, h, src_stride, params_dev, dst, dst_stride, src_fs, dst_fs, \
                       tiles_x, tiles_x * tiles_y, (int)total, chunk, maxv, roi)
    if (mode == 0 && border == 0) VS_LAUNCH(0, 0);
    else if (mode == 0) VS_LAUNCH(0, 1);
    else if (mode == 2 && border == 0) VS_LAUNCH(2, 0);
    else if (mode == 2) VS_LAUNCH(2, 1);
    else if (border == 0) VS_LAUNCH(1, 0);
    else VS_LAUNCH(1, 1);
#undef VS_LAUNCH
    return hipGetLastError();
}

hipError_t bgr_warp_c3(const void* src, int w, int h, int src_stride, int bits, const float4* params_dev, int mode, int border,
                       int max_value, void* dst, int dst_stride, int n_frames, size_t src_fs, size_t dst_fs, Roi roi, hipStream_t s) {
    if (bits == 8)
        return launch_c3<uint8_t>((const uint8_t*)src, w, h, src_stride, params_dev, mode, border, (uint8_t*)dst, dst_stride, n_frames,
                                  src_fs, dst_fs, (float)max_value, roi, s);
    return launch_c3<uint16_t>((const uint16_t*)src, w, h, src_stride, params_dev, mode, border, (uint16_t*)dst, dst_stride, n_frames,
                               src_fs, dst_fs, (float)max_value, roi, s);
}

}  // namespace vsk
