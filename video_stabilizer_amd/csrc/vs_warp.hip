// vs_warp.hip -- tuned bgr_image_warp for interleaved 3-channel u8 / u16 frames (the 1080p / 4K roofline kernel).
//
// Sampler semantics = vs_k_bgr_warp_generic (vs_kernels.hip), i.e. the reference's Lanczos2 sampler
// (generators.cpp:672-697: 5x5 window, polynomial weights, rx inner / ry outer, num and den summed
// separately, one divide) or image_warp's bilinear (generators.cpp:148-163), evaluated per channel at
// image_warp's coordinates (generators.cpp:141-142).  VS_WARP_LANCZOS2 / VS_WARP_BILINEAR results are
// bit-identical to the generic kernel and to the CPU oracle: same fp32 operations in the same order, no
// FMA contraction.  VS_WARP_LANCZOS2_FAST is the contracted form of the same sampler (vs_device.hpp), bit-identical to
// the oracle's VSO_WARP_LANCZOS2_CONTRACTED.
//
// Structure (one 256-thread workgroup = one 64x16 output tile of one frame; 64x32 and 8 rows per wave for bilinear on 8-bit frames):
//   1. The similarity is affine, so the tile's source footprint is the bounding box of its four
//      corners (fp32 rounding is monotonic, so the corners bound every pixel exactly).  For the
//      near-identity transforms of stabilisation that is ~67x19 pixels.
//   2. The footprint (+ the Lanczos halo) is read from HBM once with aligned 12-byte loads (4 pixels),
//      converted to float ONCE, and parked in LDS as one float4 {B,G,R,1} per source pixel.  Clamp /
//      constant-0 borders are resolved here, so the inner loop has no clamps and no global loads.
//   3. Lane = output column, each wave owns 4 output rows and computes them in ONE straight-line block
//      (no branch between rows: lanes / rows outside the window sample a clamped coordinate and are masked
//      at the store), so the scheduler interleaves the four rows' dependency chains.  One tap = one
//      ds_read_b128 at an immediate offset from a single address register; neighbouring lanes read
//      neighbouring 16-byte slots, so the reads are conflict-free.
//   4. A quad of lanes assembles its 12 output bytes with one DPP move + one v_perm_b32 per row and stores
//      3 aligned dwords (192 contiguous bytes per wave-row).
// Tiles whose footprint does not fit the LDS window (large rotation / zoom) take the generic
// global-memory path inside the same kernel, so every transform is supported.
//
// Cost model (tools/ubench_issue.hip, profiles/r04_ubench_issue.txt): the kernel is VALU-issue-bound, not
// HBM-bound.  Issue cost per instruction per SIMD on gfx950 at settled clocks (in-kernel clock 2.37-2.40 GHz, 4 waves per SIMD,
// wall time x in-kernel clock / instructions): v_fma_f32 2.8 cycles (4.3 with 2 waves), v_mul / v_add_f32 2.5, v_pk_{fma,mul,add}_f32
// 4.4-4.8 (= 2.2-2.4 per lane-operation: packed wins for the exact mode's separately rounded mul -> add pairs, 3.1 scalar), a mix
// of scalar and packed fp32 4.3-4.6 per instruction (worse than either pure stream), conversions / v_perm / v_med3 / v_floor / DPP
// 4.2-4.5, v_rcp 8.2, v_fma_mix_f32 4.4 (an fp16 tile buys nothing); ds_read_b128 4.1 cycles per CU (16.5 per SIMD when all
// four SIMDs read).  The exact mode needs 248 separately rounded fp32 operations per pixel (112 Horner, 16 weight products, 96 tap
// multiply / adds, 16 den adds, 8 selects) = 128 packed instructions = ~590 cycles per 64 pixels before any overhead.
#include "vs_kernels.hpp"
#include <algorithm>
#include <cmath>
#include <vector>
#include "vs_device.hpp"

using namespace vsd;

namespace {

// tuning knobs (tools/build_variant.sh builds the library with other values to compare them on the GPU)
#ifndef VS_WARP_WS_EXTRA
#define VS_WARP_WS_EXTRA 8               // staged rows beyond the tile's own in the float-tile (Lanczos2) kernels: 8 -> 24 rows, 31 KB of LDS, 5 workgroups per CU (experiment: 4 ->
                                         // 20 rows, 25.9 KB, 6 workgroups with VS_WARP_FAST_MINWAVES=6; rotations beyond ~0.9 degrees then leave the window: profiles/r06_warp_sep_occupancy.txt)
#endif
#ifndef VS_WARP_FAST_MINWAVES
#define VS_WARP_FAST_MINWAVES 4          // __launch_bounds__ waves per SIMD of the fast Lanczos2 kernels
#endif
#ifndef VS_WARP_FAST_SCHED
#define VS_WARP_FAST_SCHED 1             // scheduling fences in the fast kernels: 0 none, 1 per row pair, 2 per source row
#endif
#ifndef VS_WARP_EXACT_MINWAVES
#define VS_WARP_EXACT_MINWAVES 4
#endif
#ifndef VS_WARP_COORDS_FIRST
#define VS_WARP_COORDS_FIRST 0           // 1: a scheduling fence between the rows' position arithmetic and the sampler blocks
#endif
#ifndef VS_WARP_TILE_H
#define VS_WARP_TILE_H 16                // output rows per workgroup (4 waves: VS_WARP_TILE_H / 4 rows per wave)
#endif

#ifndef VS_WARP_WHATIF
#define VS_WARP_WHATIF 0                 // analysis builds only (wrong results): 1 one LDS read per pixel instead of 16, 2 no fill, 4 no division, 8 no store, 16 no weight chains, 32 loads hit the same lines, 256 every tile takes the rim fill (right results)
#endif
#ifndef VS_WARP_TILES_PER_WG
#define VS_WARP_TILES_PER_WG 1           // consecutive tiles of its XCD's run a workgroup walks; > 1: the next tile's source loads are in flight during the current tile's sampler blocks
#endif
#ifndef VS_WARP_TILES_PER_WG_BILINEAR
#define VS_WARP_TILES_PER_WG_BILINEAR VS_WARP_TILES_PER_WG     // the same for the bilinear mode (measured twice, float tile and raw tile: equal or slower, profiles/r04_ab_warp_bilinear.md)
#endif
#ifndef VS_WARP_ROW_BLOCK
#define VS_WARP_ROW_BLOCK 4              // rows a wave computes in one straight-line block (even); a wave's rows are walked in such blocks
#endif
#ifndef VS_WARP_FAST_PIPE
#define VS_WARP_FAST_PIPE 1              // contracted mode: 1 = the software-pipelined sampler (fast_rows_pipelined), 0 = row pairs (fast_pair); bit-identical
#endif
#ifndef VS_WARP_PIPE_AHEAD
#define VS_WARP_PIPE_AHEAD 6             // ... tap reads in flight ahead of the tap being consumed (< 8: the ring has eight slots)
#endif
#ifndef VS_WARP_PIPE_AHEAD_COMPACT
#define VS_WARP_PIPE_AHEAD_COMPACT 2     // ... in the COMPACT instantiation (six waves per SIMD: 2 -> 72 VGPRs, 3 -> 75, 4 and more spill at the 85-register cap)
#endif
constexpr int WT_W = 64, WT_H = VS_WARP_TILE_H;      // output tile
constexpr int RPW = WT_H / 4;            // output rows per wave
constexpr int RB = VS_WARP_ROW_BLOCK < RPW ? VS_WARP_ROW_BLOCK : RPW;
static_assert(RPW % RB == 0 && RB % 2 == 0, "rows per wave: a whole number of row blocks, rows in pairs");
constexpr int WS_W = 80;                 // staged source pixels per row (multiple of 4)
// (staged source rows: the tile height + 8, per kernel -- WS_H inside vs_k_bgr_warp_c3, CV_WS_H / CV16_WS_H for the fixed-point bilinear kernels)
// A staged row holds WS_W pixels but is WS_RS slots long: 81 slots = 1296 bytes = 16 (mod 128), so the fill's ds_write_b128
// (8-lane groups = 4 rows x 2 column groups, 64-byte column-group stride) touch every bank once.
constexpr int WS_RS = WS_W + 1;
// Bilinear mode on 8-bit frames: the tile holds the source BYTES, one dword {B,G,R,0} per pixel (8.4 KB instead of 31 KB: 8 workgroups
// per CU instead of 5), and the sampler converts its four taps itself (v_cvt_f32_ubyteN: the byte select is free).  The float tile's fill
// (load -> 16 conversions -> 4 x ds_write_b128 per item -> barrier) and its share of the memory traffic were 15 of the kernel's 21.8 us per
// 4K frame; with the byte tile the kernel is VALU-bound by count like its Lanczos siblings (88 vector instructions per pixel, ~70 of them
// the sampler: counters and time stamps in profiles/r04_ab_warp_bilinear.md).
// Row pitch 88 dwords = 24 (mod 64): the fill's 8-lane ds_write_b128 groups (4 rows x 2 column groups) touch every bank once.
#ifndef VS_WARP_BILINEAR_U8_TILE
#define VS_WARP_BILINEAR_U8_TILE 1
#endif
constexpr int WS_RS8 = WS_W + 8;
// ... and its output tile is taller: the per-workgroup prologue (tile geometry, fill set-up, ~200 instructions) is paid once per tile, and
// with a 14 KB tile 32 rows still leave 8 workgroups per CU.  4K: 17.3 -> 15.6 us per frame (64 rows: 15.1, but a 1080p frame is then only
// 510 tiles); the float-tile kernels lose occupancy instead (contracted Lanczos2 38.2 -> 40.3 us).  profiles/r04_ab_warp_bilinear.md.
#ifndef VS_WARP_TILE_H_BILINEAR_U8
#define VS_WARP_TILE_H_BILINEAR_U8 32
#endif
// Bilinear on 16-bit containers (10 / 12 / 16-bit frames) the same way: the tile holds the source words, 8 bytes {B | G << 16, R} per
// pixel (17 KB for a 64 x 16 output tile instead of 31 KB of float4), two pixels per ds_write_b128 in the fill, one ds_read2_b64 per window
// row in the sampler, conversions by SDWA word select.  Same 88-pixel pitch: 176 dwords = 48 (mod 64), conflict-free like the byte tile.
#ifndef VS_WARP_BILINEAR_U16_TILE
#define VS_WARP_BILINEAR_U16_TILE 1
#endif
#ifndef VS_WARP_CV_ROW_FILL
#define VS_WARP_CV_ROW_FILL 1            // interior fill of every byte / word tile: a wave slot = three staged rows x twenty column groups, the slots twelve rows apart (see vs_k_bgr_warp_cv_c3)
#endif
#ifndef VS_WARP_TILE_H_BILINEAR_U16
#define VS_WARP_TILE_H_BILINEAR_U16 16
#endif
constexpr bool raw_tile_of(int bits, int mode) { return mode == 1 && (bits == 8 ? VS_WARP_BILINEAR_U8_TILE : VS_WARP_BILINEAR_U16_TILE) != 0; }
constexpr int tile_h_of(int bits, int mode) {
    return !raw_tile_of(bits, mode) ? VS_WARP_TILE_H : (bits == 8 ? VS_WARP_TILE_H_BILINEAR_U8 : VS_WARP_TILE_H_BILINEAR_U16);
}
static_assert(VS_WARP_TILE_H_BILINEAR_U16 % 8 == 0, "rows per wave in pairs");
static_assert(VS_WARP_TILE_H_BILINEAR_U8 % 8 == 0 && (VS_WARP_TILE_H_BILINEAR_U8 + 8) / 4 * (WS_W / 4) < 1024, "fill_item's p / 20 is exact below 1024");

// analysis build (tools/warp_stamps.py): every wave of the first STAMP_WGS workgroups of a launch leaves eight s_memtime stamps
// (entry, geometry done, loads issued, loads landed, tile written, barrier passed, rows stored, stores drained) and its HW_ID / XCC_ID
#ifndef VS_WARP_STAMPS
#define VS_WARP_STAMPS 0
#endif
#if VS_WARP_STAMPS
constexpr int STAMP_WGS = 8192, STAMP_N = 10;
__device__ unsigned long long g_warp_stamps[STAMP_WGS * 4 * STAMP_N];
#define VS_STAMP(i) do { __builtin_amdgcn_sched_barrier(0); stamp[i] = __builtin_readcyclecounter(); __builtin_amdgcn_sched_barrier(0); } while (0)
#define VS_STAMP_DRAIN() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#else
#define VS_STAMP(i) ((void)0)
#define VS_STAMP_DRAIN() ((void)0)
#endif

// the aligned output stores: plain, or (experiment) non-temporal -- the output is written once and not read by this kernel
#ifndef VS_WARP_NT_STORE
#define VS_WARP_NT_STORE 0
#endif
#if VS_WARP_NT_STORE
#define VS_STORE32(p, v) __builtin_nontemporal_store((uint32_t)(v), (uint32_t*)(p))
#else
#define VS_STORE32(p, v) (*(uint32_t*)(p) = (v))
#endif

typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x3 __attribute__((ext_vector_type(3)));
typedef const __attribute__((address_space(3))) f4* lds_f4;

__device__ __forceinline__ float lerpf(float a, float b, float t) { return a * (1.0f - t) + b * t; }

__device__ __forceinline__ float ub(uint32_t q, int k) { return (float)((q >> (8 * k)) & 0xffu); }

// build rule: round half up, saturate = clamp(floor(v + 0.5), 0, maxv).  maxv is an integer, so clamping first and
// then truncating (float -> u32 conversion truncates, = floor for non-negative values) gives the same integer.
__device__ __forceinline__ uint32_t store_u(float v, float maxv) {
    return (uint32_t)__builtin_amdgcn_fmed3f(v + 0.5f, 0.0f, maxv);
}

// Three correctly rounded quotients over one denominator.  This is hipcc's own fp32 division expansion
// (v_div_scale, v_rcp, two Newton steps on the reciprocal-quotient pair, v_div_fmas, v_div_fixup) with the
// parts that are no-ops here removed: den is a sum of Lanczos weights (0.99..1.05) and the numerators are
// bounded byte sums, so no operand scaling and no special-case fix-up ever applies; what is left is the
// same fma sequence, and the reciprocal refinement is shared by the three channels (18 instructions
// instead of 33).  Outside the safe range the caller falls back to operator/ (wave-uniform branch).
__device__ __forceinline__ void div3_core(float n0, float n1, float n2, float den, float q[3]) {
    float r = __builtin_amdgcn_rcpf(den);
    const float e = __builtin_fmaf(-den, r, 1.0f);
    r = __builtin_fmaf(e, r, r);
    float a, t;
    a = n0 * r; t = __builtin_fmaf(-den, a, n0); a = __builtin_fmaf(t, r, a); t = __builtin_fmaf(-den, a, n0); q[0] = __builtin_fmaf(t, r, a);
    a = n1 * r; t = __builtin_fmaf(-den, a, n1); a = __builtin_fmaf(t, r, a); t = __builtin_fmaf(-den, a, n1); q[1] = __builtin_fmaf(t, r, a);
    a = n2 * r; t = __builtin_fmaf(-den, a, n2); a = __builtin_fmaf(t, r, a); t = __builtin_fmaf(-den, a, n2); q[2] = __builtin_fmaf(t, r, a);
}

// Exact Lanczos2 of two output pixels (rows k, k+1 of one lane) from the LDS tile; t[j] = staged pixel (iy-1, ix-1) of
// pixel j.  Per pixel: the four live taps of the 5-tap window per axis (tap 0 has weight exactly 0) as four packed Horner
// chains, each holding two adjacent taps of one axis, so the tap products are packed too; {B,G} and {R,den} accumulate as
// pairs (the tile's trailing 1.0 makes den += w2d part of the same instruction; w2d*1.0 is exact).  Every component sees
// exactly the reference's sequence of roundings (generators.cpp:31-47, 684-697).  num[j] = {numB, numG, numR, den}.
__device__ __forceinline__ void exact_pair(const lds_f4 t[2], const f2 fr[2], float num[2][4]) {
    // chain c of pixel j: 0 = x taps {1,2}, 1 = x taps {3,4}, 2 = y taps {1,2}, 3 = y taps {3,4}
    f2 x[2][4], x2[2][4], v[2][4], mm[2][4];
#pragma unroll
    for (int j = 0; j < 2; j++) {
        const f2 frx = {fr[j].x, fr[j].x}, fry = {fr[j].y, fr[j].y};
        x[j][0] = f2{-1.0f, 0.0f} - frx; x[j][1] = f2{1.0f, 2.0f} - frx;
        x[j][2] = f2{-1.0f, 0.0f} - fry; x[j][3] = f2{1.0f, 2.0f} - fry;
    }
#pragma unroll
    for (int c = 0; c < 4; c++)
#pragma unroll
        for (int j = 0; j < 2; j++) { x2[j][c] = x[j][c] * x[j][c]; v[j][c] = f2{0.000858519f, 0.000858519f}; }
    const float C[6] = {-0.0158853f, 0.128693f, -0.583468f, 1.52229f, -2.05238f, 0.999861f};
#pragma unroll
    for (int s = 0; s < ((VS_WARP_WHATIF & 16) ? 1 : 6); s++) {
#pragma unroll
        for (int c = 0; c < 4; c++)
#pragma unroll
            for (int j = 0; j < 2; j++) mm[j][c] = v[j][c] * x2[j][c];
#pragma unroll
        for (int c = 0; c < 4; c++)
#pragma unroll
            for (int j = 0; j < 2; j++) v[j][c] = C[s] + mm[j][c];
    }
    // |x| >= 2 can only happen for taps 1 (-1-frac) and 4 (2-frac)
#pragma unroll
    for (int j = 0; j < 2; j++) {
        v[j][0].x = fabsf(x[j][0].x) >= 2.0f ? 0.0f : v[j][0].x;
        v[j][1].y = fabsf(x[j][1].y) >= 2.0f ? 0.0f : v[j][1].y;
        v[j][2].x = fabsf(x[j][2].x) >= 2.0f ? 0.0f : v[j][2].x;
        v[j][3].y = fabsf(x[j][3].y) >= 2.0f ? 0.0f : v[j][3].y;
    }
    f2 nbg[2] = {f2{0.f, 0.f}, f2{0.f, 0.f}}, nrd[2] = {f2{0.f, 0.f}, f2{0.f, 0.f}};
    const f4 one_tap[2] = {t[0][0], t[1][0]};
#pragma unroll
    for (int ry = 0; ry < 4; ry++) {
        f2 p01[2], p23[2];
#pragma unroll
        for (int j = 0; j < 2; j++) {
            const float wy = (ry & 1) ? v[j][2 + (ry >> 1)].y : v[j][2 + (ry >> 1)].x;
            const f2 wyy = {wy, wy};
            p01[j] = v[j][0] * wyy; p23[j] = v[j][1] * wyy;               // w2d = wx[rx] * wy[ry]
        }
#pragma unroll
        for (int rx = 0; rx < 4; rx++) {
            f2 mbg[2], mrd[2];
#pragma unroll
            for (int j = 0; j < 2; j++) {
                const f4 val = (VS_WARP_WHATIF & 1) ? one_tap[j] : t[j][ry * WS_RS + rx];
                const f2 pp = rx < 2 ? p01[j] : p23[j];
                const float w2d = (rx & 1) ? pp.y : pp.x;
                const f2 ww = {w2d, w2d};
                mbg[j] = ww * f2{val.x, val.y};
                mrd[j] = ww * f2{val.z, val.w};
            }
#pragma unroll
            for (int j = 0; j < 2; j++) {
                nbg[j] = nbg[j] + mbg[j];                  // num_B, num_G
                nrd[j] = nrd[j] + mrd[j];                  // num_R, den (val.w == 1: den + w2d*1 == den + w2d)
            }
        }
    }
#pragma unroll
    for (int j = 0; j < 2; j++) { num[j][0] = nbg[j].x; num[j][1] = nbg[j].y; num[j][2] = nrd[j].x; num[j][3] = nrd[j].y; }
}

// VS_WARP_LANCZOS2_FAST of two output pixels: the contracted sampler of vs_device.hpp (oracle twin VSO_WARP_LANCZOS2_CONTRACTED)
// written out for the LDS tile -- Horner steps as single fmas, w2d = wx * wy a rounded product, num = fma(w2d, val, num) per
// channel and den = den + w2d in the reference's tap order (rx inner, ry outer).  Everything is scalar fp32: a packed fp32
// instruction next to scalar ones costs more than either pure stream (profiles/r04_ubench_issue.txt: fma 2.8 cycles, pk_fma 4.8,
// a 3:1 mix 4.6 per instruction).  num[j] = {numB, numG, numR, den}; the caller divides.  (Kept for VS_WARP_FAST_PIPE=0 and the
// what-if builds; the shipped contracted kernel runs fast_rows_pipelined below.)
__device__ __forceinline__ void fast_pair(const lds_f4 t[2], const f2 fr[2], float num[2][4]) {
    // weight chain c of pixel j: 0..3 = x taps 1..4, 4..7 = y taps 1..4
    float x[2][8], x2[2][8], v[2][8];
#pragma unroll
    for (int j = 0; j < 2; j++) {
        x[j][0] = -1.0f - fr[j].x; x[j][1] = 0.0f - fr[j].x; x[j][2] = 1.0f - fr[j].x; x[j][3] = 2.0f - fr[j].x;
        x[j][4] = -1.0f - fr[j].y; x[j][5] = 0.0f - fr[j].y; x[j][6] = 1.0f - fr[j].y; x[j][7] = 2.0f - fr[j].y;
    }
#pragma unroll
    for (int c = 0; c < 8; c++)
#pragma unroll
        for (int j = 0; j < 2; j++) { x2[j][c] = x[j][c] * x[j][c]; v[j][c] = 0.000858519f; }
    const float C[6] = {-0.0158853f, 0.128693f, -0.583468f, 1.52229f, -2.05238f, 0.999861f};
#pragma unroll
    for (int s = 0; s < ((VS_WARP_WHATIF & 16) ? 1 : 6); s++)
#pragma unroll
        for (int c = 0; c < 8; c++)
#pragma unroll
            for (int j = 0; j < 2; j++) v[j][c] = __builtin_fmaf(v[j][c], x2[j][c], C[s]);
#pragma unroll
    for (int j = 0; j < 2; j++) {
        // |x| >= 2 can only happen for taps 1 (-1-frac) and 4 (2-frac)
        v[j][0] = fabsf(x[j][0]) >= 2.0f ? 0.0f : v[j][0];
        v[j][3] = fabsf(x[j][3]) >= 2.0f ? 0.0f : v[j][3];
        v[j][4] = fabsf(x[j][4]) >= 2.0f ? 0.0f : v[j][4];
        v[j][7] = fabsf(x[j][7]) >= 2.0f ? 0.0f : v[j][7];
        num[j][0] = 0.0f; num[j][1] = 0.0f; num[j][2] = 0.0f; num[j][3] = 0.0f;
    }
    const f4 one_tap[2] = {t[0][0], t[1][0]};
#pragma unroll
    for (int ry = 0; ry < 4; ry++) {
#pragma unroll
        for (int rx = 0; rx < 4; rx++) {
#pragma unroll
            for (int j = 0; j < 2; j++) {
                const f4 val = (VS_WARP_WHATIF & 1) ? one_tap[j] : t[j][ry * WS_RS + rx];
                const float w2d = v[j][rx] * v[j][4 + ry];
                num[j][0] = __builtin_fmaf(w2d, val.x, num[j][0]);
                num[j][1] = __builtin_fmaf(w2d, val.y, num[j][1]);
                num[j][2] = __builtin_fmaf(w2d, val.z, num[j][2]);
                num[j][3] = num[j][3] + w2d;
            }
        }
        if (VS_WARP_FAST_SCHED >= 2) __builtin_amdgcn_sched_barrier(0);
    }
}

// VS_WARP_FAST_PIPE: the same contracted arithmetic for the RB = 4 pixels of a lane (rows k = 0..3), software-pipelined so that a wave's
// LDS reads are spread evenly over ALL of its vector work instead of arriving in bursts.  In fast_pair a pixel pair is 150 vector
// instructions of weight chains with no LDS traffic followed by 32 ds_read_b128 inside 160 instructions of tap arithmetic; the four
// waves of a workgroup leave the fill barrier together, so their tap phases coincide and one workgroup alone asks for ~80 % of the
// CU's LDS read rate during them (profiles/r04_ubench_issue.txt: a ds_read_b128 per 4 fmas is LDS-bound, 81 cycles per 4 taps
// against 44 for the fmas).  Here stage k runs three things interleaved, slice by slice (16 slices, one tap each): the taps of pixel k,
// the weight chains of pixel k + 1 and the division / store conversion of pixel k - 1 -- one ds_read_b128 per ~11 vector
// instructions throughout, issued VS_WARP_PIPE_AHEAD taps ahead of their use through a ring of eight float4 registers.  Per pixel the
// operations and their order are fast_pair's (weights: Horner fmas per chain; taps rx inner, ry outer; num = fma(w2d, val, num),
// den = den + w2d; div3_core; store_u), so the result is bit-identical: the pipelining only decides WHEN an instruction issues.
// Scheduling fences between the slices keep the compiler from undoing the interleave.
// (Round 5 tried to leave out the four |x| >= 2 selects per pixel in blocks whose fractions cannot reach them -- one min3 / max3 chain over
// the block's fractions and a ballot; they fire for frac == 0 and for a fraction rounded up to 1 only, ~8 % of the blocks at 4K.  Both ways
// to do it lost: two copies of this function behind a wave-uniform branch took the kernel from 78 to 128 VGPRs with 9 spilled (42.1 us per
// 4K frame instead of 38.3), scalar branches over the select slices cut the straight-line block into pieces the register allocator
// handles far worse (135 spilled VGPRs).  profiles/r05_warp_sep.md.)
template <int NPX, int AHEAD = VS_WARP_PIPE_AHEAD, int RS = WS_RS>
__device__ __forceinline__ void fast_rows_pipelined(const lds_f4 (&t)[NPX], const f2 (&fr)[NPX], float maxv, float (&num)[NPX][4],
                                                    uint32_t (&o)[NPX][3], bool& all_ok) {
    constexpr int NV = 8;
    static_assert(AHEAD >= 1 && AHEAD < NV && 16 % NV == 0, "ring of eight tap registers");
    const float C[6] = {-0.0158853f, 0.128693f, -0.583468f, 1.52229f, -2.05238f, 0.999861f};
    f4 vals[NV];
    float wgt[2][8];                 // finished weights: wgt[k & 1] belongs to pixel k (x taps 1..4, then y taps 1..4)
    float wx_[8], wx2[8];            // the chains under construction: arguments, their squares (values in wgt[(k + 1) & 1])
    float r_ = 0.f, q_[3] = {0.f, 0.f, 0.f};     // division state of pixel k - 1
    // one slice of the weight chains of pixel kk (72 instructions in 16 slices)
    auto w_slice = [&](int j, int kk) {
        float (&v)[8] = wgt[kk & 1];
        const float f = (j == 0) ? fr[kk].x : fr[kk].y;
        if (j < 2) {
            const int b = 4 * j;
            wx_[b + 0] = -1.0f - f; wx_[b + 1] = 0.0f - f; wx_[b + 2] = 1.0f - f; wx_[b + 3] = 2.0f - f;
#pragma unroll
            for (int c = 0; c < 4; c++) { wx2[b + c] = wx_[b + c] * wx_[b + c]; v[b + c] = 0.000858519f; }
        } else if (j < 14) {
            const int step = (j - 2) >> 1, b = 4 * ((j - 2) & 1);
#pragma unroll
            for (int c = 0; c < 4; c++) v[b + c] = __builtin_fmaf(v[b + c], wx2[b + c], C[step]);
        } else {
            const int b = 4 * (j - 14);                      // |x| >= 2 can only happen for taps 1 (-1-frac) and 4 (2-frac)
            v[b + 0] = fabsf(wx_[b + 0]) >= 2.0f ? 0.0f : v[b + 0];
            v[b + 3] = fabsf(wx_[b + 3]) >= 2.0f ? 0.0f : v[b + 3];
        }
    };
    // one slice of the division + store conversion of pixel kk (div3_core and store_u, 29 instructions in 7 slices)
    auto d_slice = [&](int j, int kk) {
        const float den = num[kk][3];
        if (j == 0) {
            all_ok = all_ok && (den > 0.5f && den < 2.0f);
            r_ = __builtin_amdgcn_rcpf(den);
            const float e = __builtin_fmaf(-den, r_, 1.0f);
            r_ = __builtin_fmaf(e, r_, r_);
        } else if (j == 2 || j == 4 || j == 6) {
            const int c = (j - 2) >> 1;
            const float n = num[kk][c];
            float a = n * r_, tt = __builtin_fmaf(-den, a, n);
            a = __builtin_fmaf(tt, r_, a); tt = __builtin_fmaf(-den, a, n);
            q_[c] = __builtin_fmaf(tt, r_, a);
        } else if (j == 8 || j == 10 || j == 12) {
            const int c = (j - 8) >> 1;
            o[kk][c] = store_u(q_[c], maxv);
        }
    };
    // the tap read that is AHEAD slices in front of tap j of pixel kk (it may belong to pixel kk + 1)
    auto issue = [&](int j, int kk) {
        const int jj = j + AHEAD, kt = kk + (jj >> 4), tj = jj & 15;
        if (kt < NPX) vals[jj % NV] = t[kt][(tj >> 2) * RS + (tj & 3)];
    };
    // ---- prologue: the first reads of pixel 0 in flight under its weight chains ----
#pragma unroll
    for (int j = 0; j < AHEAD; j++) vals[j % NV] = t[0][(j >> 2) * RS + (j & 3)];
#pragma unroll
    for (int j = 0; j < 16; j++) w_slice(j, 0);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int k = 0; k <= NPX; k++) {
        if (k < NPX) { num[k][0] = 0.0f; num[k][1] = 0.0f; num[k][2] = 0.0f; num[k][3] = 0.0f; }
#pragma unroll
        for (int j = 0; j < 16; j++) {
            if (k < NPX) {
                issue(j, k);
                const f4 val = vals[j % NV];
                const float w2d = wgt[k & 1][j & 3] * wgt[k & 1][4 + (j >> 2)];
                num[k][0] = __builtin_fmaf(w2d, val.x, num[k][0]);
                num[k][1] = __builtin_fmaf(w2d, val.y, num[k][1]);
                num[k][2] = __builtin_fmaf(w2d, val.z, num[k][2]);
                // den + w2d through the tile's trailing 1.0 (w2d * 1.0 is exact: the same sum): with val.w unused the register allocator
                // recycles that quarter of an in-flight ds_read_b128's destination, and every such reuse drains the LDS queue
                num[k][3] = __builtin_fmaf(w2d, val.w, num[k][3]);
                if (k + 1 < NPX) w_slice(j, k + 1);
            }
            if (k >= 1) d_slice(j, k - 1);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

// RN(1 / den) for den in [0.5, 2): v_rcp, one Newton step, one residual correction.  Equal to the IEEE quotient 1.0f / den for EVERY
// float of the range (exhaustive: tools/check_rcp.hip, profiles/r05_check_rcp.txt; the compiler's own expansion makes a second
// correction and scales / fixes up for the operands outside it).  The separable sampler's den = (sum wx)(sum wy) lies in [0.9995, 1.039]
// for every fraction in [0, 1] (the same tool walks all 2^30 + 1 of them), so no range test guards the call.
__device__ __forceinline__ float rcp_rn(float den) {
    float r = __builtin_amdgcn_rcpf(den);
    const float e = __builtin_fmaf(-den, r, 1.0f);
    r = __builtin_fmaf(e, r, r);
    const float tt = __builtin_fmaf(-den, r, 1.0f);
    return __builtin_fmaf(tt, r, r);
}

// VS_WARP_LANCZOS2_SEP of one output pixel, straight-line (the what-if builds and VS_WARP_FAST_PIPE=0): num = {numB, numG, numR, den}
__device__ __forceinline__ void sep_pixel(const lds_f4 t, const f2 fr, float num[4]) {
    float wx[4], wy[4];
    lanczos_weights4_fma(fr.x, wx);
    lanczos_weights4_fma(fr.y, wy);
    float h[4][3];
#pragma unroll
    for (int ry = 0; ry < 4; ry++) {
#pragma unroll
        for (int rx = 0; rx < 4; rx++) {
            const f4 val = t[ry * WS_RS + rx];
            const float vc[3] = {val.x, val.y, val.z};
#pragma unroll
            for (int c = 0; c < 3; c++) h[ry][c] = rx == 0 ? wx[0] * vc[c] : __builtin_fmaf(wx[rx], vc[c], h[ry][c]);
        }
    }
#pragma unroll
    for (int c = 0; c < 3; c++)
        num[c] = __builtin_fmaf(wy[3], h[3][c], __builtin_fmaf(wy[2], h[2][c], __builtin_fmaf(wy[1], h[1][c], wy[0] * h[0][c])));
    num[3] = lanczos_separable_den(wx, wy);
}

// VS_WARP_LANCZOS2_SEP for the RB = 4 pixels of a lane, software-pipelined like fast_rows_pipelined: stage k interleaves, slice by slice
// (16 slices, one tap each), the taps of pixel k (3 fmas per tap into the row sums h, 3 more after a row's fourth tap into num), the
// weight chains of pixel k + 1, the denominator and its reciprocal of pixel k, and the scaling / store conversion of pixel k - 1.  Per
// pixel: 72 weight-chain operations, 7 for den, 5 for RN(1 / den), 48 + 12 tap fmas, 3 products, 9 store conversions = 156 against the
// contracted form's 179 -- bit-identical to lanczos_separable_combine / the oracle's
// VSO_WARP_LANCZOS2_SEPARABLE (the pipelining only decides WHEN an instruction issues).  num[k] = {numB, numG, numR, den}; all_ok is
// not touched (rcp_rn: den cannot leave its range).
template <int NPX, int AHEAD = VS_WARP_PIPE_AHEAD, int RS = WS_RS>
__device__ __forceinline__ void sep_rows_pipelined(const lds_f4 (&t)[NPX], const f2 (&fr)[NPX], float maxv, float (&num)[NPX][4],
                                                   uint32_t (&o)[NPX][3], bool& all_ok) {
    constexpr int NV = 8;
    static_assert(AHEAD >= 1 && AHEAD < NV && 16 % NV == 0, "ring of eight tap registers");
    const float C[6] = {-0.0158853f, 0.128693f, -0.583468f, 1.52229f, -2.05238f, 0.999861f};
    f4 vals[NV];
    float wgt[2][8];                 // finished weights: wgt[k & 1] belongs to pixel k (x taps 1..4, then y taps 1..4)
    float wx_[8], wx2[8];            // the chains under construction: arguments, their squares (values in wgt[(k + 1) & 1])
    float h[3] = {0.f, 0.f, 0.f};    // the row sums of the current source row
    float rc[2] = {0.f, 0.f};        // RN(1 / den): rc[k & 1] belongs to pixel k
    float sx_ = 0.f, r_ = 0.f, qv[3] = {0.f, 0.f, 0.f};
    auto w_slice = [&](int j, int kk) {
        float (&v)[8] = wgt[kk & 1];
        const float f = (j == 0) ? fr[kk].x : fr[kk].y;
        if (j < 2) {
            const int b = 4 * j;
            wx_[b + 0] = -1.0f - f; wx_[b + 1] = 0.0f - f; wx_[b + 2] = 1.0f - f; wx_[b + 3] = 2.0f - f;
#pragma unroll
            for (int c = 0; c < 4; c++) { wx2[b + c] = wx_[b + c] * wx_[b + c]; v[b + c] = 0.000858519f; }
        } else if (j < 14) {
            const int step = (j - 2) >> 1, b = 4 * ((j - 2) & 1);
#pragma unroll
            for (int c = 0; c < 4; c++) v[b + c] = __builtin_fmaf(v[b + c], wx2[b + c], C[step]);
        } else {
            const int b = 4 * (j - 14);                      // |x| >= 2 can only happen for taps 1 (-1-frac) and 4 (2-frac)
            v[b + 0] = fabsf(wx_[b + 0]) >= 2.0f ? 0.0f : v[b + 0];
            v[b + 3] = fabsf(wx_[b + 3]) >= 2.0f ? 0.0f : v[b + 3];
        }
    };
    // den = (sum wx)(sum wy) of pixel kk and its correctly rounded reciprocal (rcp_rn), 12 instructions in 5 slices
    auto r_slice = [&](int j, int kk) {
        const float (&v)[8] = wgt[kk & 1];
        const float den = num[kk][3];
        if (j == 0) sx_ = (v[0] + v[1]) + (v[2] + v[3]);
        else if (j == 1) num[kk][3] = sx_ * ((v[4] + v[5]) + (v[6] + v[7]));
        else if (j == 3) r_ = __builtin_amdgcn_rcpf(den);
        else if (j == 6) {
            const float e = __builtin_fmaf(-den, r_, 1.0f);
            r_ = __builtin_fmaf(e, r_, r_);
        } else if (j == 9) {
            const float tt = __builtin_fmaf(-den, r_, 1.0f);
            rc[kk & 1] = __builtin_fmaf(tt, r_, r_);
        }
    };
    // scaling + store conversion of pixel kk (3 products, 3 x store_u)
    auto d_slice = [&](int j, int kk) {
        if (j == 2 || j == 4 || j == 6) {
            const int c = (j - 2) >> 1;
            qv[c] = num[kk][c] * rc[kk & 1];
        } else if (j == 8 || j == 10 || j == 12) {
            const int c = (j - 8) >> 1;
            o[kk][c] = store_u(qv[c], maxv);
        }
    };
    auto issue = [&](int j, int kk) {
        const int jj = j + AHEAD, kt = kk + (jj >> 4), tj = jj & 15;
        if (kt < NPX) vals[jj % NV] = t[kt][(tj >> 2) * RS + (tj & 3)];
    };
#pragma unroll
    for (int j = 0; j < AHEAD; j++) vals[j % NV] = t[0][(j >> 2) * RS + (j & 3)];
#pragma unroll
    for (int j = 0; j < 16; j++) w_slice(j, 0);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int k = 0; k <= NPX; k++) {
#pragma unroll
        for (int j = 0; j < 16; j++) {
            if (k < NPX) {
                issue(j, k);
                const f4 val = vals[j % NV];
                const int rx = j & 3, ry = j >> 2;
                const float wxv = wgt[k & 1][rx];
                if (rx == 0) { h[0] = wxv * val.x; h[1] = wxv * val.y; h[2] = wxv * val.z; }
                else { h[0] = __builtin_fmaf(wxv, val.x, h[0]); h[1] = __builtin_fmaf(wxv, val.y, h[1]); h[2] = __builtin_fmaf(wxv, val.z, h[2]); }
                // the tile's trailing 1.0 is not used by this form; the empty statement keeps that quarter of the in-flight ds_read_b128's
                // destination allocated until here (a recycled quarter makes the hardware drain the LDS queue, see fast_rows_pipelined)
                asm volatile("" :: "v"(val.w));
                if (rx == 3) {
                    const float wyv = wgt[k & 1][4 + ry];
                    if (ry == 0) { num[k][0] = wyv * h[0]; num[k][1] = wyv * h[1]; num[k][2] = wyv * h[2]; }
                    else { num[k][0] = __builtin_fmaf(wyv, h[0], num[k][0]); num[k][1] = __builtin_fmaf(wyv, h[1], num[k][1]); num[k][2] = __builtin_fmaf(wyv, h[2], num[k][2]); }
                }
                if (k + 1 < NPX) w_slice(j, k + 1);
                r_slice(j, k);
            }
            if (k >= 1) d_slice(j, k - 1);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

// image_warp's bilinear (generators.cpp:148-163) per channel; t = staged pixel (iy, ix)
__device__ __forceinline__ void sample_bilinear(lds_f4 t, f2 fr, float q[3]) {
    const f4 a0 = t[0], a1 = t[1], b0 = t[WS_RS], b1 = t[WS_RS + 1];
    // lerp(a,b,t) = a*(1-t) + b*t (generators.cpp:161-163) per channel, all scalar: three channels fill one and a half packed
    // instructions, and a packed fp32 instruction costs the issue time of two scalar ones (40 issue slots packed, 29 scalar)
    const float tx = fr.x, ty = fr.y, otx = 1.0f - tx, oty = 1.0f - ty;
    const float a0c[3] = {a0.x, a0.y, a0.z}, a1c[3] = {a1.x, a1.y, a1.z}, b0c[3] = {b0.x, b0.y, b0.z}, b1c[3] = {b1.x, b1.y, b1.z};
#pragma unroll
    for (int c = 0; c < 3; c++) {
        const float top = a0c[c] * otx + a1c[c] * tx;
        const float bot = b0c[c] * otx + b1c[c] * tx;
        q[c] = top * oty + bot * ty;
    }
}

// The same on the byte tile: t = dword of staged pixel (iy, ix) = B | G << 8 | R << 16.  Two ds_read2_b32 fetch the 2 x 2 window; the
// twelve conversions are exact, so every float below is the float-tile path's value: bit-identical.
__device__ __forceinline__ void sample_bilinear_u8(const __attribute__((address_space(3))) uint32_t* t, f2 fr, float q[3]) {
    const uint32_t a0 = t[0], a1 = t[1], b0 = t[WS_RS8], b1 = t[WS_RS8 + 1];
    const float tx = fr.x, ty = fr.y, otx = 1.0f - tx, oty = 1.0f - ty;
#pragma unroll
    for (int c = 0; c < 3; c++) {
        const float top = ub(a0, c) * otx + ub(a1, c) * tx;
        const float bot = ub(b0, c) * otx + ub(b1, c) * tx;
        q[c] = top * oty + bot * ty;
    }
}

// ... and on the word tile: t = the 8 bytes {B | G << 16, R | x << 16} of staged pixel (iy, ix)
typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float uw(uint32_t q, int k) { return (float)((q >> (16 * k)) & 0xffffu); }
__device__ __forceinline__ void sample_bilinear_u16(const __attribute__((address_space(3))) u32x2_t* t, f2 fr, float q[3]) {
    const u32x2_t a0 = t[0], a1 = t[1], b0 = t[WS_RS8], b1 = t[WS_RS8 + 1];
    const float tx = fr.x, ty = fr.y, otx = 1.0f - tx, oty = 1.0f - ty;
    const float a0c[3] = {uw(a0.x, 0), uw(a0.x, 1), uw(a0.y, 0)}, a1c[3] = {uw(a1.x, 0), uw(a1.x, 1), uw(a1.y, 0)};
    const float b0c[3] = {uw(b0.x, 0), uw(b0.x, 1), uw(b0.y, 0)}, b1c[3] = {uw(b1.x, 0), uw(b1.x, 1), uw(b1.y, 0)};
#pragma unroll
    for (int c = 0; c < 3; c++) {
        const float top = a0c[c] * otx + a1c[c] * tx;
        const float bot = b0c[c] * otx + b1c[c] * tx;
        q[c] = top * oty + bot * ty;
    }
}

template <typename T, int MODE, int BORDER>
__device__ __forceinline__ void warp_pixel_global(const T* __restrict__ src, int w, int h, int stride, float Wx,
                                                  float Wy, float maxv, uint32_t out[3]) {
    float flx = floorf(Wx), fly = floorf(Wy);
    int ix = sample_index(flx, w), iy = sample_index(fly, h);
    float frx = Wx - flx, fry = Wy - fly;
    auto fetch = [&](int sx, int sy, int c) -> float {
        if (BORDER == 1) return (sx < 0 || sy < 0 || sx >= w || sy >= h) ? 0.0f : (float)src[(size_t)sy * stride + (size_t)sx * 3 + c];
        return (float)src[(size_t)clampi(sy, 0, h - 1) * stride + (size_t)clampi(sx, 0, w - 1) * 3 + c];
    };
    if (MODE == 0) {
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
    } else if (MODE == 2 || MODE == 3) {
        float wx[4], wy[4];
        lanczos_weights4_fma(frx, wx);
        lanczos_weights4_fma(fry, wy);
        const float rden = MODE == 3 ? 1.0f / lanczos_separable_den(wx, wy) : 0.0f;
        for (int c = 0; c < 3; c++) {
            float v[4][4];
#pragma unroll
            for (int ry = 0; ry < 4; ry++)
#pragma unroll
                for (int rx = 0; rx < 4; rx++) v[ry][rx] = fetch(ix + rx - 1, iy + ry - 1, c);
            out[c] = store_u(MODE == 3 ? lanczos_separable_combine(v, wx, wy, rden) : lanczos_contracted_combine(v, wx, wy), maxv);
        }
    } else {
#pragma unroll
        for (int c = 0; c < 3; c++) {
            float top = lerpf(fetch(ix, iy, c), fetch(ix + 1, iy, c), frx);
            float bottom = lerpf(fetch(ix, iy + 1, c), fetch(ix + 1, iy + 1, c), frx);
            out[c] = store_u(lerpf(top, bottom, fry), maxv);
        }
    }
}

// 4 adjacent lanes hold one BGR pixel each (p = B | G<<8 | R<<16); lanes 0..2 of the quad assemble the three
// dwords of the 12-byte group from their own pixel and their right neighbour's: bytes m..m+3 of {own 3 bytes, next 3 bytes}
// = one v_perm_b32 with a per-lane selector (v_perm byte indices: 0-3 = second operand, 4-7 = first operand).
__device__ __forceinline__ uint32_t quad_sel(int m) {
    return m == 0 ? 0x04020100u : (m == 1 ? 0x05040201u : 0x06050402u);
}
__device__ __forceinline__ uint32_t quad_pack_bgr(uint32_t p, uint32_t sel) {
    const uint32_t q = (uint32_t)dpp_mov0<0x101>((int)p);   // row_shl:1 = the right neighbour's pixel (lane 3 of a quad ignores it)
    return __builtin_amdgcn_perm(q, p, sel);
}

// u16 pixels: a pair of lanes owns 2 pixels = 12 bytes = 3 dwords {B0|G0, R0|B1, G1|R1}
__device__ __forceinline__ void pair_pack_bgr16(const uint32_t o[3], int odd, uint32_t& d0, uint32_t& d1) {
    const uint32_t bg = o[0] | (o[1] << 16);
    const uint32_t nbg = (uint32_t)dpp_mov0<0x101>((int)bg);   // row_shl:1 (only even lanes use it)
    d0 = odd ? (o[1] | (o[2] << 16)) : bg;            // odd lane: G1|R1 ; even lane: B0|G0
    d1 = o[2] | (nbg << 16);                           // even lane only: R0|B1
}

// fill item of lane (rr = lane & 3, pair index p): rows 4 * (p / 20) + rr, column group p % 20 (4 source pixels)
struct FillItem { int row, g; };
__device__ __forceinline__ FillItem fill_item(int lane, int slot) {
    const int p = (lane >> 2) + 16 * slot;                   // < 256
    const int rq = (p * 205) >> 12;                          // p / 20
    return FillItem{4 * rq + (lane & 3), p - 20 * rq};
}
template <int G>
__device__ __forceinline__ FillItem fill_item_g(int lane, int slot) {      // the same map over G column groups per row (G = 18: p / 18 = (p * 3641) >> 16, exact below 2000)
    if (G == 20) return fill_item(lane, slot);
    const int p = (lane >> 2) + 16 * slot;
    const int rq = (p * 3641) >> 16;
    return FillItem{4 * rq + (lane & 3), p - G * rq};
}

// SHAPE 1 / 2 = the COMPACT windows (round 6, the contracted and separable Lanczos2 forms): 20 staged rows instead of 24 (25.9 KB of LDS instead of 31.1), at most 85 VGPRs and tap
// reads two ahead instead of six -- SIX waves per SIMD instead of five on this issue-bound kernel -- and, separable form only, 18 column groups at a 73-slot pitch (23.4 KB, <= 73
// VGPRs): SEVEN.  34.9 -> 34.2 -> 33.9 us per 4K frame isolated, `value` +3.3 % (profiles/r06_warp_sep_occupancy.txt).  The launcher
// takes them when every frame's rows (and, for shape 2, columns) fit (the host-side extents say the tile's footprint spans under 16 source rows: rotations up to ~0.9 degrees at unit scale); a tile that
// does not fit its window takes the per-pixel path in either instantiation, so the choice is about speed only -- same arithmetic, same bits.
template <typename T, int MODE, int BORDER, int SHAPE = 0>
__global__ __launch_bounds__(256, (MODE == 2 || MODE == 3) ? (SHAPE == 2 ? 7 : SHAPE == 1 ? 6 : VS_WARP_FAST_MINWAVES) : (raw_tile_of((int)sizeof(T) * 8, MODE) ? 8 : VS_WARP_EXACT_MINWAVES)) void vs_k_bgr_warp_c3(
    const T* __restrict__ src, int w, int h, int src_stride, const float4* __restrict__ params, T* __restrict__ dst,
    int dst_stride, size_t src_fs, size_t dst_fs, int tiles_x, uint32_t tiles_x_magic, int tiles_per_frame, int chunk,
    float maxv, vsk::Roi roi, const float4* __restrict__ extents) {
    constexpr bool RAWTILE = raw_tile_of((int)sizeof(T) * 8, MODE);       // the tile holds source bytes / words, not floats (both depths)
    constexpr int PXD = sizeof(T) == 1 ? 1 : 2;                          // ... dwords per staged pixel
    // this kernel's tile height and what follows from it (the namespace-scope values are those of the 16-row kernels)
    constexpr int WT_H = tile_h_of((int)sizeof(T) * 8, MODE), RPW = WT_H / 4, RB = VS_WARP_ROW_BLOCK < RPW ? VS_WARP_ROW_BLOCK : RPW, WS_H = WT_H + (RAWTILE ? 8 : (SHAPE ? 4 : VS_WARP_WS_EXTRA));
    constexpr bool COMPACT = SHAPE != 0;
    constexpr int KG = SHAPE == 2 ? 18 : WS_W / 4, KRS = SHAPE == 2 ? 4 * KG + 1 : WS_RS;      // column groups per staged row, float-tile row pitch (73 slots = 1168 bytes = 16 (mod 128), like 81)
    static_assert(!COMPACT || (!RAWTILE && (MODE == 2 || MODE == 3)), "the compact windows belong to the float-tile Lanczos2 forms");
    static_assert(SHAPE != 2 || MODE == 3, "seven waves per SIMD: the separable form only (the contracted one needs 77 registers)");
    static_assert(SHAPE != 2 || (VS_WARP_FAST_PIPE && RB == 4 && !VS_WARP_WHATIF), "the 73-slot pitch is known to the pipelined sampler only");
    constexpr int FILL_SLOTS = (WS_H / 4 * KG + 63) / 64;
    static_assert(RPW % RB == 0 && RB % 2 == 0, "rows per wave: a whole number of row blocks, rows in pairs");
    __shared__ f4 tile[RAWTILE ? 1 : WS_H * KRS];         // {B,G,R,1} per staged source pixel
    __shared__ __attribute__((aligned(16))) uint32_t tile_raw[RAWTILE ? WS_H * WS_RS8 * PXD : 4];    // bilinear: B | G << 8 | R << 16 (8-bit frames), {B | G << 16, R} (16-bit containers)
#ifdef VS_WARP_LDS_PAD
    __shared__ uint32_t lds_pad[VS_WARP_LDS_PAD / 4];       // occupancy experiments only: fewer workgroups per CU
    if (w < 0) lds_pad[threadIdx.x] = 0;
#endif
    // XCD-aware tile order: workgroups are dealt round-robin over the 8 XCDs (each with its own L2) in linear id order and
    // gridDim.x is a multiple of 8, so workgroup b of a frame works on its tile (b % 8) * chunk + b / 8: every XCD walks
    // one contiguous run of tiles in raster order and the halo rows / columns shared by neighbouring tiles hit in its L2.
    constexpr int NT = MODE == 1 ? VS_WARP_TILES_PER_WG_BILINEAR : VS_WARP_TILES_PER_WG;
#if VS_WARP_STAMPS
    unsigned long long stamp[STAMP_N] = {};
    VS_STAMP(0);
#endif
    const int tl0 = (int)(blockIdx.x & 7) * chunk + (int)(blockIdx.x >> 3) * NT;         // this workgroup's first tile
    const int tl_end = min(tiles_per_frame, (int)((blockIdx.x & 7) + 1) * chunk);        // end of its XCD's run
    if (tl0 >= tl_end) return;
    const int frame = blockIdx.y;
    const float4 P = params[frame];
    src += (size_t)frame * src_fs;
    dst += (size_t)frame * dst_fs;
    const float A1 = 1.0f + P.x, B = P.y, TX = P.z, TY = P.w;
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const bool src_aligned = ((((uintptr_t)src) | (uintptr_t)((size_t)src_stride * sizeof(T))) & 3) == 0;   // uniform
    float4 E = make_float4(0.f, 0.f, 0.f, 0.f);
    if (extents != nullptr) E = extents[frame];

    // everything about a tile that the fill and the sampler blocks need; all of it wave-uniform
    struct Geom { int x0, y0, sx_lo, sy_lo, rows, groups; bool fits, interior; };
    auto geom = [&](int tl) -> Geom {
    const int tyi = tiles_x == 1 ? tl : (int)__umulhi((uint32_t)tl, tiles_x_magic);         // tl / tiles_x (scalar unit)
    const int txi = tl - tyi * tiles_x;
    // output pixel (x, y) of the window is pixel (x + roi.x, y + roi.y) of the full frame: the sampling position is
    // computed from the full-frame coordinate, so a window equals the same rows / columns cut out of the whole warp
    const int x0 = txi * WT_W, y0 = tyi * WT_H;
    const int x1 = min(x0 + WT_W, roi.w) - 1, y1 = min(y0 + WT_H, roi.h) - 1;

    // source footprint of the tile.  Wx = fl(fl(A1*x) - fl(B*y)) + TX is monotone in x and in y (rounding is monotone), so its
    // extremes over the tile sit at corners chosen by the signs of A1 and B: 4 evaluations -- or, when the host has sent the
    // frame's extents (vsk::bgr_warp_c3: the range of A1*i - B*j and B*i + A1*j over a full tile (64 x WT_H), widened by a bound on
    // the fp32 rounding of the position arithmetic), ONE evaluation at the tile origin plus four adds: every position a pixel
    // of the tile computes lies inside [W00 + lo, W00 + hi], which is all the footprint has to guarantee (it may come out a
    // pixel larger than the exact one; the window has 8 columns and 3 rows to spare for near-identity transforms).
    const float fx0 = (float)(x0 + roi.x), fx1 = (float)(x1 + roi.x), fy0 = (float)(y0 + roi.y), fy1 = (float)(y1 + roi.y);
    float mnx, mxx, mny, mxy;
    if (extents != nullptr) {                               // (uniform)
        const float Wx00 = A1 * fx0 - B * fy0 + TX, Wy00 = B * fx0 + A1 * fy0 + TY;
        mnx = Wx00 + E.x; mxx = Wx00 + E.y; mny = Wy00 + E.z; mxy = Wy00 + E.w;
    } else {
    const float xa = A1 >= 0.f ? fx0 : fx1, xb = A1 >= 0.f ? fx1 : fx0;     // x minimising / maximising A1*x
    const float ya = B >= 0.f ? fy0 : fy1, yb = B >= 0.f ? fy1 : fy0;       // y minimising / maximising B*y
    mnx = A1 * xa - B * yb + TX; mxx = A1 * xb - B * ya + TX;
    const float xc = B >= 0.f ? fx0 : fx1, xd = B >= 0.f ? fx1 : fx0;       // x minimising / maximising B*x
    const float yc = A1 >= 0.f ? fy0 : fy1, yd = A1 >= 0.f ? fy1 : fy0;
    mny = B * xc + A1 * yc + TY; mxy = B * xd + A1 * yd + TY;
    }
    bool fits = fmaxf(fmaxf(fabsf(mnx), fabsf(mxx)), fmaxf(fabsf(mny), fabsf(mxy))) < 1.0e6f;
    int sx_lo = 0, sy_lo = 0, rows = 0, groups = 0;
    if (fits) {
        sx_lo = ((int)floorf(mnx) - 1) & ~3;               // first staged column: a multiple of 4 pixels (12 bytes)
        const int sx_hi = (int)floorf(mxx) + 2;
        sy_lo = (int)floorf(mny) - 1;
        const int sy_hi = (int)floorf(mxy) + 2;
        rows = sy_hi - sy_lo + 1;
        groups = (sx_hi - sx_lo + 4) >> 2;                 // column groups of 4 pixels
        fits = groups <= KG && rows <= WS_H;
    }
    // interior tiles (the whole staged window lies inside an aligned frame: all but the frame's rim): no clamps and no border
    // tests per item, one offset from a uniform base
    const bool interior = !(VS_WARP_WHATIF & 256) && fits && src_aligned && sx_lo >= 0 && sx_lo + 4 * groups <= w && sy_lo >= 0 && sy_lo + rows <= h;   // uniform (analysis bit 256: every tile fills like a rim tile)
    return Geom{x0, y0, sx_lo, sy_lo, rows, groups, fits, interior};
    };

    // 4 source pixels (12 bytes, one aligned load) per work item, converted once, written as 4 float4.
    // (16 + 8) rows x 20 groups = 480 items = 2 per thread; all loads are issued before the first conversion.
    u32x3 q0[FILL_SLOTS], q1[FILL_SLOTS];
    // Interior tiles of the raw-tile (bilinear) kernels take the fixed-point bilinear kernels' row-triplet item map (VS_WARP_CV_ROW_FILL): lane ->
    // (row lane / 20 of a row triplet, column group lane % 20), once per workgroup; slot s of wave wv stages rows 3 (wv + 4 s) + r3, twelve rows
    // further per slot, so the source offset advances by a uniform and the tile address by a constant.  (Not the float tiles: 64 bytes per
    // item there, and consecutive lanes 64 bytes apart put four lanes of every 16 on the same banks.)
    constexpr bool ROWFILL = RAWTILE && VS_WARP_CV_ROW_FILL != 0;
    static_assert(!ROWFILL || (WS_W / 4 == 20 && FILL_SLOTS == (WS_H + 11) / 12), "row-triplet item map");
    const int rf_r3 = (int)(((uint32_t)lane * 13u) >> 8), rf_g = lane - 20 * rf_r3, rf_row0 = 3 * wv + rf_r3;      // lane / 20, lane % 20
    auto item_of = [&](int s, bool interior) -> FillItem {
        if (ROWFILL && interior) return FillItem{rf_r3 < 3 ? rf_row0 + 12 * s : WS_H, rf_g};                // (lanes 60..63 carry no item: a row beyond every tile)
        return fill_item_g<KG>(lane, wv + 4 * s);
    };
    // the loads of an interior tile: into q0 / q1, which the fill converts -- right away, or (VS_WARP_TILES_PER_WG > 1) after the
    // previous tile's sampler blocks, so that a tile's memory latency lies under the tile before it
    auto issue = [&](const Geom& g) {
        const T* base = (VS_WARP_WHATIF & 32) ? src : src + ((size_t)g.sy_lo * src_stride + (size_t)g.sx_lo * 3);
        const uint32_t rf_off0 = (uint32_t)rf_row0 * (uint32_t)src_stride + 12u * (uint32_t)rf_g;
#pragma unroll
        for (int s = 0; s < FILL_SLOTS; s++) {
            const FillItem it = item_of(s, true);
            uint32_t off = ROWFILL ? rf_off0 + (uint32_t)(12 * s) * (uint32_t)src_stride
                                   : (uint32_t)it.row * (uint32_t)src_stride + 12u * (uint32_t)it.g;      // elements
            if (VS_WARP_WHATIF & 32) off &= 0xffcu;                  // (analysis: every load hits the same few cache lines)
            if (it.row < g.rows && it.g < g.groups) {
                q0[s] = *(const u32x3*)(base + off);
                if (sizeof(T) == 2) q1[s] = *(const u32x3*)(base + off + 6);
            }
        }
    };
    auto fill = [&](const Geom& g, bool loaded) {
        const int sx_lo = g.sx_lo, sy_lo = g.sy_lo, rows = g.rows, groups = g.groups;
        FillItem it[FILL_SLOTS];
        bool live[FILL_SLOTS], direct[FILL_SLOTS];
        const T* rowp[FILL_SLOTS];
        if (g.interior) {
            if (!loaded) issue(g);
            VS_STAMP(2);
            VS_STAMP_DRAIN();
            VS_STAMP(3);
#pragma unroll
            for (int s = 0; s < FILL_SLOTS; s++) {
                it[s] = item_of(s, true);
                live[s] = it[s].row < rows && it[s].g < groups;
                direct[s] = live[s];
                rowp[s] = src;
            }
        } else {
#pragma unroll
        for (int s = 0; s < FILL_SLOTS; s++) {
            it[s] = fill_item_g<KG>(lane, wv + 4 * s);
            live[s] = it[s].row < rows && it[s].g < groups;
            const int sy = sy_lo + it[s].row, sx = sx_lo + 4 * it[s].g;
            rowp[s] = src + (size_t)clampi(sy, 0, h - 1) * src_stride;
            const bool row_in = sy >= 0 && sy < h;
            direct[s] = live[s] && src_aligned && sx >= 0 && sx + 3 < w && (BORDER == 0 || row_in);
            if (direct[s]) {
                q0[s] = *(const u32x3*)(rowp[s] + sx * 3);
                if (sizeof(T) == 2) q1[s] = *(const u32x3*)(rowp[s] + sx * 3 + 6);
            }
        }
        }
        uint32_t one = 1u;
        asm volatile("" : "+v"(one));                        // opaque: keeps the conversion below from folding to a constant
        if (RAWTILE) {
#pragma unroll
            for (int s = 0; s < FILL_SLOTS; s++) {
                if (!live[s]) continue;
                VS_BOUNDS_CHECK(it[s].row * WS_RS8 + 4 * it[s].g + 3, WS_H * WS_RS8, 204);
                typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
                if (sizeof(T) == 2) {
                    u32x4 lo, hi;                            // pixels 0, 1 and 2, 3 of the group
                    if (direct[s]) {
                        const u32x3 a = q0[s], b = q1[s];    // a = B0G0 R0B1 G1R1 ; b = B2G2 R2B3 G3R3 (the sampler reads the low word of .y only)
                        lo = u32x4{a.x, a.y, __builtin_amdgcn_alignbyte(a.z, a.y, 2), a.z >> 16};
                        hi = u32x4{b.x, b.y, __builtin_amdgcn_alignbyte(b.z, b.y, 2), b.z >> 16};
                    } else {
                        const int sy = sy_lo + it[s].row, sx = sx_lo + 4 * it[s].g;
                        const bool row_in = sy >= 0 && sy < h;
                        uint32_t d[4][2];
#pragma unroll
                        for (int k = 0; k < 4; k++) {
                            const int pxi = sx + k;
                            if (BORDER == 1 && (!row_in || pxi < 0 || pxi >= w)) d[k][0] = d[k][1] = 0u;
                            else {
                                const T* qq = rowp[s] + clampi(pxi, 0, w - 1) * 3;
                                d[k][0] = (uint32_t)qq[0] | ((uint32_t)qq[1] << 16);
                                d[k][1] = (uint32_t)qq[2];
                            }
                        }
                        lo = u32x4{d[0][0], d[0][1], d[1][0], d[1][1]};
                        hi = u32x4{d[2][0], d[2][1], d[3][0], d[3][1]};
                    }
                    VS_BOUNDS_CHECK((it[s].row * WS_RS8 + 4 * it[s].g + 3) * 2 + 1, WS_H * WS_RS8 * 2, 209);
                    u32x4* dstp = (u32x4*)(tile_raw + 2 * VS_DEBUG_CLAMP(it[s].row * WS_RS8 + 4 * it[s].g, WS_H * WS_RS8 - 3));
                    dstp[0] = lo;
                    dstp[1] = hi;
                    continue;
                }
                u32x4 px;
                if (direct[s]) {
                    const u32x3 q = q0[s];                   // B0 G0 R0 B1 | G1 R1 B2 G2 | R2 B3 G3 R3  ->  four dwords B G R 0
                    px.x = q.x & 0x00ffffffu;
                    px.y = __builtin_amdgcn_perm(q.y, q.x, 0x0c050403u);       // {x3, y0, y1, 0}
                    px.z = __builtin_amdgcn_perm(q.z, q.y, 0x0c040302u);       // {y2, y3, z0, 0}
                    px.w = q.z >> 8;
                } else {
                    // frame border, an unaligned frame, or (constant border) a row outside the frame: pixel by pixel
                    const int sy = sy_lo + it[s].row, sx = sx_lo + 4 * it[s].g;
                    const bool row_in = sy >= 0 && sy < h;
                    uint32_t d[4];
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        const int pxi = sx + k;
                        if (BORDER == 1 && (!row_in || pxi < 0 || pxi >= w)) d[k] = 0u;
                        else {
                            const T* qq = rowp[s] + clampi(pxi, 0, w - 1) * 3;
                            d[k] = (uint32_t)qq[0] | ((uint32_t)qq[1] << 8) | ((uint32_t)qq[2] << 16);
                        }
                    }
                    px = u32x4{d[0], d[1], d[2], d[3]};
                }
                *(u32x4*)(tile_raw + VS_DEBUG_CLAMP(it[s].row * WS_RS8 + 4 * it[s].g, WS_H * WS_RS8 - 3)) = px;
            }
            return;
        }
#pragma unroll
        for (int s = 0; s < FILL_SLOTS; s++) {
            if (!live[s]) continue;
            VS_BOUNDS_CHECK(it[s].row * KRS + 4 * it[s].g + 3, WS_H * KRS, 201);     // four staged pixels of one fill item
            f4* t = tile + VS_DEBUG_CLAMP(it[s].row * KRS + 4 * it[s].g, WS_H * KRS - 3);
            if (direct[s]) {
                if (sizeof(T) == 1) {
                    const u32x3 q = q0[s];                   // B0 G0 R0 B1 | G1 R1 B2 G2 | R2 B3 G3 R3
                    // the trailing 1.0 is converted from a byte like its three neighbours: four v_cvt_f32_ubyte into four
                    // consecutive registers per pixel (with a literal 1.f the compiler builds the vector through moves)
                    t[0] = f4{ub(q.x, 0), ub(q.x, 1), ub(q.x, 2), ub(one, 0)};
                    t[1] = f4{ub(q.x, 3), ub(q.y, 0), ub(q.y, 1), ub(one, 0)};
                    t[2] = f4{ub(q.y, 2), ub(q.y, 3), ub(q.z, 0), ub(one, 0)};
                    t[3] = f4{ub(q.z, 1), ub(q.z, 2), ub(q.z, 3), ub(one, 0)};
                } else {
                    const u32x3 a = q0[s], b = q1[s];        // a = B0G0 R0B1 G1R1 ; b = B2G2 R2B3 G3R3 (16 bits each)
                    // (the 1.0 converted from a register like its neighbours, as above: no vector-building moves)
                    t[0] = f4{(float)(a.x & 0xffffu), (float)(a.x >> 16), (float)(a.y & 0xffffu), ub(one, 0)};
                    t[1] = f4{(float)(a.y >> 16), (float)(a.z & 0xffffu), (float)(a.z >> 16), ub(one, 0)};
                    t[2] = f4{(float)(b.x & 0xffffu), (float)(b.x >> 16), (float)(b.y & 0xffffu), ub(one, 0)};
                    t[3] = f4{(float)(b.y >> 16), (float)(b.z & 0xffffu), (float)(b.z >> 16), ub(one, 0)};
                }
            } else {
                // frame border, an unaligned frame, or (constant border) a row outside the frame: pixel by pixel
                const int sy = sy_lo + it[s].row, sx = sx_lo + 4 * it[s].g;
                const bool row_in = sy >= 0 && sy < h;
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const int px = sx + k;
                    if (BORDER == 1 && (!row_in || px < 0 || px >= w)) {
                        t[k] = f4{0.f, 0.f, 0.f, 1.f};
                    } else {
                        const T* q = rowp[s] + clampi(px, 0, w - 1) * 3;
                        t[k] = f4{(float)q[0], (float)q[1], (float)q[2], 1.f};
                    }
                }
            }
        }
    };

    // the sampler blocks and stores of one tile (LDS filled, barrier passed)
    auto sample_tile = [&](const Geom& g) {
    const int x0 = g.x0, y0 = g.y0, sx_lo = g.sx_lo, sy_lo = g.sy_lo;
    const bool fits = g.fits;
    const int x = x0 + lane;
    const int yw = y0 + wv * RPW;                            // first row of this wave
    if (yw >= roi.h) return;                                 // wave-uniform
    // a quad of lanes owns 4 pixels = 12 output bytes; it stores them as 3 aligned dwords when the whole
    // quad is inside the row and every row is dword aligned, else byte by byte
    const int m = lane & 3;
    const bool rows_aligned = ((((uintptr_t)dst) | (uintptr_t)((size_t)dst_stride * sizeof(T))) & 3) == 0;   // uniform

    if (!fits) {
        // footprint too large for the LDS window: per-pixel taps through L1 / L2
        for (int k = 0; k < RPW; k++) {
            const int y = yw + k;
            if (y >= roi.h || x >= roi.w) continue;
            const float fx = (float)(x + roi.x), fy = (float)(y + roi.y);
            const float Wx = A1 * fx - B * fy + TX, Wy = B * fx + A1 * fy + TY;
            uint32_t o[3];
            warp_pixel_global<T, MODE, BORDER>(src, w, h, src_stride, Wx, Wy, maxv, o);
            T* q = dst + (size_t)y * dst_stride + (size_t)x * 3;
            q[0] = (T)o[0]; q[1] = (T)o[1]; q[2] = (T)o[2];
        }
        return;
    }

    // ---- the rows of this wave, RB at a time, each group one straight-line block ----
    const int xq = min(x, roi.w - 1);                        // lanes right of the window sample its last column (masked below)
    const float fx = (float)(xq + roi.x);
    const float A1x = A1 * fx, Bx = B * fx;
    // LDS byte offset of the window origin = 16 * ((fly - oy) * WS_RS + (flx - ox)); all terms are small integers, exact in fp32
    constexpr int org = (MODE == 1) ? 0 : 1;                 // Lanczos windows start one pixel up / left of floor()
    const float c0 = RAWTILE ? -4.0f * PXD * (float)(sy_lo * WS_RS8 + sx_lo) : -16.0f * (float)((sy_lo + org) * KRS + (sx_lo + org));
    const bool lane_in = x < roi.w;
    const int yw_first = yw;
#pragma unroll 1
    for (int kb = 0; kb < RPW; kb += RB) {
    const int yw = yw_first + kb;                            // first row of this block
    if (yw >= roi.h) break;                                  // wave-uniform
    uint32_t o[RB][3];
    float num[RB][4];                                          // Lanczos modes: {numB, numG, numR, den} kept for the operator/ fallback
    bool all_ok = true;
    // rows are processed two at a time, the two rows' instructions alternating in source order: a packed-fp32 result
    // cannot feed the very next VALU instruction without a wait state on gfx950, and the other row's operation fills it
    // (VS_WARP_COORDS_FIRST: the positions of all RB rows before the first sampler block -- scalar fp32 instructions cost more
    // between packed ones than among themselves, tools/ubench_mix.hip)
    f2 fr_all[RB];
    lds_f4 t_all[RB];
#pragma unroll
    for (int k = 0; k < RB; k++) {
        const int yq = min(yw + k, roi.h - 1);               // rows below the window repeat its last row (masked below)
        const float fy = (float)(yq + roi.y);
        const float Wx = A1x - B * fy + TX;                  // generators.cpp:141
        const float Wy = Bx + A1 * fy + TY;                  // generators.cpp:142
        const float flx = floorf(Wx), fly = floorf(Wy);
        fr_all[k] = f2{Wx - flx, Wy - fly};
        // (the whole tap window of the pixel -- 4 x 4 staged pixels from boff, 2 x 2 for the bilinear mode -- lies inside the tile)
        const int boff = RAWTILE ? VS_DEBUG_CLAMP_BYTES((int)__builtin_fmaf(fly, 4.0f * PXD * WS_RS8, __builtin_fmaf(flx, 4.0f * PXD, c0)), 4 * PXD * (WS_H * WS_RS8 - (WS_RS8 + 1)), 205)
                                : VS_DEBUG_CLAMP_BYTES((int)__builtin_fmaf(fly, 16.0f * KRS, __builtin_fmaf(flx, 16.0f, c0)),
                                                       16 * (WS_H * KRS - (MODE == 1 ? KRS + 1 : 3 * KRS + 3)), 202);
        t_all[k] = RAWTILE ? (lds_f4)((const __attribute__((address_space(3))) char*)tile_raw + boff)
                          : (lds_f4)((const __attribute__((address_space(3))) char*)tile + boff);
    }
#if VS_WARP_COORDS_FIRST
    __builtin_amdgcn_sched_barrier(0);
#endif
    if (MODE == 2 && VS_WARP_FAST_PIPE && RB == 4 && !VS_WARP_WHATIF) {
        fast_rows_pipelined<RB, COMPACT ? VS_WARP_PIPE_AHEAD_COMPACT : VS_WARP_PIPE_AHEAD, KRS>(t_all, fr_all, maxv, num, o, all_ok);
    } else if (MODE == 3 && VS_WARP_FAST_PIPE && RB == 4 && !VS_WARP_WHATIF) {
        sep_rows_pipelined<RB, COMPACT ? VS_WARP_PIPE_AHEAD_COMPACT : VS_WARP_PIPE_AHEAD, KRS>(t_all, fr_all, maxv, num, o, all_ok);
    } else
#pragma unroll
    for (int kp = 0; kp < RB; kp += 2) {
        const f2 fr[2] = {fr_all[kp], fr_all[kp + 1]};
        const lds_f4 t[2] = {t_all[kp], t_all[kp + 1]};
        float q[2][3];
        if (MODE == 0) {
            exact_pair(t, fr, &num[kp]);
#pragma unroll
            for (int j = 0; j < 2; j++) {
                all_ok = all_ok && (num[kp + j][3] > 0.5f && num[kp + j][3] < 2.0f);
                if (VS_WARP_WHATIF & 4) { q[j][0] = num[kp + j][0]; q[j][1] = num[kp + j][1]; q[j][2] = num[kp + j][2] + num[kp + j][3]; }
                else div3_core(num[kp + j][0], num[kp + j][1], num[kp + j][2], num[kp + j][3], q[j]);
            }
        } else if (MODE == 3) {
#pragma unroll
            for (int j = 0; j < 2; j++) {
                sep_pixel(t[j], fr[j], num[kp + j]);
                const float r = rcp_rn(num[kp + j][3]);
                q[j][0] = num[kp + j][0] * r; q[j][1] = num[kp + j][1] * r; q[j][2] = num[kp + j][2] * r;
            }
        } else if (MODE == 2) {
            fast_pair(t, fr, &num[kp]);
#pragma unroll
            for (int j = 0; j < 2; j++) {
                all_ok = all_ok && (num[kp + j][3] > 0.5f && num[kp + j][3] < 2.0f);
                if (VS_WARP_WHATIF & 4) { q[j][0] = num[kp + j][0]; q[j][1] = num[kp + j][1]; q[j][2] = num[kp + j][2] + num[kp + j][3]; }
                else div3_core(num[kp + j][0], num[kp + j][1], num[kp + j][2], num[kp + j][3], q[j]);
            }
        } else {
            if (RAWTILE && sizeof(T) == 2) {
                sample_bilinear_u16((const __attribute__((address_space(3))) u32x2_t*)t[0], fr[0], q[0]);
                sample_bilinear_u16((const __attribute__((address_space(3))) u32x2_t*)t[1], fr[1], q[1]);
            } else if (RAWTILE) {
                sample_bilinear_u8((const __attribute__((address_space(3))) uint32_t*)t[0], fr[0], q[0]);
                sample_bilinear_u8((const __attribute__((address_space(3))) uint32_t*)t[1], fr[1], q[1]);
            } else {
                sample_bilinear(t[0], fr[0], q[0]);
                sample_bilinear(t[1], fr[1], q[1]);
            }
        }
#pragma unroll
        for (int j = 0; j < 2; j++) {
            o[kp + j][0] = store_u(q[j][0], maxv);
            o[kp + j][1] = store_u(q[j][1], maxv);
            o[kp + j][2] = store_u(q[j][2], maxv);
        }
        if (MODE == 2 && VS_WARP_FAST_SCHED >= 1) __builtin_amdgcn_sched_barrier(0);   // keeps the second pair's 32 LDS reads (128 VGPRs) behind the first pair
    }
    if (MODE != 1 && MODE != 3 && !VS_WARP_WHATIF && __any(!all_ok)) {
        // a weight sum outside (0.5, 2): cannot happen for frac in [0,1]; kept so that the result is operator/ whatever the input
#pragma unroll
        for (int k = 0; k < RB; k++) {
            float den = num[k][3];
            asm volatile("" : "+v"(den));                    // keeps the three divisions inside this branch (no speculation)
#pragma unroll
            for (int c = 0; c < 3; c++) o[k][c] = store_u(num[k][c] / den, maxv);
        }
    }

    // ---- stores ----
    // all rows of the block are final here: without this pin the compiler sinks each row's arithmetic into the conditional store
    // blocks below while its LDS reads stay hoisted at the top, and the tile values spill to scratch in between
#pragma unroll
    for (int k = 0; k < RB; k++) asm volatile("" :: "v"(o[k][0]), "v"(o[k][1]), "v"(o[k][2]));
    if (sizeof(T) == 1) {
        const bool quad_in = (x | 3) < roi.w;
        const uint32_t sel = quad_sel(m);
        const uint32_t loff = (uint32_t)(x & ~3) * 3u + 4u * (uint32_t)m;
#pragma unroll
        for (int k = 0; k < RB; k++) {
            const int y = yw + k;
            const uint32_t p = o[k][0] | (o[k][1] << 8) | (o[k][2] << 16);
            const uint32_t d = quad_pack_bgr(p, sel);        // every lane of the wave takes part in the shuffle
            if (y < roi.h) {                                 // wave-uniform
                uint8_t* orow = (uint8_t*)dst + (size_t)y * dst_stride;
                if (rows_aligned && quad_in) {
                    if (m < 3 && (!(VS_WARP_WHATIF & 8) || d == 0x12345678u)) VS_STORE32((uint32_t*)(orow + loff), d);
                } else if (lane_in) {
                    orow[(size_t)x * 3] = (uint8_t)o[k][0];
                    orow[(size_t)x * 3 + 1] = (uint8_t)o[k][1];
                    orow[(size_t)x * 3 + 2] = (uint8_t)o[k][2];
                }
            }
        }
    } else {
        const bool pair_in = (x | 1) < roi.w;
#pragma unroll
        for (int k = 0; k < RB; k++) {
            const int y = yw + k;
            uint32_t d0, d1;
            pair_pack_bgr16(o[k], x & 1, d0, d1);
            if (y < roi.h) {
                T* orow = dst + (size_t)y * dst_stride;
                if (rows_aligned && pair_in) {
                    uint32_t* q = (uint32_t*)(orow + (size_t)(x & ~1) * 3);   // 12 bytes per pixel pair
                    if (x & 1) VS_STORE32(q + 2, d0);
                    else { VS_STORE32(q, d0); VS_STORE32(q + 1, d1); }
                } else if (lane_in) {
                    orow[(size_t)x * 3] = (T)o[k][0];
                    orow[(size_t)x * 3 + 1] = (T)o[k][1];
                    orow[(size_t)x * 3 + 2] = (T)o[k][2];
                }
            }
        }
    }
    }   // row blocks
    };

    Geom g = geom(tl0);
    VS_STAMP(1);
    bool loaded = false;
    if (NT > 1 && g.interior) { issue(g); loaded = true; }
#pragma unroll 1
    for (int i = 0; i < NT; i++) {
        const bool more = i + 1 < NT && tl0 + i + 1 < tl_end;        // (uniform)
        if (g.fits && !(VS_WARP_WHATIF & 2)) fill(g, loaded);
        VS_STAMP(4);
        __syncthreads();
        VS_STAMP(5);
        Geom gn = g;
        bool loaded_n = false;
        if (more) {
            gn = geom(tl0 + i + 1);
            if (gn.interior) { issue(gn); loaded_n = true; }          // in flight during this tile's sampler blocks
        }
        sample_tile(g);
        VS_STAMP(6);
        VS_STAMP_DRAIN();
        VS_STAMP(7);
#if VS_WARP_STAMPS
        {
            const unsigned wg = blockIdx.y * gridDim.x + blockIdx.x;
            stamp[8] = __builtin_amdgcn_s_getreg(4 | (31 << 11));              // HW_REG_HW_ID
            stamp[9] = __builtin_amdgcn_s_getreg(20 | (31 << 11));             // HW_REG_XCC_ID
            if (lane == 0 && wg < (unsigned)STAMP_WGS && g.interior)
                for (int i = 0; i < STAMP_N; i++) g_warp_stamps[((size_t)wg * 4 + wv) * STAMP_N + i] = stamp[i];
        }
#endif
        if (!more) break;
        __syncthreads();                                               // every wave has read its taps: the tile buffer is free
        g = gn; loaded = loaded_n;
    }
}

// ------------------------------------------------------------------------------------------------------------------------------------
// VS_WARP_BILINEAR_CV on interleaved 8-bit BGR: cv::warpAffine(INTER_LINEAR) as the reference's stabilizer calls it (stabilizer.cpp:97-99
// -> imgproc.cpp:446-484), OpenCV 4.x's classic fixed-point path restated (the CPU restatement's cv_warp_impl carries the derivation): the
// matrix inverted in double precision, source coordinates in 1/32 pixel from integer adds (AB_BITS 10, INTER_BITS 5), four 15-bit integer
// weights 32 a b (a, b in 0..32), result = (sum + 2^14) >> 15.  Integer work end to end:
//   * one workgroup = one 64 x 64 output tile; the source footprint goes into LDS as BYTES, one dword {B,G,R,0} per pixel (the byte tile of
//     the float bilinear kernel: 23 KB at its own 80-dword pitch, 7 workgroups per CU), borders resolved in the copy;
//   * per tile the column deltas adelta / bdelta (two double products, cvRound) and the row origins X0 / Y0 are computed ONCE per workgroup
//     (one wave each) into LDS tables: per pixel the position is two integer adds;
//   * per pixel and channel: two v_perm_b32 put the channel's bytes of a window row side by side as a u16 pair {v(x), v(x + 1)}, two
//     v_dot2_u32_u16 against the row's weight pair accumulate the four taps on top of the rounding constant; ~44 vector instructions per
//     64 pixels in all against the float bilinear's 88.
// 4K: 11.1 us per frame = 0.56 of the HBM peak (float bilinear 15.5 = 0.40).  In-kernel stamps: vector-issue-bound by the sum of ALL its
// instructions -- three-operand integer instructions issue at half the fma's rate -- not by memory latency or occupancy (profiles/r05_warp_cv.md).
// Tiles whose footprint does not fit the window (large rotation / zoom) take a per-pixel global path in the same kernel.
// ------------------------------------------------------------------------------------------------------------------------------------
#ifndef VS_WARP_CV_W16
#define VS_WARP_CV_W16 1                 // sampler: 16-bit weights 64 a b (the top-left one saturated to 65535): the sample lands on a byte boundary
#endif
// (Measured and dropped, round 6 -- commit 2 of the round, profiles/r06_warp_cv.md: interior tiles filled by LDS-DMA, one `global_load_lds_dword` per 64
// staged pixels with each lane reading the four bytes at its pixel's byte-aligned address, which lands the sampler's own tile format with no registers and no
// formatting instructions.  Bit-identical (tools/ubench_glds3.hip: right for every byte offset and pitch), 13.5 us per 4K frame against 10.8: 144 four-byte
// gathers per tile cost the texture path more than 21 twelve-byte loads and their formatting cost the vector unit.)
#ifndef VS_WARP_CV_TILE_H
#define VS_WARP_CV_TILE_H 64             // output rows per workgroup: 32, or a multiple of 64 (the row-origin table is filled 32 rows per wave pass).
                                         // Measured (profiles/r05_warp_cv.md): 32 rows 12.3 us per 4K frame, 64 rows 11.1, 128 rows 12.9 -- the taller tile halves the
                                         // prologue and halo shares but leaves 3 workgroups per CU (48 KB of LDS) and the fill of one is no longer covered by the others' sampling
#endif
constexpr int CV_TH = VS_WARP_CV_TILE_H, CV_RPW = CV_TH / 4, CV_WS_H = CV_TH + 8;
#ifndef VS_WARP_CV_RS
#define VS_WARP_CV_RS WS_W               // row pitch of the 8-bit kernel's byte tile in dwords (>= WS_W, a multiple of 4).  80: 23.0 KB of LDS, SEVEN workgroups per CU (round 6);
                                         // 88 (the pitch of the float-bilinear byte tile, the 8-bit kernel's until round 6): 25.3 KB, six -- 10.7 us per 4K frame against 10.4.
                                         // A compact window (72-dword pitch, 68 rows: 19.6 KB, eight workgroups per CU) for stabilisation-sized transforms was built too:
                                         // equal to seven (profiles/r06_warp_cv.md section 5), not kept.
#endif
constexpr int CV_RS = VS_WARP_CV_RS;
static_assert(CV_RS >= WS_W && CV_RS % 4 == 0, "tile pitch");
static_assert(CV_TH % 64 == 0 || CV_TH == 32, "row-origin table: waves 2 and 3 fill it 32 rows per pass each (X0 in lanes 0..31, Y0 in lanes 32..63)");
constexpr int CV_RBK = CV_RPW < 16 ? CV_RPW : 16;          // rows of a wave sampled in one basic block
static_assert(CV_RPW % CV_RBK == 0, "whole row blocks");
constexpr int CV_FILL_SLOTS_RIM = (CV_WS_H / 4 * (WS_W / 4) + 63) / 64, CV_FILL_SLOTS_ROW = (CV_WS_H + 11) / 12;      // the rim path's item map / the row-triplet map
constexpr int CV_FILL_SLOTS = CV_FILL_SLOTS_RIM > CV_FILL_SLOTS_ROW ? CV_FILL_SLOTS_RIM : CV_FILL_SLOTS_ROW;
static_assert(CV_WS_H / 4 * (WS_W / 4) < 1024, "fill_item's p / 20 is exact below 1024");

typedef unsigned short us2 __attribute__((ext_vector_type(2)));
// packed 16-bit products of both halves of `a` with the LOW half of `b` (the second one saturating at 65535: the VOP3P clamp bit)
__device__ __forceinline__ uint32_t pk_mul_lo_u16_lo(uint32_t a, uint32_t b) {
    uint32_t r;
    asm("v_pk_mul_lo_u16 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ uint32_t pk_mul_lo_u16_lo_sat(uint32_t a, uint32_t b) {
    uint32_t r;
    asm("v_pk_mad_u16 %0, %1, %2, 0 op_sel_hi:[1,0,0] clamp" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ uint32_t udot2(uint32_t a, uint32_t b, uint32_t c) {
    return __builtin_amdgcn_udot2(__builtin_bit_cast(us2, a), __builtin_bit_cast(us2, b), c, false);
}

// The coordinate tables of a launch: per frame adelta[tab_w] | bdelta[tab_w] | X0[tab_h] | Y0[tab_h] (ints), tab_w / tab_h = the output window
// rounded up to whole tiles; entries beyond the window repeat its last column / row.  One thread per entry: each is ONE double-precision
// evaluation exactly as OpenCV's warpAffine makes it once per call (cvRound(m x 1024), cvRound((m y + t) 1024) + round_delta).
// (calls of up to vsk::kCvInlineFrames frames hand their matrices to this kernel BY VALUE, as kernel arguments -- 48 bytes per frame: no upload stands between
// the host's numbers and the tables, and the call is two stream operations (tables, warp) instead of three; larger batches read them from the parameter ring)
struct CvMatrices { double m[6 * vsk::kCvInlineFrames]; };
template <bool INLINE>
__global__ __launch_bounds__(256) void vs_k_cv_tables(const double* __restrict__ minv, CvMatrices inl, int* __restrict__ tab, int tab_w, int tab_h, vsk::Roi roi) {
    const int per = 2 * (tab_w + tab_h);
    const int i = (int)(blockIdx.x * 256u + threadIdx.x);
    if (i >= per) return;
    // (the by-value matrices sit in the kernel-argument segment: indexing them by the frame is a scalar load at a computed offset)
    double Mv[6];
#pragma unroll
    for (int k = 0; k < 6; k++) Mv[k] = INLINE ? inl.m[6 * blockIdx.y + k] : minv[6 * (size_t)blockIdx.y + k];
    const double* M = Mv;
    int v;
    if (i < 2 * tab_w) {
        const int xx = i < tab_w ? i : i - tab_w;
        v = cv_delta(i < tab_w ? M[0] : M[3], min(xx, roi.w - 1) + roi.x);
    } else {
        const int j = i - 2 * tab_w, yy = j < tab_h ? j : j - tab_h;
        const int fy = min(yy, roi.h - 1) + roi.y;
        v = j < tab_h ? cv_row_origin(M[1], M[2], fy) : cv_row_origin(M[4], M[5], fy);
    }
    tab[(size_t)blockIdx.y * (size_t)per + i] = v;
}

template <int BORDER>
__device__ __forceinline__ void cv_pixel_global(const uint8_t* __restrict__ src, int w, int h, int stride, int X, int Y, uint32_t out[3]) {
    const int sx = clampi(X >> 5, -32768, 32767), sy = clampi(Y >> 5, -32768, 32767);      // saturate_cast<short>
    const int a1 = X & 31, b1 = Y & 31, a0 = 32 - a1, b0 = 32 - b1;
    auto tap = [&](int xx, int yy, int c) -> int {
        if (BORDER == 1) { if (xx < 0 || yy < 0 || xx >= w || yy >= h) return 0; }
        else { xx = clampi(xx, 0, w - 1); yy = clampi(yy, 0, h - 1); }
        return (int)src[(size_t)yy * stride + (size_t)xx * 3 + c];
    };
#pragma unroll
    for (int c = 0; c < 3; c++)
        out[c] = (uint32_t)((tap(sx, sy, c) * (a0 * b0) + tap(sx + 1, sy, c) * (a1 * b0) + tap(sx, sy + 1, c) * (a0 * b1) + tap(sx + 1, sy + 1, c) * (a1 * b1) + 512) >> 10);
}

// (Measured and dropped, profiles/r05_warp_cv.md: workgroups that walk several tiles with the next tile's tables computed and its source loads
// in flight -- in registers -- while the current tile is sampled.  Left to the scheduler the loads sink behind the sampling code and the 18
// registers still cost occupancy (13.4 us per 4K frame against 11.5, the same at 1 / 2 / 4 / 8 tiles per workgroup); pinned early by
// scheduling fences the kernel spills 17 registers at the 80 the occupancy allows and takes 17-22 us.  Fill items dealt over the tile's own
// column groups instead of the window's 20, non-temporal stores: no change.)
template <int BORDER>
__global__ __launch_bounds__(256, 7) void vs_k_bgr_warp_cv_c3(const uint8_t* __restrict__ src, int w, int h, int src_stride,
                                                             const int* __restrict__ tab, int tab_w, int tab_h, uint8_t* __restrict__ dst, int dst_stride,
                                                             size_t src_fs, size_t dst_fs, int tiles_x, uint32_t tiles_x_magic, int tiles_per_frame,
                                                             int chunk, vsk::Roi roi) {
    __shared__ __attribute__((aligned(16))) uint32_t tile_raw[CV_WS_H * CV_RS];     // B | G << 8 | R << 16 per staged source pixel
    // The fixed-point coordinate tables -- adelta[tab_w] | bdelta[tab_w] | X0[tab_h] | Y0[tab_h] per frame -- are made ONCE PER FRAME by
    // vs_k_cv_tables in front of this launch, as cv::warpAffine makes them once per call (until round 5 every workgroup rebuilt its 64 + 64 +
    // 2 x 64 entries in fp64 and shared them through LDS behind a barrier: 29 % of a wave's life, profiles/r05_cv_stamps_final.json).  A tile
    // reads its footprint corners and its rows' origins with SCALAR loads (uniform addresses) and its lane's two column deltas with one
    // vector load each: no fp64, no table barrier, no LDS reads of the tables in the sampler.
#ifdef VS_WARP_LDS_PAD
    __shared__ uint32_t lds_pad[VS_WARP_LDS_PAD / 4];       // occupancy experiments only: fewer workgroups per CU
    if (w < 0) lds_pad[threadIdx.x] = 0;
#endif
    // XCD-aware tile order, as in vs_k_bgr_warp_c3: every XCD walks one contiguous run of tiles in raster order
    const int tl = (int)(blockIdx.x & 7) * chunk + (int)(blockIdx.x >> 3);
    if (tl >= min(tiles_per_frame, (int)((blockIdx.x & 7) + 1) * chunk)) return;
#if VS_WARP_STAMPS
    unsigned long long stamp[STAMP_N] = {};
    VS_STAMP(0);
#endif
    const int frame = blockIdx.y;
    src += (size_t)frame * src_fs;
    dst += (size_t)frame * dst_fs;
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int tyi = tiles_x == 1 ? tl : (int)__umulhi((uint32_t)tl, tiles_x_magic);
    const int txi = tl - tyi * tiles_x;
    const int x0 = txi * WT_W, y0 = tyi * CV_TH;
    const int nx = min(WT_W, roi.w - x0), ny = min(CV_TH, roi.h - y0);             // live columns / rows of this tile (>= 1)
    const int x = x0 + lane;
    // this frame's tables at this tile (tab_w / tab_h are whole tiles: columns / rows beyond the window repeat its last column / row there,
    // masked at the store)
    const int* __restrict__ const adp = tab + (size_t)frame * (size_t)(2 * (tab_w + tab_h)) + x0;
    const int* __restrict__ const bdp = adp + tab_w;
    const int* __restrict__ const X0p = adp + (2 * tab_w - x0) + y0;
    const int* __restrict__ const Y0p = X0p + tab_h;
    const int ad = adp[lane], bd = bdp[lane];
    // source footprint: X0[y] and adelta[x] are monotone (cvRound of a monotone function), so the extremes sit at the tile's corners
    // (each table entry is within +-2^29 or the tile does not "fit": the sums below stay inside 32 bits, and a tile that passes is far from the
    // wrap in X0 + adelta and from saturate_cast<short>)
    const int adA = adp[0], adB = adp[nx - 1];
    const int bdA = bdp[0], bdB = bdp[nx - 1];
    const int XA = X0p[0], XB = X0p[ny - 1];
    const int YA = Y0p[0], YB = Y0p[ny - 1];
    // |X0|, |adelta| < 2^24 in 10-bit fixed point = 2^14 pixels each: the sums stay below 2^25 (a source coordinate below 2^15: saturate_cast<short> is
    // never reached inside a tile that fits), and the window origin folded into the lane's delta (base4 << 8, |base4| < 4 * 89 * 2^14) stays inside 32 bits.
    // (2^23 until round 6: frames beyond 8192 pixels fell off the tuned path, profiles/r06_cv_shape_sweep.txt)
    const int lim = 1 << 24;
    // (one test over all eight corners, no short circuit: the eight scalar loads then issue together and are waited for once)
    bool fits = max(max(max(abs(adA), abs(adB)), max(abs(bdA), abs(bdB))), max(max(abs(XA), abs(XB)), max(abs(YA), abs(YB)))) < lim;
    const int mnX = min(XA, XB) + min(adA, adB), mxX = max(XA, XB) + max(adA, adB);
    const int mnY = min(YA, YB) + min(bdA, bdB), mxY = max(YA, YB) + max(bdA, bdB);
    int sx_lo = 0, sy_lo = 0, rows = 0, groups = 0;
    if (fits) {
        sx_lo = (mnX >> 10) & ~3;                              // first staged column: a multiple of 4 pixels (12 bytes)
        const int sx_hi = (mxX >> 10) + 1;
        sy_lo = mnY >> 10;
        const int sy_hi = (mxY >> 10) + 1;
        rows = sy_hi - sy_lo + 1;
        groups = (sx_hi - sx_lo + 4) >> 2;
        fits = groups <= WS_W / 4 && rows <= CV_WS_H;
    }
    const bool src_aligned = ((((uintptr_t)src) | (uintptr_t)src_stride) & 3) == 0;                      // uniform
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    VS_STAMP(1);
    // (row offsets are 24-bit multiplies: a pitch of 2^24 bytes or more takes the rim path, whatever the frame's height)
    const bool interior = fits && src_aligned && sx_lo >= 0 && sx_lo + 4 * groups <= w && sy_lo >= 0 && sy_lo + rows <= h &&
                          (size_t)h * (size_t)src_stride < (1ull << 32) && src_stride < (1 << 24);            // uniform
    if (interior && !(VS_WARP_WHATIF & 2)) {
        // Interior tiles (the whole staged window inside an aligned frame: all but the frame's rim): every address is one 24-bit multiply-add
        // from a uniform base, no border tests.  (Clamping the items beyond the tile's own rows / column groups onto its last row / group
        // instead of predicating them was measured: the ~28 % redundant loads cost more than the branches, 12.1 us per 4K frame against 11.1.)
        const uint8_t* base = src + ((size_t)sy_lo * src_stride + (size_t)sx_lo * 3);
#if VS_WARP_CV_ROW_FILL
        // Item map of this path: lane -> (row r3 = lane / 20 of a row triplet, column group g = lane % 20), once per tile; slot s of wave wv
        // stages rows 3 (wv + 4 s) + r3 -- twelve rows further per slot: the source address advances by a UNIFORM 12 row pitches and the tile
        // address by a constant (an immediate of the ds_write), so a slot costs one compare beside its load and its four formatting
        // instructions (the 16-items-of-4-rows map of the rim path below: seven address instructions per slot).  60 of 64 lanes carry an item.
        static_assert(WS_W / 4 == 20 && CV_FILL_SLOTS >= (CV_WS_H + 11) / 12, "row-triplet item map");      // (a last, partial slot is cut by the row test)
        const uint32_t r3 = ((uint32_t)lane * 13u) >> 8, g = (uint32_t)lane - 20u * r3;                   // lane / 20, lane % 20 (lane < 64)
        const uint32_t row0 = 3u * (uint32_t)wv + r3;
        const bool col_live = r3 < 3u && (int)g < groups;
        const uint32_t goff = __umul24(row0, (uint32_t)src_stride) + 12u * g;
        uint32_t* const tp = tile_raw + (row0 * (uint32_t)CV_RS + 4u * g);
        u32x3 q[CV_FILL_SLOTS];
        bool live[CV_FILL_SLOTS];
#pragma unroll
        for (int s = 0; s < CV_FILL_SLOTS; s++) {              // every load is issued before the first tile write
            live[s] = col_live && (int)row0 < rows - 12 * s;
            if (live[s]) q[s] = *(const u32x3*)(base + (size_t)(12 * s) * (size_t)src_stride + goff);
        }
        VS_STAMP(2);
        VS_STAMP_DRAIN();
        VS_STAMP(3);
#pragma unroll
        for (int s = 0; s < CV_FILL_SLOTS; s++) {
            if (!live[s]) continue;
            u32x4 px;                                           // B0 G0 R0 B1 | G1 R1 B2 G2 | R2 B3 G3 R3  ->  four dwords B G R 0
            px.x = q[s].x & 0x00ffffffu;
            px.y = __builtin_amdgcn_perm(q[s].y, q[s].x, 0x0c050403u);
            px.z = __builtin_amdgcn_perm(q[s].z, q[s].y, 0x0c040302u);
            px.w = q[s].z >> 8;
            VS_BOUNDS_CHECK((int)((row0 + 12u * s) * CV_RS + 4u * g) + 3, CV_WS_H * CV_RS, 217);
            *(u32x4*)(tp + 12 * s * CV_RS) = px;
        }
#else
        u32x3 q[CV_FILL_SLOTS];
        uint32_t toff[CV_FILL_SLOTS];
        bool live[CV_FILL_SLOTS];
#pragma unroll
        for (int s = 0; s < CV_FILL_SLOTS; s++) {              // every load is issued before the first tile write
            const FillItem it = fill_item(lane, wv + 4 * s);
            live[s] = it.row < rows && it.g < groups;
            if (live[s]) q[s] = *(const u32x3*)(base + (__umul24((uint32_t)it.row, (uint32_t)src_stride) + 12u * (uint32_t)it.g));
            toff[s] = (uint32_t)it.row * (uint32_t)CV_RS + 4u * (uint32_t)it.g;
        }
        VS_STAMP(2);
        VS_STAMP_DRAIN();
        VS_STAMP(3);
#pragma unroll
        for (int s = 0; s < CV_FILL_SLOTS; s++) {
            if (!live[s]) continue;
            u32x4 px;                                           // B0 G0 R0 B1 | G1 R1 B2 G2 | R2 B3 G3 R3  ->  four dwords B G R 0
            px.x = q[s].x & 0x00ffffffu;
            px.y = __builtin_amdgcn_perm(q[s].y, q[s].x, 0x0c050403u);
            px.z = __builtin_amdgcn_perm(q[s].z, q[s].y, 0x0c040302u);
            px.w = q[s].z >> 8;
            VS_BOUNDS_CHECK((int)toff[s] + 3, CV_WS_H * CV_RS, 217);
            *(u32x4*)(tile_raw + VS_DEBUG_CLAMP((int)toff[s], CV_WS_H * CV_RS - 3)) = px;
        }
#endif
    } else if (fits && !(VS_WARP_WHATIF & 2)) {
        u32x3 q[CV_FILL_SLOTS];
        FillItem it[CV_FILL_SLOTS];
        bool live[CV_FILL_SLOTS], direct[CV_FILL_SLOTS];
#pragma unroll
        for (int s = 0; s < CV_FILL_SLOTS; s++) {              // every load is issued before the first tile write
            it[s] = fill_item(lane, wv + 4 * s);
            live[s] = it[s].row < rows && it[s].g < groups;
            const int sy = sy_lo + it[s].row, sx = sx_lo + 4 * it[s].g;
            direct[s] = live[s] && src_aligned && sx >= 0 && sx + 3 < w && sy >= 0 && sy < h;
            if (direct[s]) q[s] = *(const u32x3*)(src + (size_t)sy * src_stride + (size_t)sx * 3);
        }
#pragma unroll
        for (int s = 0; s < CV_FILL_SLOTS; s++) {
            if (!live[s]) continue;
            u32x4 px;
            if (direct[s]) {                                    // B0 G0 R0 B1 | G1 R1 B2 G2 | R2 B3 G3 R3  ->  four dwords B G R 0
                px.x = q[s].x & 0x00ffffffu;
                px.y = __builtin_amdgcn_perm(q[s].y, q[s].x, 0x0c050403u);
                px.z = __builtin_amdgcn_perm(q[s].z, q[s].y, 0x0c040302u);
                px.w = q[s].z >> 8;
            } else {                                            // the frame's rim, an unaligned frame: pixel by pixel, the border rule applied here
                const int sy = sy_lo + it[s].row, sx = sx_lo + 4 * it[s].g;
                uint32_t d[4];
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const int pxi = sx + k;
                    if (BORDER == 1 && (sy < 0 || sy >= h || pxi < 0 || pxi >= w)) d[k] = 0u;
                    else {
                        const uint8_t* qq = src + (size_t)clampi(sy, 0, h - 1) * src_stride + (size_t)clampi(pxi, 0, w - 1) * 3;
                        d[k] = (uint32_t)qq[0] | ((uint32_t)qq[1] << 8) | ((uint32_t)qq[2] << 16);
                    }
                }
                px = u32x4{d[0], d[1], d[2], d[3]};
            }
            VS_BOUNDS_CHECK(it[s].row * CV_RS + 4 * it[s].g + 3, CV_WS_H * CV_RS, 211);
            *(u32x4*)(tile_raw + VS_DEBUG_CLAMP(it[s].row * CV_RS + 4 * it[s].g, CV_WS_H * CV_RS - 3)) = px;
        }
    }
    VS_STAMP(4);
    __syncthreads();
    VS_STAMP(5);

    const int yw = y0 + wv * CV_RPW;                         // first row of this wave
    if (yw >= roi.h) return;                                 // wave-uniform
    if (VS_WARP_WHATIF & 128) { if (tile_raw[threadIdx.x] == 0x12345678u) dst[threadIdx.x] = 1; return; }   // (analysis: loads + fill only)
    const int m = lane & 3;
    const bool rows_aligned = ((((uintptr_t)dst) | (uintptr_t)dst_stride) & 3) == 0;                      // uniform
    const bool lane_in = x < roi.w, quad_in = (x | 3) < roi.w;
    const uint32_t sel = quad_sel(m);
    const uint32_t loff = (uint32_t)(x & ~3) * 3u + 4u * (uint32_t)m;
    // one output pixel from the byte tile: LDS byte address of staged pixel (sy, sx) = 4 * ((sy - sy_lo) * CV_RS + (sx - sx_lo)).
    // Instruction diet (tools/ubench_int.hip, profiles/r05_ubench_int.txt: three-operand integer instructions -- v_perm, v_dot2, v_mad_u32_u24,
    // v_bfe, v_add3, v_lshl_or, and v_mul_u32_u24 -- issue at 4.2-4.35 cycles per wave per SIMD, two-operand shifts / and / add at 2.3-2.6): the
    // weights are OpenCV's own 15-bit ones, 32 a b (<= 32768: a u16), so that fy enters as (Ys & 0x3e0) = 32 fy without a bit-field extract
    // and the top row's pair is (apair << 10) - bottom pair instead of a second multiply; result = (sum + 2^14) >> 15 as in the source.
    // (the window's origin is folded into the lane's column delta once per tile: base4 is a multiple of 4, so base4 << 8 leaves the ten
    // fraction bits of X alone and ((X + (base4 << 8)) >> 8) & ~3 = 4 sx + base4: the tile offset is one multiply-add per pixel)
    const int base4 = -4 * (sy_lo * CV_RS + sx_lo);
    const uint32_t adw = (uint32_t)ad + ((uint32_t)base4 << 8);
    auto sample = [&](uint32_t Xs, uint32_t Ys) -> uint32_t {                   // Xs = X0 + adw (fits tiles), Ys = Y0 + bd
        const uint32_t fx = (Xs >> 5) & 31u, fy32 = Ys & 0x3e0u;
        const int off = VS_DEBUG_CLAMP_BYTES(((int)Ys >> 10) * (4 * CV_RS) + (((int)Xs >> 8) & ~3), 4 * (CV_WS_H * CV_RS - (CV_RS + 2)), 212);
        const __attribute__((address_space(3))) uint32_t* t = (const __attribute__((address_space(3))) uint32_t*)((const __attribute__((address_space(3))) char*)tile_raw + off);
        if (VS_WARP_WHATIF & 1) return t[0] + fx + fy32;                    // (analysis: one LDS read, no arithmetic)
        const uint32_t p00 = t[0], p01 = t[1], p10 = t[CV_RS], p11 = t[CV_RS + 1];
#if VS_WARP_CV_W16
        // 16-bit weights 64 a b = twice OpenCV's: (2 S + 2^15) >> 16 is (S + 2^14) >> 15, and the sample is byte 2 of the sum -- no shifts in front
        // of the pack.  Only the top-left weight can reach 2^16 (fx = fy = 0: the other three are 0); the packed multiply saturates it to 65535, and
        // (65535 v + 2^15) >> 16 = v for v <= 255.
        const uint32_t apair = fx * 0x1fffeu + 64u;                         // 2 (32 - fx) | 2 fx << 16
        const uint32_t wb = pk_mul_lo_u16_lo(apair, fy32), wt = pk_mul_lo_u16_lo_sat(apair, 1024u - fy32);
        constexpr uint32_t kHalf = 1u << 15;
#else
        const uint32_t apair = fx * 0xffffu + 32u;                          // (32 - fx) | fx << 16
        const uint32_t wb = apair * fy32, wt = (apair << 10) - wb;          // {32 a0 b1 | 32 a1 b1 << 16}, {32 a0 b0 | 32 a1 b0 << 16}: each <= 32768, no borrow between the halves
        constexpr uint32_t kHalf = 1u << 14;
#endif
        uint32_t o[3];
#pragma unroll
        for (int c = 0; c < 3; c++) {
            const uint32_t selc = 0x0c040c00u + 0x00010001u * (uint32_t)c;  // {byte c of the left pixel, 0, byte c of the right pixel, 0}
            const uint32_t top = __builtin_amdgcn_perm(p01, p00, selc), bot = __builtin_amdgcn_perm(p11, p10, selc);
            o[c] = udot2(bot, wb, udot2(top, wt, kHalf));                   // bits 15..22 (16..23 with the doubled weights) = the sample
        }
#if VS_WARP_CV_W16
        return __builtin_amdgcn_perm(o[2], __builtin_amdgcn_perm(o[1], o[0], 0x0c0c0602u), 0x0c060100u);     // {B, G, R, 0}
#endif
        // {B, G, R, 0}: shifts put each sample on a byte boundary (B: byte 0 of o0 >> 15, G: byte 1 of o1 >> 7, R: byte 2 of o2 << 1), two v_perm pick them
        const uint32_t bg = __builtin_amdgcn_perm(o[1] >> 7, o[0] >> 15, 0x0c0c0500u);
        return __builtin_amdgcn_perm(o[2] << 1, bg, 0x0c060100u);
    };
    if (fits && rows_aligned && nx == WT_W && yw + CV_RPW <= roi.h && (size_t)roi.h * (size_t)dst_stride < (1ull << 32)) {
        // the common case -- the tile fits its window, whole quads, whole rows: all rows are sampled in ONE basic block (the stores sit behind
        // a single lane mask afterwards), so that the scheduler can run the rows' LDS reads ahead of the previous rows' arithmetic
        uint32_t roff = (uint32_t)yw * (uint32_t)dst_stride + loff;   // one 32-bit byte offset per lane from the frame's uniform base, advanced by the row pitch
#pragma unroll 1
        for (int k0 = 0; k0 < CV_RPW; k0 += CV_RBK) {
            uint32_t d[CV_RBK];
#pragma unroll
            for (int k = 0; k < CV_RBK; k++)
                d[k] = quad_pack_bgr(sample((uint32_t)X0p[wv * CV_RPW + k0 + k] + adw, (uint32_t)Y0p[wv * CV_RPW + k0 + k] + (uint32_t)bd), sel);
            if (m < 3 && (!(VS_WARP_WHATIF & 8) || d[0] == 0x12345678u)) {
#pragma unroll
                for (int k = 0; k < CV_RBK; k++, roff += (uint32_t)dst_stride) VS_STORE32((uint32_t*)(dst + roff), d[k]);
            } else roff += (uint32_t)CV_RBK * (uint32_t)dst_stride;
        }
#if VS_WARP_STAMPS
        VS_STAMP(6);
        VS_STAMP_DRAIN();
        VS_STAMP(7);
        {
            const unsigned wg = blockIdx.y * gridDim.x + blockIdx.x;
            stamp[8] = __builtin_amdgcn_s_getreg(4 | (31 << 11));              // HW_REG_HW_ID
            stamp[9] = __builtin_amdgcn_s_getreg(20 | (31 << 11));             // HW_REG_XCC_ID
            if (lane == 0 && wg < (unsigned)STAMP_WGS && interior)
                for (int i = 0; i < STAMP_N; i++) g_warp_stamps[((size_t)wg * 4 + wv) * STAMP_N + i] = stamp[i];
        }
#endif
        return;
    }
#pragma unroll 1
    for (int k = 0; k < CV_RPW; k++) {
        const int y = yw + k;
        if (y >= roi.h) break;                               // wave-uniform
        const uint32_t Xs = (uint32_t)X0p[wv * CV_RPW + k] + (uint32_t)ad, Ys = (uint32_t)Y0p[wv * CV_RPW + k] + (uint32_t)bd;
        uint32_t p = 0u;
        if (!fits) {
            uint32_t o[3] = {0u, 0u, 0u};
            if (lane_in) cv_pixel_global<BORDER>(src, w, h, src_stride, (int)Xs >> 5, (int)Ys >> 5, o);
            p = o[0] | (o[1] << 8) | (o[2] << 16);
        } else p = sample(Xs + ((uint32_t)base4 << 8), Ys);
        const uint32_t d = quad_pack_bgr(p, sel);            // every lane of the wave takes part in the shuffle
        uint8_t* orow = dst + (size_t)y * dst_stride;
        if (rows_aligned && quad_in) {
            if (m < 3) VS_STORE32((uint32_t*)(orow + loff), d);
        } else if (lane_in) {
            orow[(size_t)x * 3] = (uint8_t)p;
            orow[(size_t)x * 3 + 1] = (uint8_t)(p >> 8);
            orow[(size_t)x * 3 + 2] = (uint8_t)(p >> 16);
        }
    }
}

// (Measured and dropped, round 6 -- commit a01891c, profiles/r06_warp_cv.md section 4: 128 x 64 output tiles, two 64-column halves per 512-thread workgroup on ONE
// staged window of 144 x 72 pixels -- halo 1.11x instead of 1.18x, 43 KB of LDS, 24 waves per CU as before, the per-wave sampler unchanged.  Bit-identical,
// 11.7 us per 4K frame against 10.9: the wider workgroup's fill (eight waves behind one barrier) is covered worse by three workgroups per CU than by six.)

// ------------------------------------------------------------------------------------------------------------------------------------
// VS_WARP_BILINEAR_CV on interleaved 16-bit containers (10 / 12 / 16-bit BGR): OpenCV's remapBilinear<Cast<float, ushort>> -- the same
// fixed-point coordinates as the 8-bit path, FLOAT weights a b / 1024 (exact), result = cvRound(v00 w00 + v01 w01 + v10 w10 + v11 w11) with the
// products and sums in float, left to right.  While every sample of a tile is below 2^14 each product (14 + 10 bits) and each partial sum
// (<= the largest sample, 10 fraction bits) is EXACT in fp32, so the float expression equals the integer one: S = sum 32 a b v (< 2^29),
// result = round-half-even(S / 2^15) -- and the tile takes the 8-bit kernel's v_perm / v_dot2_u32_u16 sampler on a word tile (8 bytes
// {B | G << 16, R} per staged pixel: the float bilinear kernel's).  A tile that holds a sample >= 2^14 (full 16-bit content) evaluates the
// float expression as written.  Tile 64 x 32 (28 KB of LDS); tables, footprint, stores as in the 8-bit kernel.
// ------------------------------------------------------------------------------------------------------------------------------------
#ifndef VS_WARP_CV16_TILE_H
#define VS_WARP_CV16_TILE_H 32           // output rows per workgroup of the 16-bit kernel (a multiple of 4)
#endif
#ifndef VS_WARP_CV16_MINWAVES
#define VS_WARP_CV16_MINWAVES 6
#endif
#ifndef VS_WARP_CV16_WS_EXTRA
#define VS_WARP_CV16_WS_EXTRA 8          // staged rows beyond the tile's own (a multiple of 4)
#endif
constexpr int CV16_TH = VS_WARP_CV16_TILE_H, CV16_RPW = CV16_TH / 4, CV16_WS_H = CV16_TH + VS_WARP_CV16_WS_EXTRA;
#ifndef VS_WARP_CV16_RS
#define VS_WARP_CV16_RS WS_W             // row pitch of the word tile in staged pixels (8 bytes each; >= WS_W, a multiple of 4).  80: 25.6 KB of LDS, SIX workgroups per CU (round 6);
                                         // 88 (until round 6): 28.2 KB, five -- 20.27 us per 4K 10-bit frame against 19.89 (four alternating passes, profiles/r06_ab_cv16_pitch.txt);
                                         // seven (36 staged rows) 20.03: not kept
#endif
constexpr int CV16_RS = VS_WARP_CV16_RS;
static_assert(CV16_RS >= WS_W && CV16_RS % 4 == 0, "tile pitch");
static_assert(CV16_TH % 4 == 0 && CV16_TH >= 16, "four waves share a tile's rows");
constexpr int CV16_FILL_SLOTS = (CV16_WS_H / 4 * (WS_W / 4) + 63) / 64;

template <int BORDER>
__device__ __forceinline__ void cv_pixel_global_u16(const uint16_t* __restrict__ src, int w, int h, int stride, int X, int Y, int maxv, uint32_t out[3]) {
    const int sx = clampi(X >> 5, -32768, 32767), sy = clampi(Y >> 5, -32768, 32767);      // saturate_cast<short>
    const int a1 = X & 31, b1 = Y & 31, a0 = 32 - a1, b0 = 32 - b1;
    auto tap = [&](int xx, int yy, int c) -> float {
        if (BORDER == 1) { if (xx < 0 || yy < 0 || xx >= w || yy >= h) return 0.0f; }
        else { xx = clampi(xx, 0, w - 1); yy = clampi(yy, 0, h - 1); }
        return (float)src[(size_t)yy * stride + (size_t)xx * 3 + c];
    };
    const float k = 1.0f / 1024.0f;
    const float w00 = (float)(a0 * b0) * k, w01 = (float)(a1 * b0) * k, w10 = (float)(a0 * b1) * k, w11 = (float)(a1 * b1) * k;
#pragma unroll
    for (int c = 0; c < 3; c++) {
        const float sum = tap(sx, sy, c) * w00 + tap(sx + 1, sy, c) * w01 + tap(sx, sy + 1, c) * w10 + tap(sx + 1, sy + 1, c) * w11;
        out[c] = (uint32_t)min(max((int)rintf(sum), 0), maxv);
    }
}

template <int BORDER>
__global__ __launch_bounds__(256, VS_WARP_CV16_MINWAVES) void vs_k_bgr_warp_cv_c3_u16(const uint16_t* __restrict__ src, int w, int h, int src_stride,
                                                                 const int* __restrict__ tab, int tab_w, int tab_h, uint16_t* __restrict__ dst, int dst_stride,
                                                                 size_t src_fs, size_t dst_fs, int tiles_x, uint32_t tiles_x_magic, int tiles_per_frame,
                                                                 int chunk, int maxv, vsk::Roi roi) {
    __shared__ __attribute__((aligned(16))) uint32_t tile_raw[CV16_WS_H * CV16_RS * 2];      // {B | G << 16, R} per staged source pixel
    const int tl = (int)(blockIdx.x & 7) * chunk + (int)(blockIdx.x >> 3);
    if (tl >= min(tiles_per_frame, (int)((blockIdx.x & 7) + 1) * chunk)) return;
    const int frame = blockIdx.y;
    src += (size_t)frame * src_fs;
    dst += (size_t)frame * dst_fs;
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int tyi = tiles_x == 1 ? tl : (int)__umulhi((uint32_t)tl, tiles_x_magic);
    const int txi = tl - tyi * tiles_x;
    const int x0 = txi * WT_W, y0 = tyi * CV16_TH;
    const int nx = min(WT_W, roi.w - x0), ny = min(CV16_TH, roi.h - y0);
    const int x = x0 + lane;
    // (the per-frame tables of vs_k_cv_tables, read as in the 8-bit kernel: corners and row origins by scalar loads, the lane's deltas by vector loads)
    const int* __restrict__ const adp = tab + (size_t)frame * (size_t)(2 * (tab_w + tab_h)) + x0;
    const int* __restrict__ const bdp = adp + tab_w;
    const int* __restrict__ const X0p = adp + (2 * tab_w - x0) + y0;
    const int* __restrict__ const Y0p = X0p + tab_h;
    const int ad = adp[lane], bd = bdp[lane];
    const int adA = adp[0], adB = adp[nx - 1];
    const int bdA = bdp[0], bdB = bdp[nx - 1];
    const int XA = X0p[0], XB = X0p[ny - 1];
    const int YA = Y0p[0], YB = Y0p[ny - 1];
    const int lim = 1 << 24;                                 // (as in the 8-bit kernel: every entry within 2^14 pixels, the sums and base8 << 7 inside 32 bits)
    bool fits = max(max(max(abs(adA), abs(adB)), max(abs(bdA), abs(bdB))), max(max(abs(XA), abs(XB)), max(abs(YA), abs(YB)))) < lim;
    const int mnX = min(XA, XB) + min(adA, adB), mxX = max(XA, XB) + max(adA, adB);
    const int mnY = min(YA, YB) + min(bdA, bdB), mxY = max(YA, YB) + max(bdA, bdB);
    int sx_lo = 0, sy_lo = 0, rows = 0, groups = 0;
    if (fits) {
        sx_lo = (mnX >> 10) & ~3;
        const int sx_hi = (mxX >> 10) + 1;
        sy_lo = mnY >> 10;
        const int sy_hi = (mxY >> 10) + 1;
        rows = sy_hi - sy_lo + 1;
        groups = (sx_hi - sx_lo + 4) >> 2;
        fits = groups <= WS_W / 4 && rows <= CV16_WS_H;
    }
    const bool src_aligned = ((((uintptr_t)src) | ((uintptr_t)src_stride * 2)) & 3) == 0;                // uniform
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    uint32_t seen = 0;                                       // OR of every staged sample (as packed pairs): bits 14 / 15 of a half set = a sample >= 2^14
    const bool interior = fits && src_aligned && sx_lo >= 0 && sx_lo + 4 * groups <= w && sy_lo >= 0 && sy_lo + rows <= h &&
                          (size_t)h * (size_t)src_stride * 2 < (1ull << 32) && src_stride < (1 << 23);       // uniform (24-bit row-offset multiplies)
    if (interior) {                                          // (as in the 8-bit kernel: one 24-bit multiply-add per address from a uniform base, no border tests)
        const uint8_t* base = (const uint8_t*)(src + ((size_t)sy_lo * src_stride + (size_t)sx_lo * 3));
#if VS_WARP_CV_ROW_FILL
        // (the 8-bit kernel's row-triplet item map: lane -> (row lane / 20 of a triplet, column group lane % 20), slots twelve rows apart)
        static_assert(WS_W / 4 == 20 && CV16_FILL_SLOTS == (CV16_WS_H + 11) / 12, "row-triplet item map");
        const uint32_t r3 = ((uint32_t)lane * 13u) >> 8, g = (uint32_t)lane - 20u * r3;
        const uint32_t row0 = 3u * (uint32_t)wv + r3;
        const bool col_live = r3 < 3u && (int)g < groups;
        const uint32_t goff = __umul24(row0, 2u * (uint32_t)src_stride) + 24u * g;
        uint32_t* const tp = tile_raw + 2u * (row0 * (uint32_t)CV16_RS + 4u * g);
        u32x3 qa[CV16_FILL_SLOTS], qb[CV16_FILL_SLOTS];
        bool live[CV16_FILL_SLOTS];
#pragma unroll
        for (int s = 0; s < CV16_FILL_SLOTS; s++) {
            live[s] = col_live && (int)row0 < rows - 12 * s;
            if (live[s]) {
                const uint8_t* gp = base + (size_t)(12 * s) * (2 * (size_t)src_stride) + goff;
                qa[s] = *(const u32x3*)gp;
                qb[s] = *(const u32x3*)(gp + 12);
            }
        }
#pragma unroll
        for (int s = 0; s < CV16_FILL_SLOTS; s++) {
            if (!live[s]) continue;
            const u32x3 a = qa[s], b = qb[s];                   // a = B0G0 R0B1 G1R1 ; b = B2G2 R2B3 G3R3
            seen |= a.x | a.y | a.z | b.x | b.y | b.z;
            VS_BOUNDS_CHECK((int)(2u * ((row0 + 12u * s) * CV16_RS + 4u * g)) + 7, CV16_WS_H * CV16_RS * 2, 218);
            u32x4* dstp = (u32x4*)(tp + 2 * 12 * s * CV16_RS);
            dstp[0] = u32x4{a.x, a.y & 0xffffu, __builtin_amdgcn_alignbyte(a.z, a.y, 2), a.z >> 16};
            dstp[1] = u32x4{b.x, b.y & 0xffffu, __builtin_amdgcn_alignbyte(b.z, b.y, 2), b.z >> 16};
        }
#else
        u32x3 qa[CV16_FILL_SLOTS], qb[CV16_FILL_SLOTS];
        uint32_t toff[CV16_FILL_SLOTS];
        bool live[CV16_FILL_SLOTS];
#pragma unroll
        for (int s = 0; s < CV16_FILL_SLOTS; s++) {
            const FillItem it = fill_item(lane, wv + 4 * s);
            live[s] = it.row < rows && it.g < groups;
            if (live[s]) {
                const uint8_t* gp = base + (__umul24((uint32_t)it.row, 2u * (uint32_t)src_stride) + 24u * (uint32_t)it.g);
                qa[s] = *(const u32x3*)gp;
                qb[s] = *(const u32x3*)(gp + 12);
            }
            toff[s] = 2u * ((uint32_t)it.row * (uint32_t)CV16_RS + 4u * (uint32_t)it.g);
        }
#pragma unroll
        for (int s = 0; s < CV16_FILL_SLOTS; s++) {
            if (!live[s]) continue;
            const u32x3 a = qa[s], b = qb[s];                   // a = B0G0 R0B1 G1R1 ; b = B2G2 R2B3 G3R3
            seen |= a.x | a.y | a.z | b.x | b.y | b.z;
            VS_BOUNDS_CHECK((int)toff[s] + 7, CV16_WS_H * CV16_RS * 2, 218);
            u32x4* dstp = (u32x4*)(tile_raw + VS_DEBUG_CLAMP((int)toff[s], CV16_WS_H * CV16_RS * 2 - 7));
            dstp[0] = u32x4{a.x, a.y & 0xffffu, __builtin_amdgcn_alignbyte(a.z, a.y, 2), a.z >> 16};
            dstp[1] = u32x4{b.x, b.y & 0xffffu, __builtin_amdgcn_alignbyte(b.z, b.y, 2), b.z >> 16};
        }
#endif
    } else if (fits) {
        u32x3 qa[CV16_FILL_SLOTS], qb[CV16_FILL_SLOTS];
        FillItem it[CV16_FILL_SLOTS];
        bool live[CV16_FILL_SLOTS], direct[CV16_FILL_SLOTS];
#pragma unroll
        for (int s = 0; s < CV16_FILL_SLOTS; s++) {
            it[s] = fill_item(lane, wv + 4 * s);
            live[s] = it[s].row < rows && it[s].g < groups;
            const int sy = sy_lo + it[s].row, sx = sx_lo + 4 * it[s].g;
            direct[s] = live[s] && src_aligned && sx >= 0 && sx + 3 < w && sy >= 0 && sy < h;
            if (direct[s]) {
                const uint16_t* gp = src + (size_t)sy * src_stride + (size_t)sx * 3;
                qa[s] = *(const u32x3*)gp;
                qb[s] = *(const u32x3*)(gp + 6);
            }
        }
#pragma unroll
        for (int s = 0; s < CV16_FILL_SLOTS; s++) {
            if (!live[s]) continue;
            u32x4 lo, hi;                                       // pixels 0, 1 and 2, 3 of the group as {B | G << 16, R}
            if (direct[s]) {
                const u32x3 a = qa[s], b = qb[s];               // a = B0G0 R0B1 G1R1 ; b = B2G2 R2B3 G3R3
                lo = u32x4{a.x, a.y & 0xffffu, __builtin_amdgcn_alignbyte(a.z, a.y, 2), a.z >> 16};
                hi = u32x4{b.x, b.y & 0xffffu, __builtin_amdgcn_alignbyte(b.z, b.y, 2), b.z >> 16};
                seen |= a.x | a.y | a.z | b.x | b.y | b.z;
            } else {
                const int sy = sy_lo + it[s].row, sx = sx_lo + 4 * it[s].g;
                uint32_t d[4][2];
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const int pxi = sx + k;
                    if (BORDER == 1 && (sy < 0 || sy >= h || pxi < 0 || pxi >= w)) d[k][0] = d[k][1] = 0u;
                    else {
                        const uint16_t* qq = src + (size_t)clampi(sy, 0, h - 1) * src_stride + (size_t)clampi(pxi, 0, w - 1) * 3;
                        d[k][0] = (uint32_t)qq[0] | ((uint32_t)qq[1] << 16);
                        d[k][1] = (uint32_t)qq[2];
                    }
                    seen |= d[k][0] | d[k][1];
                }
                lo = u32x4{d[0][0], d[0][1], d[1][0], d[1][1]};
                hi = u32x4{d[2][0], d[2][1], d[3][0], d[3][1]};
            }
            VS_BOUNDS_CHECK((it[s].row * CV16_RS + 4 * it[s].g + 3) * 2 + 1, CV16_WS_H * CV16_RS * 2, 215);
            u32x4* dstp = (u32x4*)(tile_raw + 2 * VS_DEBUG_CLAMP(it[s].row * CV16_RS + 4 * it[s].g, CV16_WS_H * CV16_RS - 3));
            dstp[0] = lo;
            dstp[1] = hi;
        }
    }
    // (barrier) ... and does ANY staged sample of the tile reach 2^14?  Then the float expression is not exact and is evaluated as written.
    const bool wide = __syncthreads_or((int)((seen & 0xc000c000u) != 0u)) != 0;

    const int yw = y0 + wv * CV16_RPW;
    if (yw >= roi.h) return;                                 // wave-uniform
    const bool rows_aligned = ((((uintptr_t)dst) | ((uintptr_t)dst_stride * 2)) & 3) == 0;               // uniform
    const bool lane_in = x < roi.w, pair_in = (x | 1) < roi.w;
    const int base8 = -8 * (sy_lo * CV16_RS + sx_lo);       // LDS byte address of staged pixel (sy, sx) = 8 * ((sy - sy_lo) * CV16_RS + (sx - sx_lo))
    typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
    // one output pixel of a tile that fits (Xs carries the window origin: base8 is a multiple of 8, base8 << 7 leaves X's fraction bits alone)
    auto sample = [&](uint32_t Xs, uint32_t Ys, uint32_t (&o)[3]) {
        const uint32_t fx = (Xs >> 5) & 31u, fy = (Ys >> 5) & 31u;
        const int off = VS_DEBUG_CLAMP_BYTES(((int)Ys >> 10) * (8 * CV16_RS) + (((int)Xs >> 7) & ~7), 8 * (CV16_WS_H * CV16_RS - (CV16_RS + 2)), 216);
        const __attribute__((address_space(3))) u32x2* t = (const __attribute__((address_space(3))) u32x2*)((const __attribute__((address_space(3))) char*)tile_raw + off);
        const u32x2 p00 = t[0], p01 = t[1], p10 = t[CV16_RS], p11 = t[CV16_RS + 1];
        if (!wide) {                                         // (uniform) every sample < 2^14: the integer form of the same value
            const uint32_t apair = fx * 0xffffu + 32u;                      // (32 - fx) | fx << 16
            const uint32_t wb = apair * (fy << 5), wt = (apair << 10) - wb; // {32 a0 b | 32 a1 b << 16}
            const uint32_t top[3] = {__builtin_amdgcn_perm(p01.x, p00.x, 0x05040100u), __builtin_amdgcn_perm(p01.x, p00.x, 0x07060302u), __builtin_amdgcn_perm(p01.y, p00.y, 0x05040100u)};
            const uint32_t bot[3] = {__builtin_amdgcn_perm(p11.x, p10.x, 0x05040100u), __builtin_amdgcn_perm(p11.x, p10.x, 0x07060302u), __builtin_amdgcn_perm(p11.y, p10.y, 0x05040100u)};
#pragma unroll
            for (int c = 0; c < 3; c++) {
                const uint32_t S = udot2(bot[c], wb, udot2(top[c], wt, 0u));         // = 2^15 x the float sum, exactly
                o[c] = min((S + 16383u + ((S >> 15) & 1u)) >> 15, (uint32_t)maxv);   // cvRound: half to even
            }
        } else {
            const float kk = 1.0f / 1024.0f;
            const float w00 = (float)((32u - fx) * (32u - fy)) * kk, w01 = (float)(fx * (32u - fy)) * kk, w10 = (float)((32u - fx) * fy) * kk, w11 = (float)(fx * fy) * kk;
            const uint32_t v00[3] = {p00.x & 0xffffu, p00.x >> 16, p00.y & 0xffffu}, v01[3] = {p01.x & 0xffffu, p01.x >> 16, p01.y & 0xffffu};
            const uint32_t v10[3] = {p10.x & 0xffffu, p10.x >> 16, p10.y & 0xffffu}, v11[3] = {p11.x & 0xffffu, p11.x >> 16, p11.y & 0xffffu};
#pragma unroll
            for (int c = 0; c < 3; c++) {
                const float sum = (float)v00[c] * w00 + (float)v01[c] * w01 + (float)v10[c] * w10 + (float)v11[c] * w11;
                o[c] = (uint32_t)min(max((int)rintf(sum), 0), maxv);
            }
        }
    };
    if (fits && rows_aligned && nx == WT_W && yw + CV16_RPW <= roi.h && (size_t)roi.h * (size_t)dst_stride * 2 < (1ull << 32)) {
        // the common case: the wave's rows in one basic block, the stores behind two lane masks afterwards (even lanes two dwords, odd lanes one),
        // at a 32-bit byte offset from the frame's base advanced by the row pitch
        const uint32_t adw = (uint32_t)ad + ((uint32_t)base8 << 7);
        uint32_t d0[CV16_RPW], d1[CV16_RPW];
#pragma unroll
        for (int k = 0; k < CV16_RPW; k++) {
            uint32_t o[3];
            sample((uint32_t)X0p[wv * CV16_RPW + k] + adw, (uint32_t)Y0p[wv * CV16_RPW + k] + (uint32_t)bd, o);
            pair_pack_bgr16(o, x & 1, d0[k], d1[k]);
        }
        uint32_t roff = (uint32_t)yw * (2u * (uint32_t)dst_stride) + (uint32_t)(x & ~1) * 6u;      // 12 bytes per pixel pair
        if (x & 1) {
#pragma unroll
            for (int k = 0; k < CV16_RPW; k++) VS_STORE32((uint32_t*)((uint8_t*)dst + roff + (uint32_t)k * (2u * (uint32_t)dst_stride) + 8u), d0[k]);
        } else {
#pragma unroll
            for (int k = 0; k < CV16_RPW; k++) {
                VS_STORE32((uint32_t*)((uint8_t*)dst + roff + (uint32_t)k * (2u * (uint32_t)dst_stride)), d0[k]);
                VS_STORE32((uint32_t*)((uint8_t*)dst + roff + (uint32_t)k * (2u * (uint32_t)dst_stride) + 4u), d1[k]);
            }
        }
        return;
    }
#pragma unroll 2
    for (int k = 0; k < CV16_RPW; k++) {
        const int y = yw + k;
        if (y >= roi.h) break;                               // wave-uniform
        const uint32_t Xs = (uint32_t)X0p[wv * CV16_RPW + k] + (uint32_t)ad, Ys = (uint32_t)Y0p[wv * CV16_RPW + k] + (uint32_t)bd;
        uint32_t o[3] = {0u, 0u, 0u};
        if (!fits) {
            if (lane_in) cv_pixel_global_u16<BORDER>(src, w, h, src_stride, (int)Xs >> 5, (int)Ys >> 5, maxv, o);
        } else sample(Xs + ((uint32_t)base8 << 7), Ys, o);
        uint32_t d0, d1;
        pair_pack_bgr16(o, x & 1, d0, d1);
        uint16_t* orow = dst + (size_t)y * dst_stride;
        if (rows_aligned && pair_in) {
            uint32_t* q = (uint32_t*)(orow + (size_t)(x & ~1) * 3);   // 12 bytes per pixel pair
            if (x & 1) VS_STORE32(q + 2, d0);
            else { VS_STORE32(q, d0); VS_STORE32(q + 1, d1); }
        } else if (lane_in) {
            orow[(size_t)x * 3] = (uint16_t)o[0];
            orow[(size_t)x * 3 + 1] = (uint16_t)o[1];
            orow[(size_t)x * 3 + 2] = (uint16_t)o[2];
        }
    }
}

}  // namespace

VS_BOUNDS_TU(vs_bounds_fetch_warp)

#if VS_WARP_STAMPS
// (analysis build only) copies the stamps of the last launches out and clears them; n = capacity in 64-bit words
extern "C" __attribute__((visibility("default"))) int vs_debug_warp_stamps(unsigned long long* out, size_t n) {
    const size_t total = (size_t)STAMP_WGS * 4 * STAMP_N;
    if (n < total) return -1;
    if (hipDeviceSynchronize() != hipSuccess) return -2;
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_warp_stamps), total * sizeof(unsigned long long)) != hipSuccess) return -3;
    static std::vector<unsigned long long> zero(total, 0ULL);
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_warp_stamps), zero.data(), total * sizeof(unsigned long long)) != hipSuccess) return -4;
    return STAMP_WGS;
}
#endif

namespace vsk {

template <typename T>
static hipError_t launch_c3(const T* src, int w, int h, int src_stride, const float4* params_dev, const float4* extents_dev, int mode,
                            int border, T* dst, int dst_stride, int n_frames, size_t src_fs, size_t dst_fs, float maxv, Roi roi, int compact, hipStream_t s) {
    const int th = tile_h_of((int)sizeof(T) * 8, mode);
    const int tiles_x = (roi.w + WT_W - 1) / WT_W, tiles_y = (roi.h + th - 1) / th;
    const long long tpf = (long long)tiles_x * tiles_y;
    // the kernel's tl / tiles_x is a multiply-high, exact while tl * tiles_x < 2^32: wider windows take the generic kernel
    if (tpf > 0x3fffffLL || tpf * tiles_x >= (1LL << 32)) return hipErrorNotSupported;
    const int chunk = (int)((tpf + 7) / 8);
    // tl / tiles_x == (tl * magic) >> 32 for every tl with tl * tiles_x < 2^32 (tiles_x >= 2; the kernel special-cases 1)
    const uint32_t magic = (uint32_t)(0x100000000ULL / (uint32_t)tiles_x) + 1u;
    for (int f0 = 0; f0 < n_frames; f0 += 65535) {         // gridDim.y limit
        const int nf = n_frames - f0 < 65535 ? n_frames - f0 : 65535;
        const T* sp = src + (size_t)f0 * src_fs;
        T* dp = dst + (size_t)f0 * dst_fs;
        const float4* pp = params_dev + f0;
        const float4* ep = extents_dev ? extents_dev + f0 : nullptr;
        const int nt_wg = mode == 1 ? VS_WARP_TILES_PER_WG_BILINEAR : VS_WARP_TILES_PER_WG;     // tiles a workgroup walks (the kernel's NT)
        dim3 grid((unsigned)((chunk + nt_wg - 1) / nt_wg * 8), (unsigned)nf), block(256);
#define VS_LAUNCH(M, Bd) \
        hipLaunchKernelGGL((vs_k_bgr_warp_c3<T, M, Bd>), grid, block, 0, s, sp, w, h, src_stride, pp, dp, dst_stride, src_fs, dst_fs, \
                           tiles_x, magic, (int)tpf, chunk, maxv, roi, ep)
        if (mode == 0 && border == 0) VS_LAUNCH(0, 0);
        else if (mode == 0) VS_LAUNCH(0, 1);
#define VS_LAUNCH_C(M, Bd, Sh) \
        hipLaunchKernelGGL((vs_k_bgr_warp_c3<T, M, Bd, Sh>), grid, block, 0, s, sp, w, h, src_stride, pp, dp, dst_stride, src_fs, dst_fs, \
                           tiles_x, magic, (int)tpf, chunk, maxv, roi, ep)
        else if (mode == 2 && compact && border == 0) VS_LAUNCH_C(2, 0, 1);
        else if (mode == 2 && compact) VS_LAUNCH_C(2, 1, 1);
        else if (mode == 3 && compact == 2 && border == 0) VS_LAUNCH_C(3, 0, 2);
        else if (mode == 3 && compact == 2) VS_LAUNCH_C(3, 1, 2);
        else if (mode == 3 && compact && border == 0) VS_LAUNCH_C(3, 0, 1);
        else if (mode == 3 && compact) VS_LAUNCH_C(3, 1, 1);
        else if (mode == 2 && border == 0) VS_LAUNCH(2, 0);
        else if (mode == 2) VS_LAUNCH(2, 1);
        else if (mode == 3 && border == 0) VS_LAUNCH(3, 0);
        else if (mode == 3) VS_LAUNCH(3, 1);
        else if (border == 0) VS_LAUNCH(1, 0);
        else VS_LAUNCH(1, 1);
    }
#undef VS_LAUNCH
#undef VS_LAUNCH_C
    return hipGetLastError();
}

hipError_t bgr_warp_c3(const void* src, int w, int h, int src_stride, int bits, const float4* params_dev, const float4* extents_dev,
                       int mode, int border, int max_value, void* dst, int dst_stride, int n_frames, size_t src_fs, size_t dst_fs, Roi roi,
                       int compact, hipStream_t s) {
    static const int compact_env = []() { const char* e = getenv("VS_WARP_COMPACT"); return e ? atoi(e) : -1; }();       // (A/B switch: 0 never, 1 the six-wave shape at most, 2 / unset: whatever the extents allow)
    if (extents_dev == nullptr || !(mode == 2 || mode == 3)) compact = 0;
    if (compact_env >= 0) compact = std::min(compact, compact_env);
    if (bits == 8)
        return launch_c3<uint8_t>((const uint8_t*)src, w, h, src_stride, params_dev, extents_dev, mode, border, (uint8_t*)dst, dst_stride,
                                  n_frames, src_fs, dst_fs, (float)max_value, roi, compact, s);
    return launch_c3<uint16_t>((const uint16_t*)src, w, h, src_stride, params_dev, extents_dev, mode, border, (uint16_t*)dst, dst_stride,
                               n_frames, src_fs, dst_fs, (float)max_value, roi, compact, s);
}

// -> 0: the standard window; 1: every frame's tile footprint stays under 16 source rows (E4 = bgr_warp_c3_extents' output): the 20-row windows hold floor(max) - floor(min) + 4 <= 20
// rows for every tile; 2: ... and under 65 source columns: floor(max) - floor(min) + 4 taps + 3 of alignment <= 72 = the 18 column groups of the seven-wave shape
int bgr_warp_c3_compact_shape(const float* E4, int n_frames) {
    int shape = 2;
    for (int f = 0; f < n_frames; f++) {
        if (!((double)E4[4 * f + 3] - (double)E4[4 * f + 2] < 15.999)) return 0;
        if (!((double)E4[4 * f + 1] - (double)E4[4 * f + 0] < 64.99)) shape = 1;
    }
    return shape;
}

// ints of table per frame for (bits, window): what bgr_warp_cv_c3's caller reserves (n_frames times) for `tab_dev`
static inline int cv_tile_w(int) { return WT_W; }           // (output tile width; the 128-wide experiment of round 6 varied it)
size_t bgr_warp_cv_table_ints(int bits, Roi roi) {
    const int th = bits == 16 ? CV16_TH : CV_TH, twid = cv_tile_w(bits);
    const size_t tw = (size_t)((roi.w + twid - 1) / twid) * twid, tt = (size_t)((roi.h + th - 1) / th) * th;
    return 2 * (tw + tt);
}

hipError_t bgr_warp_cv_c3(const void* src, int w, int h, int src_stride, int bits, const double* minv_dev, const double* minv_host, int* tab_dev, int border, int max_value,
                          void* dst, int dst_stride, int n_frames, size_t src_fs, size_t dst_fs, Roi roi, hipStream_t s) {
    if (bits == 16 ? (max_value < 0 || max_value > 65535) : (bits != 8 || max_value != 255)) return hipErrorNotSupported;   // (8-bit results never exceed 255: the weights sum to 1024)
    const int th = bits == 16 ? CV16_TH : CV_TH, twid = cv_tile_w(bits);
    const int tiles_x = (roi.w + twid - 1) / twid, tiles_y = (roi.h + th - 1) / th;
    const long long tpf = (long long)tiles_x * tiles_y;
    if (tpf > 0x3fffffLL || tpf * tiles_x >= (1LL << 32)) return hipErrorNotSupported;
    const int chunk = (int)((tpf + 7) / 8);
    const uint32_t magic = (uint32_t)(0x100000000ULL / (uint32_t)tiles_x) + 1u;
    const int tab_w = tiles_x * twid, tab_h = tiles_y * th;
    const size_t per = 2 * ((size_t)tab_w + (size_t)tab_h), esz = (size_t)bits / 8;
    for (int f0 = 0; f0 < n_frames; f0 += 65535) {         // gridDim.y limit
        const int nf = n_frames - f0 < 65535 ? n_frames - f0 : 65535;
        const int* tp = tab_dev + (size_t)f0 * per;
        if (minv_host && n_frames <= kCvInlineFrames) {
            CvMatrices inl{};
            for (int k = 0; k < 6 * n_frames; k++) inl.m[k] = minv_host[k];
            hipLaunchKernelGGL(vs_k_cv_tables<true>, dim3((unsigned)((per + 255) / 256), (unsigned)nf), dim3(256), 0, s, (const double*)nullptr, inl, tab_dev, tab_w, tab_h, roi);
        } else {
            if (!minv_dev) return hipErrorInvalidValue;
            hipLaunchKernelGGL(vs_k_cv_tables<false>, dim3((unsigned)((per + 255) / 256), (unsigned)nf), dim3(256), 0, s, minv_dev + 6 * (size_t)f0, CvMatrices{},
                               tab_dev + (size_t)f0 * per, tab_w, tab_h, roi);
        }
        dim3 grid((unsigned)(chunk * 8), (unsigned)nf), block(256);
        const char* sp = (const char*)src + (size_t)f0 * src_fs * esz;
        char* dp = (char*)dst + (size_t)f0 * dst_fs * esz;
        if (bits == 16) {                                    // 10 / 12 / 16-bit containers: the word-tile kernel
            if (border == 0)
                hipLaunchKernelGGL((vs_k_bgr_warp_cv_c3_u16<0>), grid, block, 0, s, (const uint16_t*)sp, w, h, src_stride, tp, tab_w, tab_h, (uint16_t*)dp, dst_stride, src_fs, dst_fs,
                                   tiles_x, magic, (int)tpf, chunk, max_value, roi);
            else
                hipLaunchKernelGGL((vs_k_bgr_warp_cv_c3_u16<1>), grid, block, 0, s, (const uint16_t*)sp, w, h, src_stride, tp, tab_w, tab_h, (uint16_t*)dp, dst_stride, src_fs, dst_fs,
                                   tiles_x, magic, (int)tpf, chunk, max_value, roi);
        } else {
            if (border == 0)
                hipLaunchKernelGGL((vs_k_bgr_warp_cv_c3<0>), grid, block, 0, s, (const uint8_t*)sp, w, h, src_stride, tp, tab_w, tab_h, (uint8_t*)dp, dst_stride, src_fs, dst_fs,
                                   tiles_x, magic, (int)tpf, chunk, roi);
            else
                hipLaunchKernelGGL((vs_k_bgr_warp_cv_c3<1>), grid, block, 0, s, (const uint8_t*)sp, w, h, src_stride, tp, tab_w, tab_h, (uint8_t*)dp, dst_stride, src_fs, dst_fs,
                                   tiles_x, magic, (int)tpf, chunk, roi);
        }
    }
    return hipGetLastError();
}

// The per-frame extents the tuned kernel's tile prologue adds to the position of a tile's origin (see the kernel): for the kernel
// parameters P = {A, B, TX, TY} of a frame, {min, max of A1*i - B*j, min, max of B*i + A1*j} over i in [0, 63], j in [0, tile height - 1],
// widened by eps = 2^-20 * M, M = a bound on every intermediate of the fp32 position arithmetic over the output window (the
// arithmetic makes four roundings of relative size 2^-24 on values below M, for the pixel and for the tile origin: 8 * 2^-24 * M;
// eps doubles that and covers the adds below), and rounded outward to float.
void bgr_warp_c3_extents(const float* P4, int n_frames, Roi roi, int bits, int mode, float* E4) {
    const int th = tile_h_of(bits, mode);                  // the tile of the kernel that (bits, mode) selects
    for (int f = 0; f < n_frames; f++) {
        const double A1 = (double)(1.0f + P4[4 * f]), B = (double)P4[4 * f + 1], TX = (double)P4[4 * f + 2], TY = (double)P4[4 * f + 3];
        const double X = (double)(roi.x + roi.w) + 64.0, Y = (double)(roi.y + roi.h) + (double)th;
        const double M = std::max(std::fabs(A1) * X + std::fabs(B) * Y + std::fabs(TX), std::fabs(B) * X + std::fabs(A1) * Y + std::fabs(TY)) + 1.0;
        const double eps = M * (1.0 / 1048576.0);
        const double ax = A1 * (WT_W - 1), bx = -B * (th - 1), ay = B * (WT_W - 1), by = A1 * (th - 1);
        const double lo_x = std::min(0.0, ax) + std::min(0.0, bx) - eps, hi_x = std::max(0.0, ax) + std::max(0.0, bx) + eps;
        const double lo_y = std::min(0.0, ay) + std::min(0.0, by) - eps, hi_y = std::max(0.0, ay) + std::max(0.0, by) + eps;
        const double v[4] = {lo_x, hi_x, lo_y, hi_y};
        for (int k = 0; k < 4; k++) {
            float q = (float)v[k];
            if (!std::isfinite(v[k]) || !std::isfinite(q)) q = (k & 1) ? 3.0e38f : -3.0e38f;      // (the kernel's |.| < 1e6 test then refuses the window)
            else if ((k & 1) ? (double)q < v[k] : (double)q > v[k]) q = std::nextafterf(q, (k & 1) ? INFINITY : -INFINITY);
            E4[4 * f + k] = q;
        }
    }
}

}  // namespace vsk
