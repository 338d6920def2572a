// vs_kernels.hip -- gfx950 kernels for the per-pipeline (operator-level) entry points.
// One kernel per reference Halide pipeline (generators.cpp); launchers at the bottom.
// All of these are HBM/latency-bound byte and fp32 work: wave64, coalesced 4..16-byte accesses,
// LDS staging for the stencils, no MFMA (DESIGN.md "Why no MFMA").
#include "vs_kernels.hpp"
#include <cstdlib>
#include <algorithm>

#include <utility>
#include "vs_device.hpp"

using namespace vsd;

// ------------------------------------------------------------------------------------------------
// pyr_down: generators.cpp:56-92.  Separable [1 4 6 4 1]/16 on clamp-to-edge input, sampled at
// (2x,2y), truncating cast.  Every intermediate of the fp32 original is an exact multiple of
// 1/256 below 2^24, so it equals the integer form (sum_j sum_i w_j w_i in) >> 8.
//
// Two forms.  Tile form (pyr_passes, used by the fused ingest kernel): block = 256 threads, output tile 64 x 16, the
// 136 x 35 gray footprint staged in LDS, a horizontal 5-tap pass on the dot-product unit, a vertical one in packed u16.
// Row-walking form (vs_k_pyr_down_rows, the standalone kernel for levels >= 2): no LDS at all.
// ------------------------------------------------------------------------------------------------
namespace {
constexpr int PD_TW = 64, PD_TH = 16;
constexpr int PD_IW = 2 * PD_TW + 8;     // 136 staged input columns: 2*x0-4 .. 2*x0+2*TW+3 (dword aligned;
                                         // columns 2..133 of them are used)
constexpr int PD_IH = 2 * PD_TH + 3;     // 35 input rows:    2*y0-2 .. 2*y0+2*TH
constexpr int PD_IWP = PD_IW;            // LDS row pitch (136 B)
}

// The two LDS passes of pyr_down on a staged 35 x 136 gray tile (rows 2*y0-2.., columns 2*x0-4..) for a 64 x 16 output tile.
// Horizontal first, on the dot-product unit: the five taps of an output are four bytes of one (byte-aligned) dword
// against {1,4,6,4} plus one byte of the next dword, i.e. two v_dot4_u32_u8; an item = two adjacent outputs of one staged
// row (35 x 32 items), stored as one dword {h, h'} (each <= 4080).  Then vertical in packed u16 (<= 65280): a thread owns four
// adjacent outputs of one row, reads five 8-byte rows of sums, v = (a+e) + 4(b+d) + 6c on both halves, >> 8 is a byte pick,
// one dword store.  All integer and exact, so the order of the two passes does not matter.
__device__ __forceinline__ void pyr_passes(const uint8_t (*tile)[PD_IWP], uint32_t (*hsum)[PD_TW / 2], int x0, int y0,
                                           uint8_t* __restrict__ out, int ow, int oh, int out_stride) {
    for (int i = threadIdx.x; i < PD_IH * (PD_TW / 2); i += 256) {
        const int r = i >> 5, p = i & 31;                      // staged row, output pair (outputs 2p, 2p+1 of the tile)
        const uint32_t* row = (const uint32_t*)&tile[r][0];
        const uint32_t d0 = row[p], d1 = row[p + 1], d2 = row[p + 2];   // staged columns 4p .. 4p+11
        // output 2p: input columns 2*(2p)-2 .. +2 = staged columns 4p+2 .. 4p+6; output 2p+1: staged columns 4p+4 .. 4p+8
        uint32_t h0 = __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(d1, d0, 2), 0x04060401u, 0u, false);
        h0 = __builtin_amdgcn_udot4(d1, 0x00010000u, h0, false);
        uint32_t h1 = __builtin_amdgcn_udot4(d1, 0x04060401u, 0u, false);
        h1 = __builtin_amdgcn_udot4(d2, 0x00000001u, h1, false);
        hsum[r][p] = h0 | (h1 << 16);
    }
    __syncthreads();
    typedef unsigned short us2v __attribute__((ext_vector_type(2)));
    const int r = threadIdx.x >> 4, q = threadIdx.x & 15;      // output row, group of four outputs
    const int oy = y0 + r, ox = x0 + 4 * q;
    if (oy < oh && ox < ow) {
        us2v lo[5], hi[5];
#pragma unroll
        for (int j = 0; j < 5; j++) {
            const uint2 v = *(const uint2*)&hsum[2 * r + j][2 * q];
            lo[j] = __builtin_bit_cast(us2v, v.x);
            hi[j] = __builtin_bit_cast(us2v, v.y);
        }
        const us2v va = (lo[0] + lo[4]) + (lo[1] + lo[3]) * (unsigned short)4 + lo[2] * (unsigned short)6;
        const us2v vb = (hi[0] + hi[4]) + (hi[1] + hi[3]) * (unsigned short)4 + hi[2] * (unsigned short)6;
        // bytes 1 and 3 of each packed pair are the four outputs >> 8
        const uint32_t packed = __builtin_amdgcn_perm(__builtin_bit_cast(uint32_t, vb), __builtin_bit_cast(uint32_t, va), 0x07050301u);
        uint8_t* dst = out + (size_t)oy * out_stride + ox;
        if (ox + 3 < ow && (((uintptr_t)dst) & 3) == 0) {
            *(uint32_t*)dst = packed;
        } else {
            for (int k = 0; k < 4 && ox + k < ow; k++) dst[k] = (uint8_t)(packed >> (8 * k));
        }
    }
}

// ------------------------------------------------------------------------------------------------
// pyr_down, row-walking form (the layout of vs_k_keyframe_rows): a wave owns a strip of 248 input columns and a band of
// PR_OUT output rows; lanes 1 .. n_live own 4 input columns = 2 output columns each and walk down the 2*PR_OUT + 3 input
// rows of the band, lane 0 and lane n_live + 1 fetch the words that hold the strip's neighbour columns.  Per input row
// every lane issues ONE dword load at (scalar row offset + its constant byte offset; words that would cross the right
// image border start at w - 4 instead), the two pixels to the left and the one to the right come from the neighbouring
// lanes (wave_shr / wave_shl DPP moves), and five v_perm_b32 pair the bytes up as {even, odd}-output operands -- their
// selectors are per-lane constants fixed before the loop, which is where clamp-to-edge and the shifted border words are
// resolved.  The horizontal 1-4-6-4-1 runs for both outputs at once in packed u16 (<= 4080), the vertical one on a
// five-row register window, again packed (<= 65280), every second input row; >> 8 is a byte pick.  No LDS, every input
// row is read once (+ 3 halo rows per band), D rows in flight.  Integer arithmetic: the same bytes as pyr_passes above
// (which the fused ingest kernel still runs on its LDS tile).
// ------------------------------------------------------------------------------------------------
namespace {
#ifndef VS_PR_OUT
#define VS_PR_OUT 16
#endif
constexpr int PR_OUT = VS_PR_OUT;           // output rows per wave (batches); a single frame is cut into bands of 4 rows: 4x the waves, a quarter of the walk
constexpr int PR_LIVE = 62;                 // lanes of a wave that produce outputs (two each)
typedef unsigned short us2 __attribute__((ext_vector_type(2)));
}

template <int PR_OUT, bool NARROW>
__global__ __launch_bounds__(256) void vs_k_pyr_down_rows(const uint8_t* __restrict__ in, int w, int h, int in_stride,
                                                          uint8_t* __restrict__ out, int ow, int oh, int out_stride,
                                                          size_t in_frame_stride, size_t out_frame_stride, int strips_x, int bands) {
    in += blockIdx.y * in_frame_stride;
    out += blockIdx.y * out_frame_stride;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
    const int job = blockIdx.x * 4 + wave;
    if (job >= strips_x * bands) return;                     // whole wave
    const int band = __builtin_amdgcn_readfirstlane(job / strips_x), strip = job - band * strips_x;
    const int x_first = strip * (PR_LIVE * 4);               // first input column of the strip = 2 * its first output column
    const int n_live = min(PR_LIVE, (ow - strip * (PR_LIVE * 2) + 1) >> 1);      // lanes 1 .. n_live: two outputs each
    const int x0 = x_first + (lane - 1) * 4, ox = x0 >> 1, oy0 = band * PR_OUT;
    const bool live = lane >= 1 && lane <= n_live;
    // Words are dwords inside the row; an image narrower than a dword is read bytewise into clamp-extended words instead
    // (wl: the width the word layout sees).
    constexpr bool narrow = NARROW;                          // w < 4
    const int wl = narrow ? (1 << 30) : w;
    auto word_off = [&](int l) { return l <= 0 ? max(x_first - 4, 0) : min(x_first + (min(l, n_live + 1) - 1) * 4, wl - 4); };
    const int off_o = word_off(lane), off_l = word_off(lane - 1), off_r = word_off(lane + 1);
    // v_perm byte selector of clamped column c: out of the second operand (bytes 0..3, word at lo) if it lies there, else out
    // of the first one (bytes 4..7, word at hi)
    auto sel = [&](int c, int hi, int lo) -> unsigned {
        c = min(max(c, 0), wl - 1);
        const unsigned d = (unsigned)(c - lo);
        return d < 4u ? d : 4u + ((unsigned)(c - hi) & 3u);
    };
    // operand pairs {for output ox, for output ox + 1} as two u16: taps -2 .. +2 around x0 and around x0 + 2
    const unsigned sel_a = 0x0c000c00u | sel(x0 - 2, off_o, off_l) | (sel(x0, off_o, off_l) << 16);           // perm(own, left)
    const unsigned sel_b = 0x0c000c00u | sel(x0 - 1, off_o, off_l) | (sel(x0 + 1, off_o, off_l) << 16);       // perm(own, left)
    const unsigned sel_c = 0x0c000c00u | sel(x0, off_o, off_o) | (sel(x0 + 2, off_o, off_o) << 16);           // perm(own, own)
    const unsigned sel_d = 0x0c000c00u | sel(x0 + 1, off_o, off_o) | (sel(x0 + 3, off_o, off_o) << 16);       // perm(own, own)
    const unsigned sel_e = 0x0c000c00u | sel(x0 + 2, off_r, off_o) | (sel(x0 + 4, off_r, off_o) << 16);       // perm(right, own)
    auto row_at = [&](int y) -> uint32_t {                   // the lane's word of input row y (clamped), y wave-uniform
        int yc;
        asm("s_max_i32 %0, %1, 0\n\ts_min_i32 %0, %0, %2" : "=&s"(yc) : "s"(__builtin_amdgcn_readfirstlane(y)), "s"(__builtin_amdgcn_readfirstlane(h - 1)) : "scc");
        const uint8_t* rp = in + ((uint32_t)(yc * in_stride) + (uint32_t)off_o);      // (a level is far below 4 GB)
        uint32_t v;
        if (!narrow) { __builtin_memcpy(&v, rp, 4); return v; }
        v = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) v |= (uint32_t)in[(uint32_t)(yc * in_stride) + (uint32_t)min(off_o + k, w - 1)] << (8 * k);
        return v;
    };
    // horizontal pass of one input row for the lane's two outputs: {h(ox), h(ox + 1)}
    auto hsum_at = [&](uint32_t own) -> us2 {
        const uint32_t left = (uint32_t)dpp_mov0<0x138>((int)own);      // lane - 1
        const uint32_t right = (uint32_t)dpp_mov0<0x130>((int)own);     // lane + 1
        const us2 a = __builtin_bit_cast(us2, __builtin_amdgcn_perm(own, left, sel_a));     // {in(x0-2), in(x0)}
        const us2 b = __builtin_bit_cast(us2, __builtin_amdgcn_perm(own, left, sel_b));     // {in(x0-1), in(x0+1)}
        const us2 c = __builtin_bit_cast(us2, __builtin_amdgcn_perm(own, own, sel_c));      // {in(x0),   in(x0+2)}
        const us2 d = __builtin_bit_cast(us2, __builtin_amdgcn_perm(own, own, sel_d));      // {in(x0+1), in(x0+3)}
        const us2 e = __builtin_bit_cast(us2, __builtin_amdgcn_perm(right, own, sel_e));    // {in(x0+2), in(x0+4)}
        return (a + e) + (b + d) * (unsigned short)4 + c * (unsigned short)6;
    };
    // window of horizontal sums of input rows 2*oy-2 .. 2*oy+2; two new rows per output row
    const int iy0 = 2 * oy0 - 2;
    us2 h0 = hsum_at(row_at(iy0)), h1 = hsum_at(row_at(iy0 + 1)), h2 = hsum_at(row_at(iy0 + 2));
    constexpr int D = 4;
    uint32_t ring[D];
#pragma unroll
    for (int j = 0; j < D; j++) ring[j] = row_at(iy0 + 3 + j);
    const int rows_out = min(PR_OUT, oh - oy0);
    const bool two = ox + 1 < ow;
#pragma unroll 1
    for (int r0 = 0; r0 < rows_out; r0 += D / 2) {
#pragma unroll
        for (int j = 0; j < D / 2; j++) {
            const int r = r0 + j;
            const us2 h3 = hsum_at(ring[2 * j]), h4 = hsum_at(ring[2 * j + 1]);
            ring[2 * j] = row_at(iy0 + 3 + 2 * r + D);
            ring[2 * j + 1] = row_at(iy0 + 4 + 2 * r + D);
            if (r < rows_out) {                              // uniform
                const us2 v = (h0 + h4) + (h1 + h3) * (unsigned short)4 + h2 * (unsigned short)6;
                const uint32_t packed = __builtin_amdgcn_perm(0u, __builtin_bit_cast(uint32_t, v), 0x0c0c0301u);   // {v.x >> 8, v.y >> 8}
                uint8_t* dst = out + ((uint32_t)((oy0 + r) * out_stride) + (uint32_t)ox);
                if (live) {
                    if (two) { const uint16_t t = (uint16_t)packed; __builtin_memcpy(dst, &t, 2); }
                    else dst[0] = (uint8_t)packed;
                }
            }
            h0 = h2; h1 = h3; h2 = h4;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// bgr_to_gray: stands in for cv::cvtColor(BGR2GRAY) at alignment.cpp:212 (OpenCV 4.x 15-bit fixed
// point).  4 pixels per thread: 12 B in (three dwords when aligned), 4 B out.
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void vs_k_bgr_to_gray(const T* __restrict__ src, int w, int h, int src_stride,
                                                        int shift_to_8, uint8_t* __restrict__ dst, int dst_stride,
                                                        size_t src_frame_stride, size_t dst_frame_stride) {
    src += blockIdx.z * src_frame_stride;
    dst += blockIdx.z * dst_frame_stride;
    const int x = (blockIdx.x * 256 + threadIdx.x) * 4, y = blockIdx.y;
    if (x >= w) return;
    const T* p = src + (size_t)y * src_stride + (size_t)x * 3;
    uint8_t* o = dst + (size_t)y * dst_stride + x;
    uint32_t g[4];
    const int n = min(4, w - x);
    if (sizeof(T) == 1 && n == 4 && (((uintptr_t)p) & 3) == 0) {
        const uint32_t* p4 = (const uint32_t*)p;
        uint32_t a = p4[0], b = p4[1], c = p4[2];
        uint32_t px[12] = {a & 255, (a >> 8) & 255, (a >> 16) & 255, a >> 24, b & 255, (b >> 8) & 255,
                           (b >> 16) & 255, b >> 24, c & 255, (c >> 8) & 255, (c >> 16) & 255, c >> 24};
#pragma unroll
        for (int k = 0; k < 4; k++)
            g[k] = (px[3 * k] * 3735u + px[3 * k + 1] * 19235u + px[3 * k + 2] * 9798u + 16384u) >> 15;
    } else {
        for (int k = 0; k < n; k++)
            g[k] = ((uint32_t)p[3 * k] * 3735u + (uint32_t)p[3 * k + 1] * 19235u + (uint32_t)p[3 * k + 2] * 9798u + 16384u) >> 15;
    }
    for (int k = 0; k < n; k++) { g[k] >>= shift_to_8; g[k] = g[k] > 255u ? 255u : g[k]; }
    if (n == 4 && (((uintptr_t)o) & 3) == 0) {
        *(uint32_t*)o = g[0] | (g[1] << 8) | (g[2] << 16) | (g[3] << 24);
    } else {
        for (int k = 0; k < n; k++) o[k] = (uint8_t)g[k];
    }
}

// ------------------------------------------------------------------------------------------------
// Fused ingest: BGR -> gray level 0 AND level 1 of the pyramid in one pass (alignment.cpp:212 + the first
// PyrDown of :220-223).  A workgroup converts the 136 x 35 gray footprint of one 64 x 16 level-1 tile into
// LDS straight from the BGR frame, writes its own 128 x 32 block of level 0, and runs pyr_down's two LDS
// passes on the staged tile.  Level 0 is never re-read from HBM (saves W*H bytes per frame); the halo
// (19 % more BGR reads, mostly L2 hits) is recomputed instead.  Clamp-to-edge of the gray image = gray of
// the clamped BGR pixel, so the result is the same bytes as bgr_to_gray followed by pyr_down.
// ------------------------------------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ uint32_t gray_of(const T* p, int shift_to_8) {
    uint32_t g = ((uint32_t)p[0] * 3735u + (uint32_t)p[1] * 19235u + (uint32_t)p[2] * 9798u + 16384u) >> 15;
    g >>= shift_to_8;
    return g > 255u ? 255u : g;
}

template <typename T>
__global__ __launch_bounds__(256) void vs_k_ingest_pyr(const T* __restrict__ src_all, int w, int h, int src_stride,
                                                       int shift_to_8, uint8_t* __restrict__ g0_all, uint8_t* __restrict__ g1_all,
                                                       int ow, int oh, size_t src_frame_stride, size_t pyr_frame_stride,
                                                       int tiles_x, uint32_t tiles_x_magic, int tiles_per_frame, int chunk) {
    __shared__ __attribute__((aligned(8))) uint8_t tile[PD_IH][PD_IWP];
    __shared__ __attribute__((aligned(8))) uint32_t hsum[PD_IH][PD_TW / 2];
    // One workgroup per tile, frame = blockIdx.y.  XCD-aware order inside a frame (see vs_warp.hip): gridDim.x is a multiple of 8,
    // so each XCD walks one contiguous raster run of the frame's tiles.  The tile split is a multiply-high on the scalar unit
    // (tl / tiles_x == (tl * magic) >> 32 while tl * tiles_x < 2^32; a vector-unit division here was a sixth of the kernel's
    // vector instructions).
    {
    const int tl = (int)(blockIdx.x & 7) * chunk + (int)(blockIdx.x >> 3);
    if (tl < tiles_per_frame) {                                   // (uniform)
    const int frame = blockIdx.y;
    const int tyi = tiles_x == 1 ? tl : (int)__umulhi((uint32_t)tl, tiles_x_magic);
    const int txi = tl - tyi * tiles_x;
    const T* __restrict__ src = src_all + (size_t)frame * src_frame_stride;
    uint8_t* __restrict__ g0 = g0_all + (size_t)frame * pyr_frame_stride;
    uint8_t* __restrict__ g1 = g1_all + (size_t)frame * pyr_frame_stride;
    const int x0 = txi * PD_TW, y0 = tyi * PD_TH;          // level-1 tile origin
    const int ix0 = 2 * x0 - 4, iy0 = 2 * y0 - 2;          // level-0 origin of the staged tile
    // gray = (B*3735 + G*19235 + R*9798 + 16384) >> 15 on the dot-product units.  u8: the pixel's bytes {B,G,R,x} against the
    // weights split into high and low bytes (x meets weight 0): two v_dot4_u32_u8 + one shift-add; the four pixels of a
    // 12-byte group are three v_alignbyte away.  u16: {B,G} . {3735,19235} is one v_dot2_u32_u16, R one multiply-add.
    constexpr uint32_t kWLo = (3735u & 255u) | ((19235u & 255u) << 8) | ((9798u & 255u) << 16);
    constexpr uint32_t kWHi = (3735u >> 8) | ((19235u >> 8) << 8) | ((9798u >> 8) << 16);
    // (row offsets below are 24-bit x 24-bit multiplies: full rate, where a 32-bit multiply is a quarter-rate instruction)
    const bool frame_aligned = ((((uintptr_t)src) | ((uintptr_t)src_stride * sizeof(T))) & 3u) == 0 && src_stride < (1 << 24) &&
                               h < (1 << 24) && w < (1 << 24);   // uniform
    const bool g0_aligned = ((((uintptr_t)g0) | (uintptr_t)w) & 3u) == 0;
    // gray of the four pixels of one 12- / 24-byte group (dword aligned)
    auto gray4 = [&](const uint32_t* q, uint32_t (&g)[4]) {
        if (sizeof(T) == 1) {                                 // q: B0 G0 R0 B1 | G1 R1 B2 G2 | R2 B3 G3 R3
            const uint32_t px[4] = {q[0], __builtin_amdgcn_alignbyte(q[1], q[0], 3), __builtin_amdgcn_alignbyte(q[2], q[1], 2), q[2] >> 8};
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const uint32_t hi = __builtin_amdgcn_udot4(px[k], kWHi, 0u, false);
                const uint32_t lo = __builtin_amdgcn_udot4(px[k], kWLo, 16384u, false);
                g[k] = ((hi << 8) + lo) >> 15;                // <= 255 for 8-bit input: no clamp
                g[k] >>= shift_to_8;
            }
        } else {                                              // q: B0G0 R0B1 G1R1 B2G2 | R2B3 G3R3
            const uint32_t bg[4] = {q[0], __builtin_amdgcn_alignbyte(q[2], q[1], 2), q[3], __builtin_amdgcn_alignbyte(q[5], q[4], 2)};
            const uint32_t rr[4] = {q[1] & 0xffffu, q[2] >> 16, q[4] & 0xffffu, q[5] >> 16};
            constexpr uint32_t kWBG = 3735u | (19235u << 16);
#pragma unroll
            for (int k = 0; k < 4; k++) {
                typedef unsigned short us2 __attribute__((ext_vector_type(2)));
                const uint32_t t = __builtin_amdgcn_udot2(__builtin_bit_cast(us2, bg[k]), __builtin_bit_cast(us2, kWBG), 16384u, false);
                g[k] = (t + rr[k] * 9798u) >> 15;
                g[k] >>= shift_to_8;
                g[k] = g[k] > 255u ? 255u : g[k];
            }
        }
    };
    // level 0: the part of the staged tile that is this workgroup's own 128 x 32 block (rows 2..33, columns 4..131)
    auto stage = [&](int r, int c4, int gx, uint32_t v) {
        *(uint32_t*)&tile[r][c4] = v;
        if (r >= 2 && r < 34 && c4 >= 4 && c4 < 132) {
            const int oy = iy0 + r, ox = gx;                  // = 2*y0 + (r-2), 2*x0 + (c4-4): never negative here
            if (oy < h && ox < w) {
                uint8_t* dst = g0 + (size_t)oy * w + ox;
                if (g0_aligned && ox + 3 < w) {
                    *(uint32_t*)dst = v;
                } else {
                    for (int k = 0; k < 4 && ox + k < w; k++) dst[k] = (uint8_t)(v >> (8 * k));
                }
            }
        }
    };
    constexpr int N_ITEMS = PD_IH * (PD_IW / 4), QW = sizeof(T) == 1 ? 3 : 6;
    static_assert(PD_TW == 64 && PD_TH == 16, "the item layout below is written for a 35 x 34-group footprint around a 32 x 32-group block");
    if (frame_aligned && w >= 4) {
        // Every group of the footprint is ONE load of 12 / 24 bytes at a column clamped into the row (a group that hangs over
        // the left / right image border reads the nearest full group instead and picks its clamped pixels afterwards), and
        // all of a thread's loads are issued before the first conversion: five requests in flight per thread.
        // Items: the workgroup's own 32 x 32 groups (staged rows 2..33, groups 1..32) are four per thread -- row
        // 2 + (tid >> 5) + 8 * it, group 1 + (tid & 31): no index arithmetic, no range tests, level 0 is written from them;
        // the 166 halo groups (rows 0, 1, 34 and the two side columns) are a fifth item of the first 166 threads.
        const int tid = (int)threadIdx.x;
        // packs the grays of four pixels; u8: g = byte 2 of 2 * (B*3735 + G*19235 + R*9798 + 16384) with the weights split as
        // (w >> 7) * 256 + 2 * (w & 127) -- two v_dot4_u32_u8 and one shift-add per pixel, three v_perm_b32 per group
        auto gray_word = [&](const uint32_t* q) -> uint32_t {
            if (sizeof(T) == 1) {
                constexpr uint32_t kH = (3735u >> 7) | ((19235u >> 7) << 8) | ((9798u >> 7) << 16);
                constexpr uint32_t kL = (2u * (3735u & 127u)) | ((2u * (19235u & 127u)) << 8) | ((2u * (9798u & 127u)) << 16);
                const uint32_t px[4] = {q[0], __builtin_amdgcn_alignbyte(q[1], q[0], 3), __builtin_amdgcn_alignbyte(q[2], q[1], 2), q[2] >> 8};
                uint32_t t[4];
#pragma unroll
                for (int k = 0; k < 4; k++)
                    t[k] = (__builtin_amdgcn_udot4(px[k], kH, 0u, false) << 8) + __builtin_amdgcn_udot4(px[k], kL, 32768u, false);
                const uint32_t t01 = __builtin_amdgcn_perm(t[1], t[0], 0x0c0c0602u), t23 = __builtin_amdgcn_perm(t[3], t[2], 0x0c0c0602u);
                return __builtin_amdgcn_perm(t23, t01, 0x05040100u);
            }
            uint32_t g[4];
            gray4(q, g);
            return g[0] | (g[1] << 8) | (g[2] << 16) | (g[3] << 24);
        };
        auto load_group_at = [&](uint32_t (&q)[QW], uint32_t byte_off) {
            const uint32_t* gp = (const uint32_t*)((const uint8_t*)src + byte_off);   // (a frame is below 4 GB)
            if (sizeof(T) == 1) {
                typedef uint32_t u32x3 __attribute__((ext_vector_type(3)));
                const u32x3 t = *(const u32x3*)gp;
                q[0] = t.x; q[1] = t.y; q[2] = t.z;
            } else {
                typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
                typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
                const u32x4 ta = *(const u32x4*)gp;
                const u32x2 tb = *(const u32x2*)(gp + 4);
                q[0] = ta.x; q[1] = ta.y; q[2] = ta.z; q[3] = ta.w; q[QW - 2] = tb.x; q[QW - 1] = tb.y;
            }
        };
        auto load_group = [&](uint32_t (&q)[QW], int gy, int gxc) {
            load_group_at(q, (__umul24((uint32_t)gy, (uint32_t)src_stride) + (uint32_t)(gxc * 3)) * (uint32_t)sizeof(T));
        };
        // border group: pixel k = clamp(gx + k) of the row = byte clamp(gx + k) - gxc of the packed word
        auto border_sel = [&](int gx, int gxc) {
            uint32_t sel = 0;
#pragma unroll
            for (int k = 0; k < 4; k++) sel |= (uint32_t)(clampi(gx + k, 0, w - 1) - gxc) << (8 * k);
            return sel;
        };
        const int r_own = 2 + (tid >> 5), c_own = 1 + (tid & 31);
        const int gx_own = ix0 + 4 * c_own, gxc_own = min(gx_own, w - 4);                 // gx_own = 2 * x0 + 4 * (tid & 31) >= 0
        int r_h, c_h;                                                                     // the halo item
        if (tid < 3 * 34) { const int k = (tid >= 34) + (tid >= 68); r_h = k == 2 ? PD_IH - 1 : k; c_h = tid - 34 * k; }
        else { const int t = min(tid - 3 * 34, 63); r_h = 2 + (t >> 1); c_h = (t & 1) * 33; }
        const bool has_halo = tid < N_ITEMS - 1024;
        const int gx_h = ix0 + 4 * c_h, gxc_h = clampi(gx_h, 0, w - 4);
        uint32_t q[5][QW];
        uint32_t* const tile_w = (uint32_t*)&tile[0][0];
        // Interior tiles (the whole 136 x 35 footprint inside the image, dword-aligned level 0: all but the frame's rim -- 82 % of the
        // tiles at 1080p, 90 % at 4K): no clamps, no border selectors, no partial stores; the four own rows are uniform strides apart.
        const bool interior = g0_aligned && ix0 >= 0 && ix0 + PD_IW <= w && iy0 >= 0 && iy0 + PD_IH <= h;   // uniform
        if (interior) {
            const uint32_t o_own = (__umul24((uint32_t)(iy0 + r_own), (uint32_t)src_stride) + (uint32_t)(gx_own * 3)) * (uint32_t)sizeof(T);
            const uint32_t o_step = 8u * (uint32_t)src_stride * (uint32_t)sizeof(T);
#pragma unroll
            for (int it = 0; it < 4; it++) load_group_at(q[it], o_own + (uint32_t)it * o_step);
            load_group_at(q[4], (__umul24((uint32_t)(iy0 + r_h), (uint32_t)src_stride) + (uint32_t)(gx_h * 3)) * (uint32_t)sizeof(T));
            const uint32_t g_own = __umul24((uint32_t)(iy0 + r_own), (uint32_t)w) + (uint32_t)gx_own, g_step = 8u * (uint32_t)w;
            const int t_own = r_own * (PD_IWP / 4) + c_own;
#pragma unroll
            for (int it = 0; it < 4; it++) {
                const uint32_t v = gray_word(q[it]);
                tile_w[t_own + 8 * it * (PD_IWP / 4)] = v;
                *(uint32_t*)(g0 + (g_own + (uint32_t)it * g_step)) = v;
            }
            if (has_halo) tile_w[r_h * (PD_IWP / 4) + c_h] = gray_word(q[4]);
        } else {
#pragma unroll
        for (int it = 0; it < 4; it++) load_group(q[it], min(iy0 + r_own + 8 * it, h - 1), gxc_own);
        load_group(q[4], clampi(iy0 + r_h, 0, h - 1), gxc_h);
        const bool own_border = gx_own != gxc_own, own_dword = g0_aligned && gx_own + 3 < w;
        const uint32_t own_sel = border_sel(gx_own, gxc_own);
#pragma unroll
        for (int it = 0; it < 4; it++) {
            uint32_t v = gray_word(q[it]);
            if (own_border) v = __builtin_amdgcn_perm(0u, v, own_sel);
            const int r = r_own + 8 * it, oy = iy0 + r;                                   // level-0 row 2 * y0 + (r - 2)
            tile_w[r * (PD_IWP / 4) + c_own] = v;
            if (oy < h && gx_own < w) {
                uint8_t* dst = g0 + (__umul24((uint32_t)oy, (uint32_t)w) + (uint32_t)gx_own);
                if (own_dword) *(uint32_t*)dst = v;
                else for (int k = 0; k < 4 && gx_own + k < w; k++) dst[k] = (uint8_t)(v >> (8 * k));
            }
        }
        if (has_halo) {
            uint32_t v = gray_word(q[4]);
            if (gx_h != gxc_h) v = __builtin_amdgcn_perm(0u, v, border_sel(gx_h, gxc_h));
            tile_w[r_h * (PD_IWP / 4) + c_h] = v;
        }
        }
    } else {
        for (int i = threadIdx.x; i < N_ITEMS; i += 256) {
            const int r = i / (PD_IW / 4), c4 = (i - r * (PD_IW / 4)) * 4;
            const int gy = clampi(iy0 + r, 0, h - 1), gx = ix0 + c4;
            const T* row = src + (size_t)gy * src_stride;
            uint32_t v = 0;
#pragma unroll
            for (int k = 0; k < 4; k++) v |= gray_of(row + (size_t)clampi(gx + k, 0, w - 1) * 3, shift_to_8) << (8 * k);
            stage(r, c4, gx, v);
        }
    }
    __syncthreads();
    pyr_passes(tile, hsum, x0, y0, g1, ow, oh, ow);
    }
    }
}

// ------------------------------------------------------------------------------------------------
// grad_xy: generators.cpp:202-224.  API-parity kernel only (the engine never writes gradients).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void vs_k_grad_xy(const uint8_t* __restrict__ in, int w, int h, int stride,
                                                    float* __restrict__ gx, float* __restrict__ gy) {
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (x >= w) return;
    const uint8_t* row = in + (size_t)y * stride;
    float l = (float)row[max(x - 1, 0)], r = (float)row[min(x + 1, w - 1)];
    float u = (float)in[(size_t)max(y - 1, 0) * stride + x], d = (float)in[(size_t)min(y + 1, h - 1) * stride + x];
    gx[(size_t)y * w + x] = 0.5f * (r - l);
    gy[(size_t)y * w + x] = 0.5f * (d - u);
}

// ------------------------------------------------------------------------------------------------
// grad_argmax from float planes: generators.cpp:260-294.  One wave per tile.  The reference's
// unscheduled Halide::argmax is a serial scan (r.x inner, r.y outer) that updates on strict '>',
// i.e. the winner is the largest |g| and, among equals, the smallest scan index.  As a wave
// max-reduction: key = (bits(|g|) << 32) | (0xffffffff - scan_index); |g| >= 0 so its IEEE bits
// order like unsigned integers.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void vs_k_grad_argmax(const float* __restrict__ gx, const float* __restrict__ gy,
                                                        int w, int h, int ts, int tx, int ty,
                                                        uint16_t* __restrict__ lmx, uint16_t* __restrict__ lmy) {
    const int tile = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (tile >= tx * ty) return;
    const int tyi = tile / tx, txi = tile % tx;
    const int bx = txi * ts, by = tyi * ts, n = ts * ts;
    unsigned long long kx = 0, ky = 0;
    for (int i = lane; i < n; i += 64) {
        int ry = i / ts, rx = i % ts;
        size_t o = (size_t)(by + ry) * w + (bx + rx);
        unsigned long long inv = 0xffffffffu - (unsigned)i;
        unsigned long long a = ((unsigned long long)__float_as_uint(fabsf(gx[o])) << 32) | inv;
        unsigned long long b = ((unsigned long long)__float_as_uint(fabsf(gy[o])) << 32) | inv;
        kx = a > kx ? a : kx;
        ky = b > ky ? b : ky;
    }
    kx = wave_max_u64(kx);
    ky = wave_max_u64(ky);
    if (lane == 0) {
        const size_t nt = (size_t)tx * ty;
        int ix = (int)(0xffffffffu - (unsigned)(kx & 0xffffffffu)), iy = (int)(0xffffffffu - (unsigned)(ky & 0xffffffffu));
        lmx[tile] = (uint16_t)(bx + ix % ts);
        lmx[nt + tile] = (uint16_t)(by + ix / ts);
        lmy[tile] = (uint16_t)(bx + iy % ts);
        lmy[nt + tile] = (uint16_t)(by + iy / ts);
    }
}

// ------------------------------------------------------------------------------------------------
// sparse_jac from float planes: generators.cpp:332-386.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void vs_k_sparse_jac(const float* __restrict__ gx, const float* __restrict__ gy,
                                                       int w, int h, const uint16_t* __restrict__ lmx,
                                                       const uint16_t* __restrict__ lmy, int nt,
                                                       float* __restrict__ jx, float* __restrict__ jy) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= nt) return;
    const float cx = (float)w * 0.5f, cy = (float)h * 0.5f, scale = 1.f / (float)w;
    int ix0 = min((int)lmx[i], w - 1), iy0 = min((int)lmx[nt + i], h - 1);
    int ix1 = min((int)lmy[i], w - 1), iy1 = min((int)lmy[nt + i], h - 1);
    float u0 = (float)ix0 - cx, v0 = (float)iy0 - cy, u1 = (float)ix1 - cx, v1 = (float)iy1 - cy;
    float g0 = gx[(size_t)iy0 * w + ix0], g1 = gy[(size_t)iy1 * w + ix1];
    jx[i] = 2.f * g0 * u0 * scale;
    jx[nt + i] = 2.f * g0 * (-v0) * scale;
    jx[2 * (size_t)nt + i] = 2.f * g0;
    jx[3 * (size_t)nt + i] = 0.f;
    jy[i] = 2.f * g1 * v1 * scale;
    jy[nt + i] = 2.f * g1 * u1 * scale;
    jy[2 * (size_t)nt + i] = 0.f;
    jy[3 * (size_t)nt + i] = 2.f * g1;
}

// ------------------------------------------------------------------------------------------------
// Fused keyframe pass: grad_xy + grad_argmax + sparse_jac straight from the u8 level
// (alignment.cpp:237-276 runs them back to back; the 8 B/px gradient planes are never needed).
// |0.5f*(a-b)| orders exactly like the integer |a-b| in [0,255], so the arg-max runs on u32
// keys (|a-b| << 16) | (0xffff - scan_index) -- ts <= 64 keeps scan_index < 4096.
// One wave per tile, 4 tiles per block.  Inside a tile one work item = 8 adjacent pixels of one row: it loads
// its row (10 bytes, x-1..x+8) and the rows above / below (8 bytes each) with unaligned 8-byte loads and
// forms 8 |gx| and 8 |gy| from registers (~1/3 load per pixel instead of 4 byte gathers); items whose loads
// would leave the image take a clamped byte path.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint64_t load_u64_unaligned(const uint8_t* p) {
    uint64_t r;
    __builtin_memcpy(&r, p, 8);
    return r;
}

__global__ __launch_bounds__(256) void vs_k_keyframe(const uint8_t* __restrict__ img, int w, int h, int stride,
                                                     int ts, int tx, int ty, uint16_t* __restrict__ lmx,
                                                     uint16_t* __restrict__ lmy, float* __restrict__ jx,
                                                     float* __restrict__ jy, size_t img_frame_stride,
                                                     size_t lm_frame_stride, size_t jac_frame_stride, bool aos) {
    img += blockIdx.y * img_frame_stride;
    lmx += blockIdx.y * lm_frame_stride; lmy += blockIdx.y * lm_frame_stride;
    jx += blockIdx.y * jac_frame_stride; jy += blockIdx.y * jac_frame_stride;
    // 16 lanes per tile, 4 tiles per wave, 16 tiles per block: the per-tile tail (Jacobian + 12 stores) of four
    // tiles runs side by side and a wave has 4x the loads in flight
    const int tile = blockIdx.x * 16 + (threadIdx.x >> 4), lane = threadIdx.x & 15;
    const bool live = tile < tx * ty;
    const int tyi = live ? tile / tx : 0, txi = live ? tile % tx : 0;
    const int bx = txi * ts, by = tyi * ts;
    const int chunks = (ts + 7) >> 3, items = live ? ts * chunks : 0;
    unsigned kx = 0, ky = 0;
    for (int it = lane; it < items; it += 16) {
        const int ry = it / chunks, cx = (it - ry * chunks) * 8;
        const int x = bx + cx, y = by + ry, n = min(8, ts - cx);
        const uint8_t* row = img + (size_t)y * stride;
        uint32_t c[10], up[8], dn[8];   // c[k] = in(x-1+k, y)
        if (x >= 1 && x + 8 < w) {
            const uint64_t a = load_u64_unaligned(row + x - 1);
            const uint32_t b = (uint32_t)row[x + 7] | ((uint32_t)row[x + 8] << 8);
            const uint64_t u = load_u64_unaligned(img + (size_t)max(y - 1, 0) * stride + x);
            const uint64_t d = load_u64_unaligned(img + (size_t)min(y + 1, h - 1) * stride + x);
#pragma unroll
            for (int k = 0; k < 8; k++) {
                c[k] = (uint32_t)(a >> (8 * k)) & 0xffu;
                up[k] = (uint32_t)(u >> (8 * k)) & 0xffu;
                dn[k] = (uint32_t)(d >> (8 * k)) & 0xffu;
            }
            c[8] = b & 0xffu; c[9] = b >> 8;
        } else {
            const uint8_t* ur = img + (size_t)max(y - 1, 0) * stride;
            const uint8_t* dr = img + (size_t)min(y + 1, h - 1) * stride;
#pragma unroll
            for (int k = 0; k < 10; k++) c[k] = row[clampi(x - 1 + k, 0, w - 1)];
#pragma unroll
            for (int k = 0; k < 8; k++) { up[k] = ur[min(x + k, w - 1)]; dn[k] = dr[min(x + k, w - 1)]; }
        }
        const unsigned inv0 = 0xffffu - (unsigned)(ry * ts + cx);
#pragma unroll
        for (int k = 0; k < 8; k++) {
            if (k < n) {
                const unsigned inv = inv0 - (unsigned)k;
                const unsigned a = ((unsigned)abs((int)c[k + 2] - (int)c[k]) << 16) | inv;
                const unsigned b = ((unsigned)abs((int)dn[k] - (int)up[k]) << 16) | inv;
                kx = a > kx ? a : kx;
                ky = b > ky ? b : ky;
            }
        }
    }
#pragma unroll
    for (int off = 8; off > 0; off >>= 1) {       // max over the tile's 16 lanes
        const unsigned ox = __shfl_down(kx, off, 16), oy = __shfl_down(ky, off, 16);
        kx = ox > kx ? ox : kx;
        ky = oy > ky ? oy : ky;
    }
    if (lane == 0 && live) {
        const size_t nt = (size_t)tx * ty;
        int sx = (int)(0xffffu - (kx & 0xffffu)), sy = (int)(0xffffu - (ky & 0xffffu));
        int ix0 = bx + sx % ts, iy0 = by + sx / ts, ix1 = bx + sy % ts, iy1 = by + sy / ts;
        if (aos) {     // engine tables: one {x, y} pair / one float4 per tile and set (a gather then touches one line, not six)
            ((uint32_t*)lmx)[tile] = (uint32_t)ix0 | ((uint32_t)iy0 << 16);
            ((uint32_t*)lmy)[tile] = (uint32_t)ix1 | ((uint32_t)iy1 << 16);
        } else {
            lmx[tile] = (uint16_t)ix0; lmx[nt + tile] = (uint16_t)iy0;
            lmy[tile] = (uint16_t)ix1; lmy[nt + tile] = (uint16_t)iy1;
        }
        // generators.cpp:346-385 (the min(.., w-1) clamps are no-ops: keypoints lie inside the image)
        const float cx = (float)w * 0.5f, cy = (float)h * 0.5f, scale = 1.f / (float)w;
        float g0 = 0.5f * ((float)img[(size_t)iy0 * stride + min(ix0 + 1, w - 1)] - (float)img[(size_t)iy0 * stride + max(ix0 - 1, 0)]);
        float g1 = 0.5f * ((float)img[(size_t)min(iy1 + 1, h - 1) * stride + ix1] - (float)img[(size_t)max(iy1 - 1, 0) * stride + ix1]);
        float u0 = (float)ix0 - cx, v0 = (float)iy0 - cy, u1 = (float)ix1 - cx, v1 = (float)iy1 - cy;
        const float jx0 = 2.f * g0 * u0 * scale, jx1 = 2.f * g0 * (-v0) * scale, jx2 = 2.f * g0, jx3 = 0.f;
        const float jy0 = 2.f * g1 * v1 * scale, jy1 = 2.f * g1 * u1 * scale, jy2 = 0.f, jy3 = 2.f * g1;
        if (aos) {
            ((float4*)jx)[tile] = make_float4(jx0, jx1, jx2, jx3);
            ((float4*)jy)[tile] = make_float4(jy0, jy1, jy2, jy3);
        } else {
            jx[tile] = jx0; jx[nt + tile] = jx1; jx[2 * nt + tile] = jx2; jx[3 * nt + tile] = jx3;
            jy[tile] = jy0; jy[nt + tile] = jy1; jy[2 * nt + tile] = jy2; jy[3 * nt + tile] = jy3;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// The same pass for the tile sizes the reference instantiates (even 2..20, CMakeLists.txt:212-253), laid out for
// coalescing: a wave owns a horizontal strip of tiles, one lane per group of GW = 4 (2 when TS % 4 != 0) adjacent columns,
// and walks down the TS rows of the strip.  Per row EVERY lane issues one load of GW bytes at (scalar row pointer + its own
// constant byte offset): lanes 1 .. n_live carry the tiles' column groups, lane 0 and lane n_live + 1 fetch the words that
// hold the strip's left / right neighbour columns (at the image border: the word that holds the clamped column), so the
// pixels left and right of a group always come from the neighbouring lanes (wave_shr / wave_shl DPP moves) through a
// per-lane v_perm selector that is fixed before the loop -- no per-row border cases, no address arithmetic on the vector
// unit.  Rows above / below live in a 3-row window of registers (the row loop is unrolled by three so that the window
// rotates by renaming), every row is loaded once (+ 2 halo rows per strip).  Bytes are moved to bits 16..23 with
// v_perm_b32 and each arg-max key (|a - b| << 16) | (0xffff - (ry * TS + k)) is ONE v_sad_u32 whose addend is a scalar
// register (the row term is wave-uniform); a lane's column offset is subtracted once after the loop, which gives the
// tile-scan key of the generic kernel.  The TS / GW lanes of a tile meet in an LDS atomic max.
// Round 2: 43 -> 23 vector instructions per row and lane (the border loads, their branches and the 64-bit address
// arithmetic went away).
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned sad_u32_s(unsigned a, unsigned b, unsigned c) {      // |a - b| + c, c wave-uniform
    unsigned r;
    asm("v_sad_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "s"(c));
    return r;
}
__device__ __forceinline__ unsigned max3u(unsigned a, unsigned b, unsigned c) { return max(max(a, b), c); }

constexpr int keyframe_rows_tiles_per_wave(int ts) { return 62 / (ts / ((ts % 4 == 0) ? 4 : 2)); }

// body of one 256-thread workgroup (4 strips); `block` = the workgroup's index within its level, pointers already at the frame
template <int TS>
__device__ __forceinline__ void keyframe_rows_body(const uint8_t* __restrict__ img, int w, int h, int stride, int tx, int ty,
                                                   uint16_t* __restrict__ lmx, uint16_t* __restrict__ lmy, float* __restrict__ jx,
                                                   float* __restrict__ jy, int strips_x, int block, unsigned (*s_key)[64][2], bool aos) {
    constexpr int GW = (TS % 4 == 0) ? 4 : 2;    // columns per lane
    constexpr int LPT = TS / GW;                 // lanes per tile
    constexpr int TPW = keyframe_rows_tiles_per_wave(TS);   // tiles per wave (a strip of TPW tiles along x) + two halo lanes
    ((unsigned*)s_key)[threadIdx.x] = 0u;
    ((unsigned*)s_key)[threadIdx.x + 256] = 0u;
    __syncthreads();
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
    const int strip = block * 4 + wave;          // strip index: tile row tyi, TPW tiles starting at tile column sx0
    const int tyi = __builtin_amdgcn_readfirstlane(strip / strips_x);
    const int sx0 = (strip - tyi * strips_x) * TPW;
    const int n_live = min(TPW, tx - sx0) * LPT; // lanes 1 .. n_live carry column groups, adjacent in x
    const bool wave_live = tyi < ty && n_live > 0;            // wave-uniform
    const int li = lane - 1, tw = max(li, 0) / LPT, g = li - tw * LPT;
    const int txi = sx0 + tw;
    const bool live = wave_live && lane >= 1 && lane <= n_live;
    const int bx = txi * TS, by = tyi * TS, cx = g * GW;
    if (wave_live) {
        const int x_first = sx0 * TS, x = x_first + li * GW;
        const int off_lo = max(x_first - GW, 0), off_hi = min(x_first + n_live * GW, w - GW);     // the two halo words
        const int off = lane == 0 ? off_lo : (lane <= n_live ? x : off_hi);
        const int off_left = lane <= 1 ? off_lo : x - GW, off_right = lane >= n_live ? off_hi : x + GW;
        // v_perm selectors {0, byte, 0, 0}: in(x - 1) out of the left neighbour's word, in(x + GW) out of the right one's
        const unsigned sel_l = 0x0c000c0cu | ((unsigned)((max(x - 1, 0) - off_left) & 3) << 16);
        const unsigned sel_r = 0x0c000c0cu | ((unsigned)((min(x + GW, w - 1) - off_right) & 3) << 16);
        auto row_at = [&](int j) -> uint32_t {                     // row by + j (clamped): scalar pointer + lane offset
            int y;                                                  // clampi(by + j, 0, h - 1), kept on the scalar unit
            asm("s_max_i32 %0, %1, 0\n\ts_min_i32 %0, %0, %2" : "=&s"(y) : "s"(__builtin_amdgcn_readfirstlane(by + j)), "s"(__builtin_amdgcn_readfirstlane(h - 1)) : "scc");
            const uint8_t* rp = img + ((uint32_t)(y * stride) + (uint32_t)off);      // (a level is far below 4 GB)
            if (GW == 4) { uint32_t v; __builtin_memcpy(&v, rp, 4); return v; }
            uint16_t t; __builtin_memcpy(&t, rp, 2); return t;
        };
        unsigned mx = 0, my = 0;
        unsigned r0[GW], r1[GW], r2[GW];         // in(x + k, row) << 16 of the two rows above the arriving one (rotating)
        // one arriving row: p[k] = in(x - 1 + k, by + j) << 16; gx of row j from p, gy of row j - 1 from p and row j - 2 (= A)
#define VS_KF_STEP(WORD, J, A, T, DO_GX, DO_GY)                                                                         \
        {                                                                                                               \
            const uint32_t own_ = (WORD);                                                                               \
            const uint32_t left_ = (uint32_t)dpp_mov0<0x138>((int)own_);     /* wave_shr:1 -> lane - 1 */               \
            const uint32_t right_ = (uint32_t)dpp_mov0<0x130>((int)own_);    /* wave_shl:1 -> lane + 1 */               \
            unsigned p_[GW + 2];                                                                                        \
            p_[0] = __builtin_amdgcn_perm(0u, left_, sel_l);                                                            \
            _Pragma("unroll") for (int k = 0; k < GW; k++) p_[k + 1] = __builtin_amdgcn_perm(0u, own_, 0x0c000c0cu | ((unsigned)k << 16)); \
            p_[GW + 1] = __builtin_amdgcn_perm(0u, right_, sel_r);                                                      \
            if (DO_GX) {                                                                                                \
                const unsigned c_ = 0xffffu - (unsigned)((J) * TS);                                                     \
                mx = max3u(mx, sad_u32_s(p_[2], p_[0], c_), sad_u32_s(p_[3], p_[1], c_ - 1u));                          \
                if (GW == 4) mx = max3u(mx, sad_u32_s(p_[GW], p_[GW - 2], c_ - 2u), sad_u32_s(p_[GW + 1], p_[GW - 1], c_ - 3u)); \
            }                                                                                                           \
            if (DO_GY) {                                                                                                \
                const unsigned c_ = 0xffffu - (unsigned)(((J) - 1) * TS);                                               \
                my = max3u(my, sad_u32_s(p_[1], A[0], c_), sad_u32_s(p_[2], A[1], c_ - 1u));                            \
                if (GW == 4) my = max3u(my, sad_u32_s(p_[GW - 1], A[GW - 2], c_ - 2u), sad_u32_s(p_[GW], A[GW - 1], c_ - 3u)); \
            }                                                                                                           \
            _Pragma("unroll") for (int k = 0; k < GW; k++) T[k] = p_[k + 1];                                            \
        }
        // three rows are in flight before the first one is needed and every step refills the slot it consumed.  The loop
        // stays rolled: fully unrolled, the compiler hoists all TS + 2 row loads and the kernel drops to 2 waves / SIMD.
        const uint32_t w_m1 = row_at(-1), w_0 = row_at(0);
        uint32_t ring[3] = {row_at(1), row_at(2), row_at(3)};
        VS_KF_STEP(w_m1, -1, r0, r0, false, false)
        VS_KF_STEP(w_0, 0, r0, r1, true, false)
        constexpr int TRIPLES = (TS - 1) / 3, REM = (TS - 1) % 3;
        int j = 1;
#pragma unroll 1
        for (int t = 0; t < TRIPLES; t++, j += 3) {
            VS_KF_STEP(ring[0], j, r0, r2, true, true)
            ring[0] = row_at(j + 3);
            VS_KF_STEP(ring[1], j + 1, r1, r0, true, true)
            ring[1] = row_at(j + 4);
            VS_KF_STEP(ring[2], j + 2, r2, r1, true, true)
            ring[2] = row_at(j + 5);
        }
        // the rows left over (j = TS - REM .. TS - 1) and the row below the tile (j = TS: gy of the last row only)
        if (REM == 0) {
            VS_KF_STEP(ring[0], TS, r0, r2, false, true)
        } else if (REM == 1) {
            VS_KF_STEP(ring[0], TS - 1, r0, r2, true, true)
            VS_KF_STEP(ring[1], TS, r1, r0, false, true)
        } else {
            VS_KF_STEP(ring[0], TS - 2, r0, r2, true, true)
            VS_KF_STEP(ring[1], TS - 1, r1, r0, true, true)
            VS_KF_STEP(ring[2], TS, r2, r1, false, true)
        }
#undef VS_KF_STEP
        if (live) {
            atomicMax(&s_key[wave][tw][0], mx - (unsigned)cx);     // key = (g << 16) | (0xffff - (ry * TS + cx + k))
            atomicMax(&s_key[wave][tw][1], my - (unsigned)cx);
        }
    }
    __syncthreads();
    if (live && g == 0) {
        const int tile = tyi * tx + txi;
        const unsigned kx = s_key[wave][tw][0], ky = s_key[wave][tw][1];
        const size_t nt = (size_t)tx * ty;
        const int sx = (int)(0xffffu - (kx & 0xffffu)), sy = (int)(0xffffu - (ky & 0xffffu));
        const int ix0 = bx + sx % TS, iy0 = by + sx / TS, ix1 = bx + sy % TS, iy1 = by + sy / TS;
        if (aos) {     // engine tables: one {x, y} pair / one float4 per tile and set (a gather then touches one line, not six)
            ((uint32_t*)lmx)[tile] = (uint32_t)ix0 | ((uint32_t)iy0 << 16);
            ((uint32_t*)lmy)[tile] = (uint32_t)ix1 | ((uint32_t)iy1 << 16);
        } else {
            lmx[tile] = (uint16_t)ix0; lmx[nt + tile] = (uint16_t)iy0;
            lmy[tile] = (uint16_t)ix1; lmy[nt + tile] = (uint16_t)iy1;
        }
        // generators.cpp:346-385 (the min(.., w-1) clamps are no-ops: keypoints lie inside the image)
        const float cxf = (float)w * 0.5f, cyf = (float)h * 0.5f, scale = 1.f / (float)w;
        const float g0 = 0.5f * ((float)img[(size_t)iy0 * stride + min(ix0 + 1, w - 1)] - (float)img[(size_t)iy0 * stride + max(ix0 - 1, 0)]);
        const float g1 = 0.5f * ((float)img[(size_t)min(iy1 + 1, h - 1) * stride + ix1] - (float)img[(size_t)max(iy1 - 1, 0) * stride + ix1]);
        const float u0 = (float)ix0 - cxf, v0 = (float)iy0 - cyf, u1 = (float)ix1 - cxf, v1 = (float)iy1 - cyf;
        const float jx0 = 2.f * g0 * u0 * scale, jx1 = 2.f * g0 * (-v0) * scale, jx2 = 2.f * g0, jx3 = 0.f;
        const float jy0 = 2.f * g1 * v1 * scale, jy1 = 2.f * g1 * u1 * scale, jy2 = 0.f, jy3 = 2.f * g1;
        if (aos) {
            ((float4*)jx)[tile] = make_float4(jx0, jx1, jx2, jx3);
            ((float4*)jy)[tile] = make_float4(jy0, jy1, jy2, jy3);
        } else {
            jx[tile] = jx0; jx[nt + tile] = jx1; jx[2 * nt + tile] = jx2; jx[3 * nt + tile] = jx3;
            jy[tile] = jy0; jy[nt + tile] = jy1; jy[2 * nt + tile] = jy2; jy[3 * nt + tile] = jy3;
        }
    }
}

template <int TS>
__global__ __launch_bounds__(256) void vs_k_keyframe_rows(const uint8_t* __restrict__ img, int w, int h, int stride, int tx,
                                                          int ty, uint16_t* __restrict__ lmx, uint16_t* __restrict__ lmy,
                                                          float* __restrict__ jx, float* __restrict__ jy,
                                                          size_t img_frame_stride, size_t lm_frame_stride,
                                                          size_t jac_frame_stride, int strips_x, bool aos) {
    __shared__ unsigned s_key[4][64][2];
    keyframe_rows_body<TS>(img + blockIdx.y * img_frame_stride, w, h, stride, tx, ty, lmx + blockIdx.y * lm_frame_stride,
                           lmy + blockIdx.y * lm_frame_stride, jx + blockIdx.y * jac_frame_stride, jy + blockIdx.y * jac_frame_stride,
                           strips_x, (int)blockIdx.x, s_key, aos);
}

// The keyframe pass of EVERY pyramid level of every keyframe in one launch (alignment.cpp:237-276 loops GradXY -> GradArgMax ->
// SparseJacobian over the levels; here: blockIdx.x walks the levels' workgroups back to back, blockIdx.y = keyframe).  Levels
// are dense images inside one pyramid slot, keypoint / Jacobian tables likewise (vs_engine.hip "Data layout").  The tile size
// differs per level (imgproc.cpp:151-162), so the body is selected by a switch over the ten sizes the rule can return; the
// branch is workgroup-uniform.
__global__ __launch_bounds__(256) void vs_k_keyframe_levels(const uint8_t* __restrict__ pyr, uint16_t* __restrict__ lm,
                                                            float* __restrict__ jac, size_t pyr_frame_stride,
                                                            size_t lm_frame_stride, size_t jac_frame_stride, vsk::KeyframeLevels L) {
    __shared__ unsigned s_key[4][64][2];
    int l = 0, block = (int)blockIdx.x;
    while (l + 1 < L.n && block >= L.lv[l].blocks) { block -= L.lv[l].blocks; l++; }     // scalar: <= 16 levels
    const vsk::KeyframeLevel& q = L.lv[l];
    const size_t nt = (size_t)q.tx * q.ty;
    const uint8_t* img = pyr + blockIdx.y * pyr_frame_stride + q.img_off;
    uint16_t* lmx = lm + blockIdx.y * lm_frame_stride + q.lm_off;
    float* jx = jac + blockIdx.y * jac_frame_stride + q.jac_off;
#define VS_KF(TS) case TS: keyframe_rows_body<TS>(img, q.w, q.h, q.w, q.tx, q.ty, lmx, lmx + 2 * nt, jx, jx + 4 * nt, q.strips_x, block, s_key, true); break;
    switch (q.ts) { VS_KF(2) VS_KF(4) VS_KF(6) VS_KF(8) VS_KF(10) VS_KF(12) VS_KF(14) VS_KF(16) VS_KF(18) VS_KF(20) default: break; }
#undef VS_KF
}

// ------------------------------------------------------------------------------------------------
// sparse_warpdiff: generators.cpp:646-700.  One thread per tile keypoint.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void vs_k_sparse_warpdiff(const uint8_t* __restrict__ tmpl,
                                                            const uint8_t* __restrict__ key, int w, int h, int stride,
                                                            const uint16_t* __restrict__ lm, int nt, float A, float B,
                                                            float TX, float TY, uint16_t* __restrict__ out) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= nt) return;
    int tile_x = min((int)lm[i], w - 1), tile_y = min((int)lm[nt + i], h - 1);
    float ox = (float)tile_x, oy = (float)tile_y;
    float Wx = (1.0f + A) * ox - B * oy + TX;
    float Wy = B * ox + (1.0f + A) * oy + TY;
    float v = lanczos_sample_u8(key, w, h, stride, Wx, Wy);
    float diff = fabsf(v - (float)tmpl[(size_t)tile_y * stride + tile_x]);
    diff = fminf(fmaxf(diff, 0.0f), 65535.0f);
    out[i] = (uint16_t)diff;
}

// ------------------------------------------------------------------------------------------------
// sparse_ica: generators.cpp:429-596.  One block; each thread walks its strided share of both
// point sets, accumulating double(J*residual) per set, then a block tree-sum.  The reference
// reduces serially in index order; the tree differs from that by ~1e-16 relative (fp64 sums of
// fp32 products), far inside the 1e-4 parameter gate.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void ica_accumulate(const uint8_t* __restrict__ tmpl, const uint8_t* __restrict__ key,
                                               int w, int h, int stride, const uint16_t* __restrict__ sel, int n,
                                               const float* __restrict__ jac, float A1, float B, float TX, float TY,
                                               double acc[4]) {
    for (int r = threadIdx.x; r < n; r += blockDim.x) {
        int px = sel[r], py = sel[n + r];
        float ox = (float)px, oy = (float)py;
        float Wx = A1 * ox - B * oy + TX;
        float Wy = B * ox + A1 * oy + TY;
        float warped = lanczos_sample_u8(key, w, h, stride, Wx, Wy);
        float tv = (float)tmpl[(size_t)min(py, h - 1) * stride + min(px, w - 1)];
        float residual = tv - warped;
#pragma unroll
        for (int c = 0; c < 4; c++) acc[c] += (double)(jac[(size_t)c * n + r] * residual);
    }
}

__global__ __launch_bounds__(1024) void vs_k_sparse_ica(const uint8_t* __restrict__ tmpl, const uint8_t* __restrict__ key,
                                                        int w, int h, int stride, const uint16_t* __restrict__ selx,
                                                        int nx, const uint16_t* __restrict__ sely, int ny,
                                                        const float* __restrict__ jacx, const float* __restrict__ jacy,
                                                        float A, float B, float TX, float TY, double* __restrict__ out) {
    __shared__ double red[16 * 8];
    double acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const float A1 = 1.0f + A;
    ica_accumulate(tmpl, key, w, h, stride, selx, nx, jacx, A1, B, TX, TY, acc);
    ica_accumulate(tmpl, key, w, h, stride, sely, ny, jacy, A1, B, TX, TY, acc + 4);
    block_sum<8>(acc, red);
    if (threadIdx.x < 4) out[threadIdx.x] = (acc[threadIdx.x] + acc[4 + threadIdx.x]) * 0.5f;
}

// ------------------------------------------------------------------------------------------------
// image_warp: generators.cpp:126-164.  Bilinear, u8 -> f32, clamp-to-edge.
// Halide float lerp(a,b,t) = a*(1-t) + b*t.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float lerpf(float a, float b, float t) { return a * (1.0f - t) + b * t; }

__global__ __launch_bounds__(256) void vs_k_image_warp(const uint8_t* __restrict__ in, int w, int h, int stride,
                                                       float A, float B, float TX, float TY, float* __restrict__ out,
                                                       int ow, int oh) {
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (x >= ow) return;
    float Wx = (1.0f + A) * (float)x - B * (float)y + TX;
    float Wy = B * (float)x + (1.0f + A) * (float)y + TY;
    // generators.cpp:150-153 takes the fraction from the converted index, W - float(int(floor(W))): the same value as W - floor(W) for every
    // position inside the int range, and undefined outside it -- every sampler here and in the oracle uses W - floor(W), defined everywhere
    const float flx = floorf(Wx), fly = floorf(Wy);
    float wx = Wx - flx, wy = Wy - fly;
    int fx = sample_index(flx, w), fy = sample_index(fly, h);  // (the taps below cannot overflow)
    int x0 = clampi(fx, 0, w - 1), x1 = clampi(fx + 1, 0, w - 1);
    const uint8_t* r0 = in + (size_t)clampi(fy, 0, h - 1) * stride;
    const uint8_t* r1 = in + (size_t)clampi(fy + 1, 0, h - 1) * stride;
    float top = lerpf((float)r0[x0], (float)r0[x1], wx);
    float bottom = lerpf((float)r1[x0], (float)r1[x1], wx);
    out[(size_t)y * ow + x] = lerpf(top, bottom, wy);
}

// ------------------------------------------------------------------------------------------------
// bgr_image_warp (general path): any channel count, u8/u16, Lanczos2 or bilinear, clamp or
// constant border, integer or float output.  One thread per output pixel, taps through L1/L2.
// The sampler is the reference's (generators.cpp:672-697 / 148-163) applied per channel with
// image_warp's coordinates (generators.cpp:141-142).  The tuned u8 BGR kernels live in
// vs_warp.hip; this one is the semantic baseline they are tested against.
// ------------------------------------------------------------------------------------------------
template <typename T, int MODE, int BORDER, bool F32OUT>
__global__ __launch_bounds__(256) void vs_k_bgr_warp_generic(const T* __restrict__ src, int w, int h, int src_stride,
                                                             int channels, const float4* __restrict__ params,
                                                             int max_value, void* __restrict__ dstv, int dst_stride,
                                                             size_t src_frame_stride, size_t dst_frame_stride, vsk::Roi roi) {
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (x >= roi.w) return;
    const float4 P = params[blockIdx.z];
    src += blockIdx.z * src_frame_stride;
    const float A = P.x, B = P.y, TX = P.z, TY = P.w;
    const float fx = (float)(x + roi.x), fy = (float)(y + roi.y);   // full-frame coordinate of this window pixel
    float Wx = (1.0f + A) * fx - B * fy + TX;
    float Wy = B * fx + (1.0f + A) * fy + TY;
    float flx = floorf(Wx), fly = floorf(Wy);
    int ix = sample_index(flx, w), iy = sample_index(fly, h);
    float frx = Wx - flx, fry = Wy - fly;
    float wx[4], wy[4];
    if (MODE == 0) { lanczos_weights4(frx, wx); lanczos_weights4(fry, wy); }
    if (MODE == 2 || MODE == 3) { lanczos_weights4_fma(frx, wx); lanczos_weights4_fma(fry, wy); }
    const float rden = MODE == 3 ? 1.0f / lanczos_separable_den(wx, wy) : 0.0f;      // IEEE divide: the correctly rounded reciprocal
    for (int c = 0; c < channels; c++) {
        float v;
        if (MODE == 2 || MODE == 3) {
            // VS_WARP_LANCZOS2_FAST / VS_WARP_LANCZOS2_SEP (vs_device.hpp): the contracted and the separable sampler, here with float output too
            float t[4][4];
#pragma unroll
            for (int ry = 0; ry < 4; ry++) {
                int sy = iy + ry - 1;
#pragma unroll
                for (int rx = 0; rx < 4; rx++) {
                    int sx = ix + rx - 1;
                    if (BORDER == 1) {
                        t[ry][rx] = (sx < 0 || sy < 0 || sx >= w || sy >= h) ? 0.0f
                                    : (float)src[(size_t)sy * src_stride + (size_t)sx * channels + c];
                    } else {
                        t[ry][rx] = (float)src[(size_t)clampi(sy, 0, h - 1) * src_stride + (size_t)clampi(sx, 0, w - 1) * channels + c];
                    }
                }
            }
            v = MODE == 3 ? lanczos_separable_combine(t, wx, wy, rden) : lanczos_contracted_combine(t, wx, wy);
        } else if (MODE == 0) {
            float num = 0.0f, den = 0.0f;
#pragma unroll
            for (int ry = 0; ry < 4; ry++) {
                int sy = iy + ry - 1;
#pragma unroll
                for (int rx = 0; rx < 4; rx++) {
                    int sx = ix + rx - 1;
                    float w2d = wx[rx] * wy[ry];
                    float val;
                    if (BORDER == 1) {
                        val = (sx < 0 || sy < 0 || sx >= w || sy >= h) ? 0.0f
                              : (float)src[(size_t)sy * src_stride + (size_t)sx * channels + c];
                    } else {
                        val = (float)src[(size_t)clampi(sy, 0, h - 1) * src_stride + (size_t)clampi(sx, 0, w - 1) * channels + c];
                    }
                    num = num + w2d * val;
                    den = den + w2d;
                }
            }
            v = num / den;
        } else {
            float t[4];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                int sx = ix + (k & 1), sy = iy + (k >> 1);
                if (BORDER == 1) {
                    t[k] = (sx < 0 || sy < 0 || sx >= w || sy >= h) ? 0.0f
                           : (float)src[(size_t)sy * src_stride + (size_t)sx * channels + c];
                } else {
                    t[k] = (float)src[(size_t)clampi(sy, 0, h - 1) * src_stride + (size_t)clampi(sx, 0, w - 1) * channels + c];
                }
            }
            v = lerpf(lerpf(t[0], t[1], frx), lerpf(t[2], t[3], frx), fry);
        }
        size_t o = (size_t)y * dst_stride + (size_t)x * channels + c;
        if (F32OUT) {
            ((float*)dstv + blockIdx.z * dst_frame_stride)[o] = v;
        } else {
            float r = floorf(v + 0.5f);
            r = fminf(fmaxf(r, 0.0f), (float)max_value);
            ((T*)dstv + blockIdx.z * dst_frame_stride)[o] = (T)r;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// VS_WARP_BILINEAR_CV (general path): cv::warpAffine(INTER_LINEAR)'s fixed-point bilinear, one thread per output pixel, any channel
// count, u8 (integer weights, (sum + 2^14) >> 15) or u16 containers (float weights, cvRound).  Restates OpenCV 4.x imgwarp.cpp
// (WarpAffineInvoker + remapBilinear) exactly as the oracle's cv_warp_impl does; M = the frame's output -> source matrix.
// ------------------------------------------------------------------------------------------------
template <typename T, int BORDER>
__global__ __launch_bounds__(256) void vs_k_bgr_warp_cv_generic(const T* __restrict__ src, int w, int h, int src_stride, int channels,
                                                                const double* __restrict__ minv, int max_value, T* __restrict__ dst,
                                                                int dst_stride, size_t src_fs, size_t dst_fs, vsk::Roi roi) {
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (x >= roi.w) return;
    const double* M = minv + 6 * (size_t)blockIdx.z;
    src += blockIdx.z * src_fs;
    dst += blockIdx.z * dst_fs;
    const int fx_ = x + roi.x, fy_ = y + roi.y;                       // full-frame coordinate of this window pixel
    const int X = (int)((unsigned)cv_row_origin(M[1], M[2], fy_) + (unsigned)cv_delta(M[0], fx_)) >> 5;
    const int Y = (int)((unsigned)cv_row_origin(M[4], M[5], fy_) + (unsigned)cv_delta(M[3], fx_)) >> 5;
    const int sx = clampi(X >> 5, -32768, 32767), sy = clampi(Y >> 5, -32768, 32767);      // saturate_cast<short>
    const int a1 = X & 31, b1 = Y & 31, a0 = 32 - a1, b0 = 32 - b1;
    auto tap = [&](int xx, int yy, int c) -> int {
        if (BORDER == 1) { if (xx < 0 || yy < 0 || xx >= w || yy >= h) return 0; }
        else { xx = clampi(xx, 0, w - 1); yy = clampi(yy, 0, h - 1); }
        return (int)src[(size_t)yy * src_stride + (size_t)xx * channels + c];
    };
    for (int c = 0; c < channels; c++) {
        const int v00 = tap(sx, sy, c), v01 = tap(sx + 1, sy, c), v10 = tap(sx, sy + 1, c), v11 = tap(sx + 1, sy + 1, c);
        int r;
        if (sizeof(T) == 1) {
            r = (v00 * (a0 * b0 * 32) + v01 * (a1 * b0 * 32) + v10 * (a0 * b1 * 32) + v11 * (a1 * b1 * 32) + (1 << 14)) >> 15;
        } else {
            const float k = 1.0f / 1024.0f;
            const float sum = (float)v00 * ((float)(a0 * b0) * k) + (float)v01 * ((float)(a1 * b0) * k) + (float)v10 * ((float)(a0 * b1) * k) +
                              (float)v11 * ((float)(a1 * b1) * k);
            r = (int)rintf(sum);
        }
        dst[(size_t)y * dst_stride + (size_t)x * channels + c] = (T)min(max(r, 0), max_value);
    }
}

// ------------------------------------------------------------------------------------------------
// Counter calibration: a plain copy with the warp kernel's access width (12 bytes per lane,
// global_load/store_dwordx3).  MI355X_MICROARCH.md says FETCH_SIZE is only calibrated for 16-byte
// streams on gfx950 and asks for a known-byte-count run in the kernel's own access pattern.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void vs_k_calib_copy12(const uint32_t* __restrict__ src, uint32_t* __restrict__ dst,
                                                         size_t n_groups) {
    typedef uint32_t u32x3 __attribute__((ext_vector_type(3)));
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n_groups; i += (size_t)gridDim.x * 256)
        *(u32x3*)(dst + 3 * i) = *(const u32x3*)(src + 3 * i);
}

// ================================================================================================
// launchers
// ================================================================================================
namespace vsk {

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }

hipError_t calib_copy12(const void* src, void* dst, size_t bytes, hipStream_t s) {
    hipLaunchKernelGGL(vs_k_calib_copy12, dim3(4096), dim3(256), 0, s, (const uint32_t*)src, (uint32_t*)dst, bytes / 12);
    return hipGetLastError();
}

hipError_t pyr_down(const uint8_t* in, int w, int h, int in_stride, uint8_t* out, int ow, int oh, int out_stride,
                    int n_frames, size_t in_fs, size_t out_fs, hipStream_t s) {
    if (ow <= 0 || oh <= 0 || n_frames <= 0) return hipSuccess;
    // a wave's walk is a chain of dependent row loads: with few frames in flight shorter bands finish sooner (one 960x540
    // frame: 21 -> 9 us), with many the long bands re-read fewer halo rows
    const int strips_x = cdiv(ow, 2 * PR_LIVE);
    const int bands = cdiv(oh, n_frames <= 4 ? 4 : PR_OUT);
    const dim3 grid(cdiv(strips_x * bands, 4), n_frames);
#define VS_PD(ROWS, NARROW)                                                                                             \
    hipLaunchKernelGGL((vs_k_pyr_down_rows<ROWS, NARROW>), grid, dim3(256), 0, s, in, w, h, in_stride, out, ow, oh, out_stride, in_fs, \
                       out_fs, strips_x, bands)
    if (n_frames <= 4) { if (w < 4) VS_PD(4, true); else VS_PD(4, false); }
    else { if (w < 4) VS_PD(PR_OUT, true); else VS_PD(PR_OUT, false); }
#undef VS_PD
    return hipGetLastError();
}

hipError_t bgr_to_gray(const void* src, int w, int h, int src_stride, int bits, int shift_to_8, uint8_t* dst,
                       int dst_stride, int n_frames, size_t src_fs, size_t dst_fs, hipStream_t s) {
    dim3 grid(cdiv(w, 1024), h, n_frames);
    if (bits == 8)
        hipLaunchKernelGGL(vs_k_bgr_to_gray<uint8_t>, grid, dim3(256), 0, s, (const uint8_t*)src, w, h, src_stride,
                           shift_to_8, dst, dst_stride, src_fs, dst_fs);
    else
        hipLaunchKernelGGL(vs_k_bgr_to_gray<uint16_t>, grid, dim3(256), 0, s, (const uint16_t*)src, w, h, src_stride,
                           shift_to_8, dst, dst_stride, src_fs, dst_fs);
    return hipGetLastError();
}

hipError_t ingest_pyr(const void* src, int w, int h, int src_stride, int bits, int shift_to_8, uint8_t* g0, uint8_t* g1,
                      int n_frames, size_t src_fs, size_t pyr_fs, hipStream_t s) {
    const int ow = w / 2, oh = h / 2;
    const int tiles_x = cdiv(w, 2 * PD_TW), tiles_y = cdiv(h, 2 * PD_TH);
    const long long tpf = (long long)tiles_x * tiles_y;
    if (tpf > 0x3fffffLL || tiles_x > 1024) return hipErrorNotSupported;      // (tl * tiles_x < 2^32 for the multiply-high below)
    const int chunk = (int)((tpf + 7) / 8);
    const uint32_t magic = (uint32_t)(0x100000000ULL / (uint32_t)tiles_x) + 1u;    // tiles_x >= 2; the kernel special-cases 1
    const size_t esz = bits == 8 ? 1 : 2;
    for (int f0 = 0; f0 < n_frames; f0 += 65535) {          // gridDim.y limit
        const int nf = std::min(65535, n_frames - f0);
        const dim3 grid((unsigned)(chunk * 8), (unsigned)nf);
        const uint8_t* sp = (const uint8_t*)src + (size_t)f0 * src_fs * esz;
        if (bits == 8)
            hipLaunchKernelGGL(vs_k_ingest_pyr<uint8_t>, grid, dim3(256), 0, s, (const uint8_t*)sp, w, h, src_stride, shift_to_8,
                               g0 + (size_t)f0 * pyr_fs, g1 + (size_t)f0 * pyr_fs, ow, oh, src_fs, pyr_fs, tiles_x, magic, (int)tpf, chunk);
        else
            hipLaunchKernelGGL(vs_k_ingest_pyr<uint16_t>, grid, dim3(256), 0, s, (const uint16_t*)sp, w, h, src_stride, shift_to_8,
                               g0 + (size_t)f0 * pyr_fs, g1 + (size_t)f0 * pyr_fs, ow, oh, src_fs, pyr_fs, tiles_x, magic, (int)tpf, chunk);
    }
    return hipGetLastError();
}

hipError_t grad_xy(const uint8_t* in, int w, int h, int stride, float* gx, float* gy, hipStream_t s) {
    hipLaunchKernelGGL(vs_k_grad_xy, dim3(cdiv(w, 256), h), dim3(256), 0, s, in, w, h, stride, gx, gy);
    return hipGetLastError();
}

hipError_t grad_argmax(const float* gx, const float* gy, int w, int h, int ts, uint16_t* lmx, uint16_t* lmy,
                       hipStream_t s) {
    int tx = w / ts, ty = h / ts;
    if (tx * ty == 0) return hipSuccess;
    hipLaunchKernelGGL(vs_k_grad_argmax, dim3(cdiv(tx * ty, 4)), dim3(256), 0, s, gx, gy, w, h, ts, tx, ty, lmx, lmy);
    return hipGetLastError();
}

hipError_t sparse_jac(const float* gx, const float* gy, int w, int h, const uint16_t* lmx, const uint16_t* lmy,
                      int nt, float* jx, float* jy, hipStream_t s) {
    if (nt == 0) return hipSuccess;
    hipLaunchKernelGGL(vs_k_sparse_jac, dim3(cdiv(nt, 256)), dim3(256), 0, s, gx, gy, w, h, lmx, lmy, nt, jx, jy);
    return hipGetLastError();
}

hipError_t keyframe(const uint8_t* img, int w, int h, int stride, int ts, uint16_t* lmx, uint16_t* lmy, float* jx,
                    float* jy, int n_frames, size_t img_fs, size_t lm_fs, size_t jac_fs, hipStream_t s, bool aos) {
    int tx = w / ts, ty = h / ts;
    if (tx * ty == 0) return hipSuccess;
#define VS_ROWS(TS)                                                                                                     \
    case TS: {                                                                                                          \
        constexpr int tpw = keyframe_rows_tiles_per_wave(TS);                                                           \
        const int strips_x = cdiv(tx, tpw);                                                                             \
        hipLaunchKernelGGL(vs_k_keyframe_rows<TS>, dim3(cdiv(strips_x * ty, 4), n_frames), dim3(256), 0, s, img, w, h,  \
                           stride, tx, ty, lmx, lmy, jx, jy, img_fs, lm_fs, jac_fs, strips_x, aos);                     \
        break;                                                                                                          \
    }
    switch (ts) {   // the sizes vs_tile_size can return (imgproc.cpp:151-162)
        VS_ROWS(2) VS_ROWS(4) VS_ROWS(6) VS_ROWS(8) VS_ROWS(10) VS_ROWS(12) VS_ROWS(14) VS_ROWS(16) VS_ROWS(18) VS_ROWS(20)
    default:
        hipLaunchKernelGGL(vs_k_keyframe, dim3(cdiv(tx * ty, 16), n_frames), dim3(256), 0, s, img, w, h, stride, ts, tx, ty,
                           lmx, lmy, jx, jy, img_fs, lm_fs, jac_fs, aos);
    }
#undef VS_ROWS
    return hipGetLastError();
}

// all levels of n_frames keyframes in one launch; false = a level's tile size has no row-walking body (caller launches per level)
bool keyframe_levels_supported(const KeyframeLevels& L) {
    for (int i = 0; i < L.n; i++)
        if (L.lv[i].ts < 2 || L.lv[i].ts > 20 || (L.lv[i].ts & 1) || L.lv[i].tx * L.lv[i].ty == 0) return false;
    return L.n >= 1 && L.n <= 16;
}
hipError_t keyframe_levels(const uint8_t* pyr, uint16_t* lm, float* jac, KeyframeLevels L, int n_frames, size_t pyr_fs, size_t lm_fs,
                           size_t jac_fs, hipStream_t s) {
    int total = 0;
    for (int i = 0; i < L.n; i++) {
        const int ts = L.lv[i].ts, tpw = keyframe_rows_tiles_per_wave(ts);
        L.lv[i].strips_x = cdiv(L.lv[i].tx, tpw);
        L.lv[i].blocks = cdiv(L.lv[i].strips_x * L.lv[i].ty, 4);
        total += L.lv[i].blocks;
    }
    hipLaunchKernelGGL(vs_k_keyframe_levels, dim3(total, n_frames), dim3(256), 0, s, pyr, lm, jac, pyr_fs, lm_fs, jac_fs, L);
    return hipGetLastError();
}

hipError_t sparse_warpdiff(const uint8_t* tmpl, const uint8_t* key, int w, int h, int stride, const uint16_t* lm,
                           int nt, float A, float B, float TX, float TY, uint16_t* out, hipStream_t s) {
    if (nt == 0) return hipSuccess;
    hipLaunchKernelGGL(vs_k_sparse_warpdiff, dim3(cdiv(nt, 256)), dim3(256), 0, s, tmpl, key, w, h, stride, lm, nt, A, B,
                       TX, TY, out);
    return hipGetLastError();
}

hipError_t sparse_ica(const uint8_t* tmpl, const uint8_t* key, int w, int h, int stride, const uint16_t* selx, int nx,
                      const uint16_t* sely, int ny, const float* jacx, const float* jacy, float A, float B, float TX,
                      float TY, double* out, hipStream_t s) {
    hipLaunchKernelGGL(vs_k_sparse_ica, dim3(1), dim3(1024), 0, s, tmpl, key, w, h, stride, selx, nx, sely, ny, jacx,
                       jacy, A, B, TX, TY, out);
    return hipGetLastError();
}

hipError_t image_warp(const uint8_t* in, int w, int h, int stride, float A, float B, float TX, float TY, float* out,
                      int ow, int oh, hipStream_t s) {
    hipLaunchKernelGGL(vs_k_image_warp, dim3(cdiv(ow, 256), oh), dim3(256), 0, s, in, w, h, stride, A, B, TX, TY, out, ow,
                       oh);
    return hipGetLastError();
}

template <typename T, bool F32OUT>
static void launch_generic(const T* src, int w, int h, int src_stride, int channels, const float4* params, int mode,
                           int border, int max_value, void* dst, int dst_stride, int n_frames, size_t src_fs,
                           size_t dst_fs, vsk::Roi roi, hipStream_t s) {
    dim3 grid(cdiv(roi.w, 256), roi.h, n_frames), block(256);
#define VS_LAUNCH(M, Bd)                                                                                          \
    hipLaunchKernelGGL((vs_k_bgr_warp_generic<T, M, Bd, F32OUT>), grid, block, 0, s, src, w, h, src_stride, channels, \
                       params, max_value, dst, dst_stride, src_fs, dst_fs, roi)
    if (mode == 0 && border == 0) VS_LAUNCH(0, 0);
    else if (mode == 0) VS_LAUNCH(0, 1);
    else if (mode == 2 && border == 0) VS_LAUNCH(2, 0);
    else if (mode == 2) VS_LAUNCH(2, 1);
    else if (mode == 3 && border == 0) VS_LAUNCH(3, 0);
    else if (mode == 3) VS_LAUNCH(3, 1);
    else if (border == 0) VS_LAUNCH(1, 0);
    else VS_LAUNCH(1, 1);
#undef VS_LAUNCH
}

hipError_t bgr_warp_generic(const void* src, int w, int h, int src_stride, int channels, int bits,
                            const float4* params_dev, int mode, int border, int max_value, void* dst, int dst_stride,
                            bool f32out, int n_frames, size_t src_fs, size_t dst_fs, Roi roi, hipStream_t s) {
    if (bits == 8) {
        if (f32out) launch_generic<uint8_t, true>((const uint8_t*)src, w, h, src_stride, channels, params_dev, mode, border, max_value, dst, dst_stride, n_frames, src_fs, dst_fs, roi, s);
        else launch_generic<uint8_t, false>((const uint8_t*)src, w, h, src_stride, channels, params_dev, mode, border, max_value, dst, dst_stride, n_frames, src_fs, dst_fs, roi, s);
    } else {
        if (f32out) launch_generic<uint16_t, true>((const uint16_t*)src, w, h, src_stride, channels, params_dev, mode, border, max_value, dst, dst_stride, n_frames, src_fs, dst_fs, roi, s);
        else launch_generic<uint16_t, false>((const uint16_t*)src, w, h, src_stride, channels, params_dev, mode, border, max_value, dst, dst_stride, n_frames, src_fs, dst_fs, roi, s);
    }
    return hipGetLastError();
}

hipError_t bgr_warp_cv_generic(const void* src, int w, int h, int src_stride, int channels, int bits, const double* minv_dev, int border,
                               int max_value, void* dst, int dst_stride, int n_frames, size_t src_fs, size_t dst_fs, Roi roi, hipStream_t s) {
    dim3 grid(cdiv(roi.w, 256), roi.h, n_frames), block(256);
#define VS_LAUNCH(T, Bd) \
    hipLaunchKernelGGL((vs_k_bgr_warp_cv_generic<T, Bd>), grid, block, 0, s, (const T*)src, w, h, src_stride, channels, minv_dev, max_value, (T*)dst, \
                       dst_stride, src_fs, dst_fs, roi)
    if (bits == 8) { if (border == 0) VS_LAUNCH(uint8_t, 0); else VS_LAUNCH(uint8_t, 1); }
    else { if (border == 0) VS_LAUNCH(uint16_t, 0); else VS_LAUNCH(uint16_t, 1); }
#undef VS_LAUNCH
    return hipGetLastError();
}

}  // namespace vsk
