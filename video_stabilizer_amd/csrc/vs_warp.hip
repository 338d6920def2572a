// vs_warp.hip -- tuned bgr_image_warp for interleaved 3-channel u8 frames (the 1080p / 4K roofline kernel).
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
//   2. The footprint (+ the Lanczos halo) is copied HBM -> LDS once as raw interleaved bytes with
//      aligned dword loads; clamp-to-edge / constant-0 borders are resolved during this copy, so the
//      inner loop has no address clamps and no global loads at all.
//   3. Each thread produces 4 adjacent pixels of one row.  A tap row (4 px x BGR = 12 bytes at an
//      arbitrary byte offset) is 4 LDS dwords re-aligned with v_alignbyte_b32 and unpacked with
//      v_cvt_f32_ubyteN; all LDS offsets are immediates off one address register.
//   4. 12 output bytes per thread leave as one 12-byte store; a wave writes 4 x 192 contiguous bytes.
// Tiles whose footprint does not fit the LDS window (large rotation / zoom) take the generic
// global-memory path inside the same kernel, so every transform is supported.
//
// Cost model (DESIGN.md "bgr_image_warp roofline"): ~380 VALU instructions per output pixel in the
// exact-order Lanczos2 form (8 polynomial weights = 120, 16 taps x 3 channels mul+add = 128, byte
// unpack = 60, 3 IEEE divides = 30, ...), which is above the HBM time of the 6 bytes the pixel moves:
// the kernel is VALU-bound, not HBM-bound, on gfx950.
#include "vs_kernels.hpp"
#include "vs_device.hpp"

using namespace vsd;

namespace {

constexpr int WT_W = 64, WT_H = 16;      // output tile
constexpr int WS_BYTES = 240;            // staged bytes per source row (80 px)
constexpr int WS_PITCH_DW = 61;          // LDS row pitch in dwords (244 B: odd pitch spreads rows over banks)
constexpr int WS_H = 32;                 // staged source rows

__device__ __forceinline__ float lerpf(float a, float b, float t) { return a * (1.0f - t) + b * t; }

__device__ __forceinline__ float ub(uint32_t q, int k) { return (float)((q >> (8 * k)) & 0xffu); }

__device__ __forceinline__ uint32_t store_u8(float v) {
    float r = floorf(v + 0.5f);                 // build rule: round half up, saturate
    r = fminf(fmaxf(r, 0.0f), 255.0f);
    return (uint32_t)r;
}

// one pixel, all three channels, straight from global memory (footprint too large for the LDS window)
template <int MODE, int BORDER>
__device__ __forceinline__ void warp_pixel_global(const uint8_t* __restrict__ src, int w, int h, int stride, float Wx,
                                                  float Wy, uint32_t out[3]) {
    float flx = floorf(Wx), fly = floorf(Wy);
    int ix = (int)flx, iy = (int)fly;
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
        for (int c = 0; c < 3; c++) out[c] = store_u8(num[c] / den);
    } else {
#pragma unroll
        for (int c = 0; c < 3; c++) {
            float top = lerpf(fetch(ix, iy, c), fetch(ix + 1, iy, c), frx);
            float bottom = lerpf(fetch(ix, iy + 1, c), fetch(ix + 1, iy + 1, c), frx);
            out[c] = store_u8(lerpf(top, bottom, fry));
        }
    }
}

__device__ __forceinline__ int floor_div3(int v) { return v >= 0 ? v / 3 : -((2 - v) / 3); }

template <int MODE, int BORDER>
__global__ __launch_bounds__(256) void vs_k_bgr_warp_u8c3(const uint8_t* __restrict__ src, int w, int h, int src_stride,
                                                          const float4* __restrict__ params, uint8_t* __restrict__ dst,
                                                          int dst_stride, size_t src_fs, size_t dst_fs) {
    __shared__ uint32_t tile[WS_H * WS_PITCH_DW + 4];
    const float4 P = params[blockIdx.z];
    src += blockIdx.z * src_fs;
    dst += blockIdx.z * dst_fs;
    const float A1 = 1.0f + P.x, B = P.y, TX = P.z, TY = P.w;
    const int x0 = blockIdx.x * WT_W, y0 = blockIdx.y * WT_H;
    const int x1 = min(x0 + WT_W, w) - 1, y1 = min(y0 + WT_H, h) - 1;

    // source footprint of the tile: the four corners, evaluated with the per-pixel expression
    float fx0 = (float)x0, fx1 = (float)x1, fy0 = (float)y0, fy1 = (float)y1;
    float cxs[4] = {A1 * fx0 - B * fy0 + TX, A1 * fx1 - B * fy0 + TX, A1 * fx0 - B * fy1 + TX, A1 * fx1 - B * fy1 + TX};
    float cys[4] = {B * fx0 + A1 * fy0 + TY, B * fx1 + A1 * fy0 + TY, B * fx0 + A1 * fy1 + TY, B * fx1 + A1 * fy1 + TY};
    float mnx = fminf(fminf(cxs[0], cxs[1]), fminf(cxs[2], cxs[3])), mxx = fmaxf(fmaxf(cxs[0], cxs[1]), fmaxf(cxs[2], cxs[3]));
    float mny = fminf(fminf(cys[0], cys[1]), fminf(cys[2], cys[3])), mxy = fmaxf(fmaxf(cys[0], cys[1]), fmaxf(cys[2], cys[3]));
    bool fits = fabsf(mnx) < 1.0e6f && fabsf(mxx) < 1.0e6f && fabsf(mny) < 1.0e6f && fabsf(mxy) < 1.0e6f;
    int sx_lo = 0, sy_lo = 0, bx0 = 0;
    if (fits) {
        sx_lo = (int)floorf(mnx) - 1;
        const int sx_hi = (int)floorf(mxx) + 2;
        sy_lo = (int)floorf(mny) - 1;
        const int sy_hi = (int)floorf(mxy) + 2;
        bx0 = (sx_lo * 3) & ~3;                                   // first staged byte of every row (dword aligned)
        fits = (sx_hi * 3 + 2 - bx0 + 1) <= WS_BYTES && (sy_hi - sy_lo + 1) <= WS_H;
        if (fits) {
            const int rows = sy_hi - sy_lo + 1;
            const int row_bytes = w * 3;
            for (int i = threadIdx.x; i < rows * (WS_BYTES / 4); i += 256) {
                const int r = i / (WS_BYTES / 4), c = i - r * (WS_BYTES / 4);
                const int sy = sy_lo + r, gb = bx0 + 4 * c;
                uint32_t v = 0;
                const bool row_in = sy >= 0 && sy < h;
                if (BORDER == 0 || row_in) {
                    const uint8_t* row = src + (size_t)clampi(sy, 0, h - 1) * src_stride;
                    if (gb >= 0 && gb + 3 < row_bytes && ((((uintptr_t)(row + gb)) & 3) == 0)) {
                        v = *(const uint32_t*)(row + gb);
                    } else {
#pragma unroll
                        for (int k = 0; k < 4; k++) {
                            const int px = floor_div3(gb + k), ch = gb + k - 3 * px;
                            uint32_t b;
                            if (BORDER == 1) b = (px < 0 || px >= w) ? 0u : row[px * 3 + ch];
                            else b = row[clampi(px, 0, w - 1) * 3 + ch];
                            v |= b << (8 * k);
                        }
                    }
                }
                tile[r * WS_PITCH_DW + c] = v;
            }
        }
    }
    __syncthreads();

    const int row = threadIdx.x >> 4, xq = threadIdx.x & 15;
    const int y = y0 + row;
    if (y >= h) return;
    const int xb = x0 + 4 * xq;
    if (xb >= w) return;
    const float fy = (float)y;
    const float By = B * fy, A1y = A1 * fy;
    uint32_t o[12];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const float fx = (float)(xb + k);
        const float Wx = A1 * fx - By + TX;          // generators.cpp:141
        const float Wy = B * fx + A1y + TY;          // generators.cpp:142
        if (!fits) {
            warp_pixel_global<MODE, BORDER>(src, w, h, src_stride, Wx, Wy, &o[3 * k]);
            continue;
        }
        const float flx = floorf(Wx), fly = floorf(Wy);
        const int ix = (int)flx, iy = (int)fly;
        const float frx = Wx - flx, fry = Wy - fly;
        if (MODE == 0) {
            float wx[4], wy[4];
            lanczos_weights4(frx, wx);
            lanczos_weights4(fry, wy);
            const int lb = (ix - 1) * 3 - bx0;                    // byte offset of tap (rx=0) in the staged row
            const uint32_t* t = tile + (iy - 1 - sy_lo) * WS_PITCH_DW + (lb >> 2);
            const uint32_t sh = (uint32_t)(lb & 3);
            float nb = 0.f, ng = 0.f, nr = 0.f, den = 0.f;
#pragma unroll
            for (int ry = 0; ry < 4; ry++) {
                const uint32_t d0 = t[ry * WS_PITCH_DW], d1 = t[ry * WS_PITCH_DW + 1], d2 = t[ry * WS_PITCH_DW + 2],
                               d3 = t[ry * WS_PITCH_DW + 3];
                const uint32_t q0 = __builtin_amdgcn_alignbyte(d1, d0, sh), q1 = __builtin_amdgcn_alignbyte(d2, d1, sh),
                               q2 = __builtin_amdgcn_alignbyte(d3, d2, sh);
                // q0 q1 q2 = B0 G0 R0 B1 | G1 R1 B2 G2 | R2 B3 G3 R3
                const float vb[4] = {ub(q0, 0), ub(q0, 3), ub(q1, 2), ub(q2, 1)};
                const float vg[4] = {ub(q0, 1), ub(q1, 0), ub(q1, 3), ub(q2, 2)};
                const float vr[4] = {ub(q0, 2), ub(q1, 1), ub(q2, 0), ub(q2, 3)};
#pragma unroll
                for (int rx = 0; rx < 4; rx++) {
                    const float w2d = wx[rx] * wy[ry];
                    nb = nb + w2d * vb[rx];
                    ng = ng + w2d * vg[rx];
                    nr = nr + w2d * vr[rx];
                    den = den + w2d;
                }
            }
            o[3 * k] = store_u8(nb / den);
            o[3 * k + 1] = store_u8(ng / den);
            o[3 * k + 2] = store_u8(nr / den);
        } else {
            const int lb = ix * 3 - bx0;
            const uint32_t* t = tile + (iy - sy_lo) * WS_PITCH_DW + (lb >> 2);
            const uint32_t sh = (uint32_t)(lb & 3);
            uint32_t q[2][2];
#pragma unroll
            for (int ry = 0; ry < 2; ry++) {
                const uint32_t d0 = t[ry * WS_PITCH_DW], d1 = t[ry * WS_PITCH_DW + 1], d2 = t[ry * WS_PITCH_DW + 2];
                q[ry][0] = __builtin_amdgcn_alignbyte(d1, d0, sh);   // B0 G0 R0 B1
                q[ry][1] = __builtin_amdgcn_alignbyte(d2, d1, sh);   // G1 R1 .. ..
            }
#pragma unroll
            for (int c = 0; c < 3; c++) {
                const float a0 = ub(q[0][0], c), a1 = c == 0 ? ub(q[0][0], 3) : ub(q[0][1], c - 1);
                const float b0 = ub(q[1][0], c), b1 = c == 0 ? ub(q[1][0], 3) : ub(q[1][1], c - 1);
                o[3 * k + c] = store_u8(lerpf(lerpf(a0, a1, frx), lerpf(b0, b1, frx), fry));
            }
        }
    }
    uint8_t* op = dst + (size_t)y * dst_stride + (size_t)xb * 3;
    if (xb + 3 < w && ((((uintptr_t)op) & 3) == 0)) {
        uint32_t p0 = o[0] | (o[1] << 8) | (o[2] << 16) | (o[3] << 24);
        uint32_t p1 = o[4] | (o[5] << 8) | (o[6] << 16) | (o[7] << 24);
        uint32_t p2 = o[8] | (o[9] << 8) | (o[10] << 16) | (o[11] << 24);
        typedef uint32_t u32x3 __attribute__((ext_vector_type(3)));
        u32x3 pk = {p0, p1, p2};
        *(u32x3*)op = pk;
    } else {
        const int n = min(4, w - xb);
        for (int k = 0; k < 3 * n; k++) op[k] = (uint8_t)o[k];
    }
}

}  // namespace

namespace vsk {

hipError_t bgr_warp_u8c3(const uint8_t* src, int w, int h, int src_stride, const float4* params_dev, int mode, int border,
                         uint8_t* dst, int dst_stride, int n_frames, size_t src_fs, size_t dst_fs, hipStream_t s) {
    if (n_frames > 65535 || (h + WT_H - 1) / WT_H > 65535) return hipErrorNotSupported;
    dim3 grid((w + WT_W - 1) / WT_W, (h + WT_H - 1) / WT_H, n_frames), block(256);
#define VS_LAUNCH(M, Bd) \
    hipLaunchKernelGGL((vs_k_bgr_warp_u8c3<M, Bd>), grid, block, 0, s, src, w, h, src_stride, params_dev, dst, dst_stride, src_fs, dst_fs)
    if (mode == 0 && border == 0) VS_LAUNCH(0, 0);
    else if (mode == 0) VS_LAUNCH(0, 1);
    else if (border == 0) VS_LAUNCH(1, 0);
    else VS_LAUNCH(1, 1);
#undef VS_LAUNCH
    return hipGetLastError();
}

}  // namespace vsk
