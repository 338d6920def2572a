/*
 * vs_oracle.cpp -- CPU restatement of the reference's alignment + warp path.
 * TEST INFRASTRUCTURE ONLY (see vs_oracle.h).  "parity unpinned": the reference cannot be
 * built here (needs Halide/OpenCV/Eigen); pinned by known answers derived from its source.
 *
 * Build: g++ -O2 -ffp-contract=off -fno-fast-math (oracle/Makefile).  No FMA contraction,
 * IEEE float/double, evaluation order exactly as written in the cited reference lines.
 */
#include "vs_oracle.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <deque>
#include <vector>
#include <thread>

/* Row parallelism for the CPU-baseline timing (mode (i): "stage-internal row parallelism over all host cores",
 * the analogue of the .parallel(y) of the reference's Halide schedules).  Off by default (1 thread).  Only loops whose
 * iterations are independent use it -- no reduction is split, so results are bit-identical for every thread count
 * (sparse_ica's serial fp64 sums stay serial, as sparse_ica.schedule.h:59-66 has them). */
static int g_vso_threads = 1;
extern "C" void vso_set_threads(int n) { g_vso_threads = n < 1 ? 1 : (n > 256 ? 256 : n); }
extern "C" int vso_get_threads(void) { return g_vso_threads; }
template <typename F>
static void vso_parallel_rows(int n, int min_rows_per_thread, F&& body) {   /* body(begin, end) */
    int t = g_vso_threads;
    if (t > n / (min_rows_per_thread > 0 ? min_rows_per_thread : 1)) t = n / (min_rows_per_thread > 0 ? min_rows_per_thread : 1);
    if (t <= 1) { body(0, n); return; }
    std::vector<std::thread> th;
    th.reserve((size_t)t - 1);
    for (int k = 1; k < t; k++) th.emplace_back([&, k] { body((int)((long long)n * k / t), (int)((long long)n * (k + 1) / t)); });
    body(0, (int)((long long)n / t));
    for (auto& x : th) x.join();
}

namespace {

inline int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

/* generators.cpp:31-47 -- 6th-order even polynomial, Horner in x^2, fp32 */
inline float lanczos2(float x) {
    float x2 = x * x;
    float val = 0.000858519f;
    val = -0.0158853f + val * x2;
    val = 0.128693f + val * x2;
    val = -0.583468f + val * x2;
    val = 1.52229f + val * x2;
    val = -2.05238f + val * x2;
    val = 0.999861f + val * x2;
    return std::fabs(x) >= 2.0f ? 0.0f : val;
}

/* floor(position) -> integer pixel index.  The reference writes cast<int>(floor(W)) (generators.cpp:147-152, 466-470, 679-683), which is undefined
 * for positions outside the int range (a Gauss-Newton run that diverges gets there: |T| ~ 1e14) and so is the window arithmetic after it.  Defined
 * here the way the product defines it: the conversion saturates and maps NaN to 0 (v_cvt_i32_f32), and an index used for ADDRESSING is then pulled
 * into [-8, n + 8) -- 4 or more outside the image every tap of the window is the same border pixel (or outside, for the constant border) anyway.
 * Identical to the plain cast for every position inside the int range. */
inline int sat_int(float f) {
    if (!(f == f)) return 0;
    if (f >= 2147483648.0f) return 2147483647;
    if (f <= -2147483648.0f) return -2147483647 - 1;
    return (int)f;
}
inline int sample_index(float fl, int n) { return clampi(sat_int(fl), -8, n + 7); }

/* Pixel fetch policies.  stride/channels in elements. */
template <typename T>
struct ImageRef {
    const T* data; int w, h, stride, channels;
    /* BoundaryConditions::repeat_edge (generators.cpp:70,146,215,454,663) */
    inline float clamped(int x, int y, int c) const {
        x = clampi(x, 0, w - 1); y = clampi(y, 0, h - 1);
        return (float)data[(size_t)y * stride + (size_t)x * channels + c];
    }
    /* cv::BORDER_CONSTANT, value 0 (imgproc.cpp:479-480) */
    inline float constant0(int x, int y, int c) const {
        if (x < 0 || y < 0 || x >= w || y >= h) return 0.0f;
        return (float)data[(size_t)y * stride + (size_t)x * channels + c];
    }
};

/* The Lanczos2 sampler shared by sparse_ica (generators.cpp:459-498) and sparse_warpdiff
 * (:672-697): floor/frac, 5+5 weights, 25 taps rx inner / ry outer, separate num and den
 * accumulators starting from 0, one divide. */
template <typename T, bool CONSTANT_BORDER>
inline float lanczos_sample(const ImageRef<T>& img, float Wx, float Wy, int c) {
    float floorWx = std::floor(Wx), floorWy = std::floor(Wy);
    float fracWx = Wx - floorWx, fracWy = Wy - floorWy;
    float wx[5], wy[5];
    for (int u = 0; u < 5; u++) {
        wx[u] = lanczos2((float)(u - 2) - fracWx);
        wy[u] = lanczos2((float)(u - 2) - fracWy);
    }
    int ix = sample_index(floorWx, img.w), iy = sample_index(floorWy, img.h);
    float sum_num = 0.0f, sum_den = 0.0f;
    for (int ry = 0; ry < 5; ry++) {
        for (int rx = 0; rx < 5; rx++) {
            float w2d = wx[rx] * wy[ry];
            float val = CONSTANT_BORDER ? img.constant0(ix + rx - 2, iy + ry - 2, c)
                                        : img.clamped(ix + rx - 2, iy + ry - 2, c);
            sum_num = sum_num + w2d * val;
            sum_den = sum_den + w2d;
        }
    }
    return sum_num / sum_den;
}

/* VSO_WARP_LANCZOS2_CONTRACTED -- the same sampler as the reference's own build target would round it.  The reference
 * compiles its generators for "x86-64-...-fma-..." (CMakeLists.txt:151) and never asks for strict_float (no occurrence in
 * the tree), so LLVM is free to contract a multiply into the add that consumes it (Halide emits its float arithmetic with
 * the contract flag unless strict_float is requested; Halide's source is not in this container, so this is stated from its
 * published behaviour, not from a file here).  Applied to the expression trees as written:
 *   lanczos2 (generators.cpp:38-44)   "c + val * x2"  ->  fma(val, x2, c), six times; x2 = x * x and the select unchanged
 *   sum_num  (generators.cpp:694)     "+= w_2d * val" ->  fma(w_2d, val, sum_num), w_2d = wx * wy a rounded product (:687)
 *   sum_den  (generators.cpp:695)     "+= w_2d"       ->  sum_den + w_2d (no multiply of its own to fuse)
 *   sum_num / sum_den                 one IEEE divide (:697)
 * Same tap order (rxy.x inner, rxy.y outer), same accumulators from 0, same sampling position as the un-contracted form:
 * the two modes differ only in where the sampler rounds.  std::fmaf is exact-then-round-once by definition, so this
 * function means the same on every machine. */
inline float lanczos2_contracted(float x) {
    float x2 = x * x;
    float val = 0.000858519f;
    val = std::fmaf(val, x2, -0.0158853f);
    val = std::fmaf(val, x2, 0.128693f);
    val = std::fmaf(val, x2, -0.583468f);
    val = std::fmaf(val, x2, 1.52229f);
    val = std::fmaf(val, x2, -2.05238f);
    val = std::fmaf(val, x2, 0.999861f);
    return std::fabs(x) >= 2.0f ? 0.0f : val;
}
template <typename T, bool CONSTANT_BORDER>
inline float lanczos_sample_contracted(const ImageRef<T>& img, float Wx, float Wy, int c) {
    float floorWx = std::floor(Wx), floorWy = std::floor(Wy);
    float fracWx = Wx - floorWx, fracWy = Wy - floorWy;
    float wx[5], wy[5];
    for (int u = 0; u < 5; u++) {
        wx[u] = lanczos2_contracted((float)(u - 2) - fracWx);
        wy[u] = lanczos2_contracted((float)(u - 2) - fracWy);
    }
    int ix = sample_index(floorWx, img.w), iy = sample_index(floorWy, img.h);
    float sum_num = 0.0f, sum_den = 0.0f;
    for (int ry = 0; ry < 5; ry++) {
        for (int rx = 0; rx < 5; rx++) {
            float w2d = wx[rx] * wy[ry];
            float val = CONSTANT_BORDER ? img.constant0(ix + rx - 2, iy + ry - 2, c)
                                        : img.clamped(ix + rx - 2, iy + ry - 2, c);
            sum_num = std::fmaf(w2d, val, sum_num);
            sum_den = sum_den + w2d;
        }
    }
    return sum_num / sum_den;
}

/* VSO_WARP_LANCZOS2_SEPARABLE -- the third member of the sampler family (twin of the product's VS_WARP_LANCZOS2_SEP): the same
 * weights as the contracted form (lanczos2_contracted, generators.cpp:38-46), the same sampling position and the same 4 x 4 live
 * taps, but the window sum is taken rows first, then columns, and the denominator as the product of the two 1-D weight sums --
 * sum_{ry,rx} wx*wy*v = sum_ry wy * (sum_rx wx * v) and sum_{ry,rx} wx*wy = (sum wx)(sum wy) in real arithmetic: a reassociation
 * of generators.cpp:687-697, inside the slack a non-strict_float Halide build has (CMakeLists.txt:151), admitted to benchmarks only
 * through SURVEY 8(d)'s integer gate against the UN-contracted order (tests/test_warp_gate_*.py).  Written out:
 *   h[ry]  = fma(wx4, v4, fma(wx3, v3, fma(wx2, v2, wx1 * v1)))          (taps 1..4 of the 5-tap window; tap 0 weighs exactly 0)
 *   num    = fma(wy4, h4, fma(wy3, h3, fma(wy2, h2, wy1 * h1)))
 *   den    = ((wx1 + wx2) + (wx3 + wx4)) * ((wy1 + wy2) + (wy3 + wy4))
 *   result = num * RN(1 / den)                                             (one correctly rounded reciprocal for all channels)
 * std::fmaf is exact-then-round-once and 1.0f / den is IEEE: the function means the same on every machine. */
template <typename T, bool CONSTANT_BORDER>
inline float lanczos_sample_separable(const ImageRef<T>& img, float Wx, float Wy, int c) {
    float floorWx = std::floor(Wx), floorWy = std::floor(Wy);
    float fracWx = Wx - floorWx, fracWy = Wy - floorWy;
    float wx[4], wy[4];
    for (int u = 0; u < 4; u++) {
        wx[u] = lanczos2_contracted((float)(u - 1) - fracWx);
        wy[u] = lanczos2_contracted((float)(u - 1) - fracWy);
    }
    int ix = sample_index(floorWx, img.w), iy = sample_index(floorWy, img.h);
    float hrow[4];
    for (int ry = 0; ry < 4; ry++) {
        float v[4];
        for (int rx = 0; rx < 4; rx++)
            v[rx] = CONSTANT_BORDER ? img.constant0(ix + rx - 1, iy + ry - 1, c) : img.clamped(ix + rx - 1, iy + ry - 1, c);
        hrow[ry] = std::fmaf(wx[3], v[3], std::fmaf(wx[2], v[2], std::fmaf(wx[1], v[1], wx[0] * v[0])));
    }
    float num = std::fmaf(wy[3], hrow[3], std::fmaf(wy[2], hrow[2], std::fmaf(wy[1], hrow[1], wy[0] * hrow[0])));
    float den = ((wx[0] + wx[1]) + (wx[2] + wx[3])) * ((wy[0] + wy[1]) + (wy[2] + wy[3]));
    float r = 1.0f / den;
    return num * r;
}

/* image_warp's bilinear sampler (generators.cpp:148-163).  Halide float lerp(a,b,t) is
 * a*(1-t) + b*t. */
inline float lerpf(float a, float b, float t) { return a * (1.0f - t) + b * t; }

template <typename T, bool CONSTANT_BORDER>
inline float bilinear_sample(const ImageRef<T>& img, float Wx, float Wy, int c) {
    /* generators.cpp:150-153 takes the fraction from the converted index, W - float(int(floor(W))): equal to W - floor(W) for every position
     * inside the int range and undefined outside it; W - floor(W) is defined everywhere, and it is what every kernel of the product uses */
    float flx = std::floor(Wx), fly = std::floor(Wy);
    float wx = Wx - flx, wy = Wy - fly;
    int fx = sample_index(flx, img.w), fy = sample_index(fly, img.h);
    float x0y0, x1y0, x0y1, x1y1;
    if (CONSTANT_BORDER) {
        x0y0 = img.constant0(fx, fy, c); x1y0 = img.constant0(fx + 1, fy, c);
        x0y1 = img.constant0(fx, fy + 1, c); x1y1 = img.constant0(fx + 1, fy + 1, c);
    } else {
        x0y0 = img.clamped(fx, fy, c); x1y0 = img.clamped(fx + 1, fy, c);
        x0y1 = img.clamped(fx, fy + 1, c); x1y1 = img.clamped(fx + 1, fy + 1, c);
    }
    float top = lerpf(x0y0, x1y0, wx);
    float bottom = lerpf(x0y1, x1y1, wx);
    return lerpf(top, bottom, wy);
}

/* ---- VSO_WARP_BILINEAR_CV: cv::warpAffine(INTER_LINEAR) as the reference's stabilizer calls it ------------------------------------
 * imgproc.cpp:446-484 warpBySimilarityTransform hands cv::warpAffine the FORWARD matrix of the transform WITHOUT WARP_INVERSE_MAP, INTER_LINEAR,
 * BORDER_CONSTANT, value 0.  OpenCV is not in this image and the reference does not pin its version (README.md:22: apt libopencv-dev), so what
 * follows restates the PUBLISHED algorithm of OpenCV 4.x's classic path (modules/imgproc/src/imgwarp.cpp: cv::warpAffine, WarpAffineInvoker,
 * remapBilinear; 4.5.4 / 4.6.0 are what Ubuntu 22.04 / 24.04 ship) -- "parity unpinned (OpenCV version)" like every other OpenCV stand-in:
 *   1. the 2x3 matrix is inverted in double precision, in OpenCV's own operation order (D = M0 M4 - M1 M3, ...);
 *   2. source coordinates are FIXED POINT: AB_BITS = 10, adelta[x] = cvRound(M0 x 1024), bdelta[x] = cvRound(M3 x 1024), per row
 *      X0 = cvRound((M1 y + M2) 1024) + 16, Y0 likewise (round_delta = 1024 / 32 / 2), X = (X0 + adelta[x]) >> 5: 1/32-pixel positions,
 *      sx = saturate_cast<short>(X >> 5), fx = X & 31 (INTER_BITS = 5);
 *   3. 8-bit samples: four integer weights (32 - fx)(32 - fy) 32, ... (BilinearTab_i: the float products of 1/32 fractions times 2^15 are
 *      exact integers, they sum to 2^15 without the table's fix-up), result = (sum w v + 2^14) >> 15 (FixedPtCast<int, uchar, 15>);
 *      16-bit containers: float weights w = a b / 1024 (exact), result = saturate_cast<ushort>(v00 w00 + v01 w01 + v10 w10 + v11 w11),
 *      the products and sums in float, left to right, cvRound (remapBilinear<Cast<float, ushort>, float>; exact for samples below 2^14);
 *   4. taps outside the frame read the border value 0 (BORDER_CONSTANT) or the nearest edge pixel (VSO_BORDER_CLAMP = BORDER_REPLICATE).
 * The transform is the one handed to warpBySimilarityTransform -- the FORWARD map; every other mode of this file takes the sampling map.
 * cvRound is round-half-to-even (lrint under the default rounding mode); a coordinate outside the int range saturates here (x86's cvtsd2si
 * returns INT_MIN there -- no frame of any size gets near it).  The integer stages are exact: the product is held to this bit for bit. */
inline int cv_round_sat(double v) {
    if (!(v == v)) return 0;
    if (v >= 2147483647.0) return 2147483647;
    if (v <= -2147483648.0) return -2147483647 - 1;
    return (int)std::nearbyint(v);
}
/* imgproc.cpp:457-466 + cv::warpAffine's inversion; Minv maps an output pixel to its source position */
void cv_inverse_matrix(const vso_transform* t, int w, int h, double M[6]) {
    double cx = (w - 1) * 0.5, cy = (h - 1) * 0.5;
    double tx_ul = t->TX - t->A * cx + t->B * cy;
    double ty_ul = t->TY - t->B * cx - t->A * cy;
    M[0] = 1.0 + t->A; M[1] = -t->B; M[2] = tx_ul;
    M[3] = t->B;       M[4] = 1.0 + t->A; M[5] = ty_ul;
    double D = M[0] * M[4] - M[1] * M[3];
    D = D != 0 ? 1. / D : 0;
    double A11 = M[4] * D, A22 = M[0] * D;
    M[0] = A11; M[1] *= -D;
    M[3] *= -D; M[4] = A22;
    double b1 = -M[0] * M[2] - M[1] * M[5];
    double b2 = -M[3] * M[2] - M[4] * M[5];
    M[2] = b1; M[5] = b2;
}
template <typename T>
void cv_warp_impl(const T* src, int w, int h, int src_stride, int channels, const vso_transform* t, int border, int max_value,
                  T* dst, int dst_stride) {
    double M[6];
    cv_inverse_matrix(t, w, h, M);
    const int AB_BITS = 10, INTER_BITS = 5, AB_SCALE = 1 << AB_BITS, TAB = 1 << INTER_BITS, round_delta = AB_SCALE / TAB / 2;
    std::vector<int> adelta((size_t)w), bdelta((size_t)w);
    for (int x = 0; x < w; x++) {
        adelta[(size_t)x] = cv_round_sat(M[0] * x * AB_SCALE);
        bdelta[(size_t)x] = cv_round_sat(M[3] * x * AB_SCALE);
    }
    auto tap = [&](int xx, int yy, int c) -> int {
        if (border == VSO_BORDER_CONSTANT) {
            if (xx < 0 || yy < 0 || xx >= w || yy >= h) return 0;
        } else { xx = clampi(xx, 0, w - 1); yy = clampi(yy, 0, h - 1); }
        return (int)src[(size_t)yy * src_stride + (size_t)xx * channels + c];
    };
    vso_parallel_rows(h, 8, [&](int y_begin, int y_end) {
    for (int y = y_begin; y < y_end; y++) {
        const int X0 = (int)((unsigned)cv_round_sat((M[1] * y + M[2]) * AB_SCALE) + (unsigned)round_delta);
        const int Y0 = (int)((unsigned)cv_round_sat((M[4] * y + M[5]) * AB_SCALE) + (unsigned)round_delta);
        for (int x = 0; x < w; x++) {
            const int X = (int)((unsigned)X0 + (unsigned)adelta[(size_t)x]) >> (AB_BITS - INTER_BITS);
            const int Y = (int)((unsigned)Y0 + (unsigned)bdelta[(size_t)x]) >> (AB_BITS - INTER_BITS);
            const int sx = clampi(X >> INTER_BITS, -32768, 32767), sy = clampi(Y >> INTER_BITS, -32768, 32767);   /* saturate_cast<short> */
            const int fx = X & (TAB - 1), fy = Y & (TAB - 1);
            const int a0 = TAB - fx, a1 = fx, b0 = TAB - fy, b1 = fy;
            for (int c = 0; c < channels; c++) {
                const int v00 = tap(sx, sy, c), v01 = tap(sx + 1, sy, c), v10 = tap(sx, sy + 1, c), v11 = tap(sx + 1, sy + 1, c);
                int r;
                if (sizeof(T) == 1) {
                    const int acc = v00 * (a0 * b0 * 32) + v01 * (a1 * b0 * 32) + v10 * (a0 * b1 * 32) + v11 * (a1 * b1 * 32);
                    r = (acc + (1 << 14)) >> 15;
                } else {
                    const float w00 = (float)(a0 * b0) * (1.0f / 1024.0f), w01 = (float)(a1 * b0) * (1.0f / 1024.0f);
                    const float w10 = (float)(a0 * b1) * (1.0f / 1024.0f), w11 = (float)(a1 * b1) * (1.0f / 1024.0f);
                    const float sum = (float)v00 * w00 + (float)v01 * w01 + (float)v10 * w10 + (float)v11 * w11;
                    r = (int)std::nearbyint(sum);
                }
                r = r < 0 ? 0 : (r > max_value ? max_value : r);
                dst[(size_t)y * dst_stride + (size_t)x * channels + c] = (T)r;
            }
        }
    }
    });
}

template <typename T>
void bgr_warp_impl(const T* src, int w, int h, int src_stride, int channels,
                   const vso_transform* t, int mode, int border, int max_value,
                   T* dst_int, float* dst_f32, int dst_stride) {
    float p[4];
    vso_ul_params_warp(t, w, h, p);
    const float A = p[0], B = p[1], TX = p[2], TY = p[3];
    ImageRef<T> img{src, w, h, src_stride, channels};
    vso_parallel_rows(h, 8, [&](int y_begin, int y_end) {
    for (int y = y_begin; y < y_end; y++) {
        for (int x = 0; x < w; x++) {
            /* generators.cpp:141-142 */
            float Wx = (1.0f + A) * (float)x - B * (float)y + TX;
            float Wy = B * (float)x + (1.0f + A) * (float)y + TY;
            for (int c = 0; c < channels; c++) {
                float v;
                if (mode == VSO_WARP_LANCZOS2)
                    v = border == VSO_BORDER_CONSTANT ? lanczos_sample<T, true>(img, Wx, Wy, c)
                                                      : lanczos_sample<T, false>(img, Wx, Wy, c);
                else if (mode == VSO_WARP_LANCZOS2_CONTRACTED)
                    v = border == VSO_BORDER_CONSTANT ? lanczos_sample_contracted<T, true>(img, Wx, Wy, c)
                                                      : lanczos_sample_contracted<T, false>(img, Wx, Wy, c);
                else if (mode == VSO_WARP_LANCZOS2_SEPARABLE)
                    v = border == VSO_BORDER_CONSTANT ? lanczos_sample_separable<T, true>(img, Wx, Wy, c)
                                                      : lanczos_sample_separable<T, false>(img, Wx, Wy, c);
                else
                    v = border == VSO_BORDER_CONSTANT ? bilinear_sample<T, true>(img, Wx, Wy, c)
                                                      : bilinear_sample<T, false>(img, Wx, Wy, c);
                size_t o = (size_t)y * dst_stride + (size_t)x * channels + c;
                if (dst_f32) {
                    dst_f32[o] = v;
                } else {
                    /* build rule: round half up, saturate */
                    float r = std::floor(v + 0.5f);
                    r = r < 0.0f ? 0.0f : (r > (float)max_value ? (float)max_value : r);
                    dst_int[o] = (T)r;
                }
            }
        }
    }
    });
}

/* ---- 4x4 symmetric eigen-solver (stands in for cv::SVD / Mat::inv(DECOMP_SVD)) ---------- */
/* Cyclic Jacobi, fixed sweep order (p,q) = (0,1),(0,2),(0,3),(1,2),(1,3),(2,3). */
void jacobi_eig4(const double Hin[16], double eval[4], double V[16]) {
    double a[4][4];
    for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) { a[i][j] = Hin[i * 4 + j]; V[i * 4 + j] = (i == j) ? 1.0 : 0.0; }
    for (int sweep = 0; sweep < 32; sweep++) {
        double off = 0.0, diag = 0.0;
        for (int i = 0; i < 4; i++) { diag += a[i][i] * a[i][i]; for (int j = i + 1; j < 4; j++) off += a[i][j] * a[i][j]; }
        if (off <= 1e-300 || off <= 1e-34 * diag) break;
        for (int p = 0; p < 3; p++) {
            for (int q = p + 1; q < 4; q++) {
                double apq = a[p][q];
                if (apq == 0.0) continue;
                double theta = (a[q][q] - a[p][p]) / (2.0 * apq);
                double tt = (theta >= 0.0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
                double c = 1.0 / std::sqrt(tt * tt + 1.0), s = tt * c;
                for (int k = 0; k < 4; k++) {           /* columns p,q of a */
                    double akp = a[k][p], akq = a[k][q];
                    a[k][p] = c * akp - s * akq; a[k][q] = s * akp + c * akq;
                }
                for (int k = 0; k < 4; k++) {           /* rows p,q of a */
                    double apk = a[p][k], aqk = a[q][k];
                    a[p][k] = c * apk - s * aqk; a[q][k] = s * apk + c * aqk;
                }
                for (int k = 0; k < 4; k++) {
                    double vkp = V[k * 4 + p], vkq = V[k * 4 + q];
                    V[k * 4 + p] = c * vkp - s * vkq; V[k * 4 + q] = s * vkp + c * vkq;
                }
            }
        }
    }
    for (int i = 0; i < 4; i++) eval[i] = a[i][i];
}

}  // namespace

extern "C" {

void vso_aligner_params_default(vso_aligner_params* p) {
    p->phase_correlate = 0; p->phase_correlate_threshold = 0.5; p->threshold = 0.02;
    p->smallest_fraction = 0.8f; p->max_iters = 64; p->pyramid_min_width = 20; p->pyramid_min_height = 20;
    p->max_displacement = 10.0;
}
void vso_stabilizer_params_default(vso_stabilizer_params* p) {
    vso_aligner_params_default(&p->aligner);
    p->lag = 10; p->smoother_memory = 5; p->lambda = 4.0; p->enable_smoother = 1; p->crop_pixels = 32;
    p->min_disp = 48.0; p->max_disp = 64.0; p->min_decay = 0.9; p->max_decay = 0.7;
    p->warp_mode = VSO_WARP_BILINEAR_CV; p->warp_border = VSO_BORDER_CONSTANT;   /* imgproc.cpp:472-481: cv::warpAffine INTER_LINEAR, BORDER_CONSTANT */
}

float vso_lanczos2(float x) { return lanczos2(x); }

/* generators.cpp:66-91.  fp32 separable blur, y then x, every intermediate is exact in fp32
 * (SURVEY a2), sample at (2x,2y), truncating cast. */
void vso_pyr_down(const uint8_t* in, int w, int h, int in_stride, uint8_t* out, int ow, int oh, int out_stride) {
    const float coeffs[5] = {1.0f / 16, 4.0f / 16, 6.0f / 16, 4.0f / 16, 1.0f / 16};
    auto px = [&](int x, int y) { return (float)in[(size_t)clampi(y, 0, h - 1) * in_stride + clampi(x, 0, w - 1)]; };
    auto blur_y = [&](int x, int y) {
        return (coeffs[0] * px(x, y - 2) + coeffs[1] * px(x, y - 1) + coeffs[2] * px(x, y) +
                coeffs[3] * px(x, y + 1) + coeffs[4] * px(x, y + 2));
    };
    vso_parallel_rows(oh, 8, [&](int y_begin, int y_end) {
    for (int y = y_begin; y < y_end; y++)
        for (int x = 0; x < ow; x++) {
            int X = 2 * x, Y = 2 * y;
            float v = (coeffs[0] * blur_y(X - 2, Y) + coeffs[1] * blur_y(X - 1, Y) + coeffs[2] * blur_y(X, Y) +
                       coeffs[3] * blur_y(X + 1, Y) + coeffs[4] * blur_y(X + 2, Y));
            out[(size_t)y * out_stride + x] = (uint8_t)v;
        }
    });
}

/* generators.cpp:215-223 */
void vso_grad_xy(const uint8_t* in, int w, int h, int stride, float* gx, float* gy) {
    auto px = [&](int x, int y) { return (float)in[(size_t)clampi(y, 0, h - 1) * stride + clampi(x, 0, w - 1)]; };
    vso_parallel_rows(h, 16, [&](int y_begin, int y_end) {
    for (int y = y_begin; y < y_end; y++)
        for (int x = 0; x < w; x++) {
            gx[(size_t)y * w + x] = 0.5f * (px(x + 1, y) - px(x - 1, y));
            gy[(size_t)y * w + x] = 0.5f * (px(x, y + 1) - px(x, y - 1));
        }
    });
}

/* imgproc.cpp:151-162 */
int vso_tile_size(int w, int h) {
    const int min_tiles = 1000, max_tile_size = 20;
    int tile_size = 2;
    for (int i = 4; i <= max_tile_size; i += 2) {
        int tx = w / i, ty = h / i;
        if (tx * ty < min_tiles) break;
        tile_size = i;
    }
    return tile_size;
}

/* generators.cpp:275-293.  Halide::argmax over RDom(0..ts,0..ts) with no schedule applied
 * (:321): serial scan r.x inner / r.y outer, update only on strict '>' => ties go to the
 * smallest r.y then smallest r.x; all-zero tile => (0,0). */
void vso_grad_argmax(const float* gx, const float* gy, int w, int h, int ts, uint16_t* lmx, uint16_t* lmy) {
    int tx = w / ts, ty = h / ts;
    for (int pass = 0; pass < 2; pass++) {
        const float* g = pass == 0 ? gx : gy;
        uint16_t* lm = pass == 0 ? lmx : lmy;
        vso_parallel_rows(ty, 2, [&](int y_begin, int y_end) {
        for (int y = y_begin; y < y_end; y++)
            for (int x = 0; x < tx; x++) {
                int bx = 0, by = 0;
                float best = std::fabs(g[(size_t)(y * ts) * w + x * ts]);
                for (int ry = 0; ry < ts; ry++)
                    for (int rx = 0; rx < ts; rx++) {
                        float v = std::fabs(g[(size_t)(y * ts + ry) * w + (x * ts + rx)]);
                        if (v > best) { best = v; bx = rx; by = ry; }
                    }
                lm[(size_t)0 * tx * ty + y * tx + x] = (uint16_t)(bx + x * ts);
                lm[(size_t)1 * tx * ty + y * tx + x] = (uint16_t)(by + y * ts);
            }
        });
    }
}

/* generators.cpp:346-385 */
void vso_sparse_jac(const float* gx, const float* gy, int w, int h, const uint16_t* lmx, const uint16_t* lmy,
                    int tx, int ty, float* jx, float* jy) {
    const size_t n = (size_t)tx * ty;
    float cx = (float)w * 0.5f, cy = (float)h * 0.5f;
    float scale = 1.f / (float)w;
    for (size_t i = 0; i < n; i++) {
        int ix0 = std::min<int>(lmx[i], w - 1), iy0 = std::min<int>(lmx[n + i], h - 1);
        int ix1 = std::min<int>(lmy[i], w - 1), iy1 = std::min<int>(lmy[n + i], h - 1);
        float u0 = (float)ix0 - cx, v0 = (float)iy0 - cy;
        float u1 = (float)ix1 - cx, v1 = (float)iy1 - cy;
        float g0 = gx[(size_t)iy0 * w + ix0], g1 = gy[(size_t)iy1 * w + ix1];
        jx[0 * n + i] = 2.f * g0 * u0 * scale;
        jx[1 * n + i] = 2.f * g0 * (-v0) * scale;
        jx[2 * n + i] = 2.f * g0;
        jx[3 * n + i] = 0.f;
        jy[0 * n + i] = 2.f * g1 * v1 * scale;
        jy[1 * n + i] = 2.f * g1 * u1 * scale;
        jy[2 * n + i] = 0.f;
        jy[3 * n + i] = 2.f * g1;
    }
}

/* generators.cpp:660-700 */
void vso_sparse_warpdiff(const uint8_t* tmpl, const uint8_t* key, int w, int h, int stride, const uint16_t* lm,
                         int tx, int ty, float A, float B, float TX, float TY, uint16_t* out) {
    const size_t n = (size_t)tx * ty;
    ImageRef<uint8_t> keyimg{key, w, h, stride, 1};
    vso_parallel_rows((int)n, 256, [&](int i_begin, int i_end) {
    for (size_t i = (size_t)i_begin; i < (size_t)i_end; i++) {
        int tile_x = std::min<int>(lm[i], w - 1), tile_y = std::min<int>(lm[n + i], h - 1);
        float orig_x = (float)tile_x, orig_y = (float)tile_y;
        float Wx = (1.0f + A) * orig_x - B * orig_y + TX;
        float Wy = B * orig_x + (1.0f + A) * orig_y + TY;
        float interpolated = lanczos_sample<uint8_t, false>(keyimg, Wx, Wy, 0);
        float diff = std::fabs(interpolated - (float)tmpl[(size_t)tile_y * stride + tile_x]);
        diff = diff < 0.0f ? 0.0f : (diff > 65535.0f ? 65535.0f : diff);
        out[i] = (uint16_t)diff;
    }
    });
}

/* generators.cpp:451-596.  reduce_4_x then reduce_4_y, each serial over r in index order
 * (sparse_ica.schedule.h:59-66,121-128); fp32 product widened to fp64 (:563-566). */
void vso_sparse_ica(const uint8_t* tmpl, const uint8_t* key, int w, int h, int stride, const uint16_t* selx, int nx,
                    const uint16_t* sely, int ny, const float* jacx, const float* jacy, float A, float B, float TX,
                    float TY, double out[4]) {
    ImageRef<uint8_t> keyimg{key, w, h, stride, 1};
    double acc[2][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
    for (int set = 0; set < 2; set++) {
        const uint16_t* sel = set == 0 ? selx : sely;
        const float* jac = set == 0 ? jacx : jacy;
        const int n = set == 0 ? nx : ny;
        for (int r = 0; r < n; r++) {
            float orig_x = (float)sel[r], orig_y = (float)sel[(size_t)n + r];
            float Wx = (1.0f + A) * orig_x - B * orig_y + TX;
            float Wy = B * orig_x + (1.0f + A) * orig_y + TY;
            float warped_val = lanczos_sample<uint8_t, false>(keyimg, Wx, Wy, 0);
            int tmpl_x = std::min<int>(sel[r], w - 1), tmpl_y = std::min<int>(sel[(size_t)n + r], h - 1);
            float template_val = (float)tmpl[(size_t)tmpl_y * stride + tmpl_x];
            float residual = template_val - warped_val;
            for (int c = 0; c < 4; c++) acc[set][c] += (double)(jac[(size_t)c * n + r] * residual);
        }
    }
    for (int c = 0; c < 4; c++) out[c] = (acc[0][c] + acc[1][c]) * 0.5f;
}

/* generators.cpp:139-164 */
void vso_image_warp(const uint8_t* in, int w, int h, int stride, float A, float B, float TX, float TY, float* out,
                    int ow, int oh) {
    ImageRef<uint8_t> img{in, w, h, stride, 1};
    for (int y = 0; y < oh; y++)
        for (int x = 0; x < ow; x++) {
            float Wx = (1.0f + A) * (float)x - B * (float)y + TX;
            float Wy = B * (float)x + (1.0f + A) * (float)y + TY;
            out[(size_t)y * ow + x] = bilinear_sample<uint8_t, false>(img, Wx, Wy, 0);
        }
}

/* imgproc.cpp:69-75 / 98-103: width()*0.5f is a float, promoted to double in the expression */
void vso_ul_params_sparse(const vso_transform* t, int w, int h, float out4[4]) {
    out4[0] = (float)t->A;
    out4[1] = (float)t->B;
    out4[2] = (float)(t->TX - t->A * (w * 0.5f) + t->B * (h * 0.5f));
    out4[3] = (float)(t->TY - t->B * (w * 0.5f) - t->A * (h * 0.5f));
}
/* imgproc.cpp:125-131: doubles, converted to float at the call */
void vso_ul_params_warp(const vso_transform* t, int w, int h, float out4[4]) {
    double cx = (w - 1) * 0.5, cy = (h - 1) * 0.5;
    double tx_ul = t->TX - t->A * cx + t->B * cy;
    double ty_ul = t->TY - t->B * cx - t->A * cy;
    out4[0] = (float)t->A; out4[1] = (float)t->B; out4[2] = (float)tx_ul; out4[3] = (float)ty_ul;
}

void vso_cv_inverse_matrix(const vso_transform* t, int w, int h, double M[6]) { cv_inverse_matrix(t, w, h, M); }

void vso_bgr_image_warp(const void* src, int w, int h, int src_stride, int channels, int bits, const vso_transform* t,
                        int mode, int border, int max_value, void* dst, int dst_stride) {
    if (mode == VSO_WARP_BILINEAR_CV) {                     /* t is the FORWARD transform (see cv_warp_impl) */
        if (bits == 8) cv_warp_impl<uint8_t>((const uint8_t*)src, w, h, src_stride, channels, t, border, max_value, (uint8_t*)dst, dst_stride);
        else cv_warp_impl<uint16_t>((const uint16_t*)src, w, h, src_stride, channels, t, border, max_value, (uint16_t*)dst, dst_stride);
        return;
    }
    if (bits == 8)
        bgr_warp_impl<uint8_t>((const uint8_t*)src, w, h, src_stride, channels, t, mode, border, max_value,
                               (uint8_t*)dst, nullptr, dst_stride);
    else
        bgr_warp_impl<uint16_t>((const uint16_t*)src, w, h, src_stride, channels, t, mode, border, max_value,
                                (uint16_t*)dst, nullptr, dst_stride);
}
void vso_bgr_image_warp_f32(const void* src, int w, int h, int src_stride, int channels, int bits,
                            const vso_transform* t, int mode, int border, float* dst, int dst_stride) {
    if (bits == 8)
        bgr_warp_impl<uint8_t>((const uint8_t*)src, w, h, src_stride, channels, t, mode, border, 0, nullptr, dst,
                               dst_stride);
    else
        bgr_warp_impl<uint16_t>((const uint16_t*)src, w, h, src_stride, channels, t, mode, border, 0, nullptr, dst,
                                dst_stride);
}

/* stands in for cv::cvtColor(BGR2GRAY) (alignment.cpp:212): OpenCV 4.x 15-bit fixed point */
int vso_format_bits(int format) {
    switch (format) {
        case VSO_FMT_GRAY8: case VSO_FMT_BGR8: return 8;
        case VSO_FMT_BGR10: return 10;
        case VSO_FMT_BGR12: return 12;
        case VSO_FMT_BGR16_FULL: return 16;
        default: return 0;
    }
}

void vso_bgr_to_gray(const void* src, int w, int h, int src_stride, int bits, int shift_to_8, uint8_t* dst,
                     int dst_stride) {
    vso_parallel_rows(h, 16, [&](int y_begin, int y_end) {
    for (int y = y_begin; y < y_end; y++)
        for (int x = 0; x < w; x++) {
            uint32_t b, g, r;
            if (bits == 8) {
                const uint8_t* p = (const uint8_t*)src + (size_t)y * src_stride + (size_t)x * 3;
                b = p[0]; g = p[1]; r = p[2];
            } else {
                const uint16_t* p = (const uint16_t*)src + (size_t)y * src_stride + (size_t)x * 3;
                b = p[0]; g = p[1]; r = p[2];
            }
            uint32_t v = (b * 3735u + g * 19235u + r * 9798u + 16384u) >> 15;
            v >>= shift_to_8;
            dst[(size_t)y * dst_stride + x] = (uint8_t)(v > 255u ? 255u : v);
        }
    });
}

/* imgproc.cpp:333-359 */
vso_transform vso_transform_inverse(const vso_transform* t) {
    double p = 1.0 + t->A, q = t->B;
    double denom = p * p + q * q;
    vso_transform r;
    r.A = (p / denom) - 1.0;
    r.B = -q / denom;
    r.TX = (-p * t->TX - q * t->TY) / denom;
    r.TY = (q * t->TX - p * t->TY) / denom;
    return r;
}
/* imgproc.cpp:361-387: T3(p) = t2(t1(p)) */
vso_transform vso_transform_compose(const vso_transform* t1, const vso_transform* t2) {
    double p1 = 1.0 + t1->A, q1 = t1->B, p2 = 1.0 + t2->A, q2 = t2->B;
    vso_transform r;
    r.A = (p2 * p1 - q2 * q1) - 1.0;
    r.B = (p2 * q1 + q2 * p1);
    r.TX = p2 * t1->TX - q2 * t1->TY + t2->TX;
    r.TY = q2 * t1->TX + p2 * t1->TY + t2->TY;
    return r;
}
/* imgproc.cpp:389-394 */
vso_point vso_transform_warp(const vso_transform* t, vso_point p) {
    vso_point W;
    W.x = (1 + t->A) * p.x - t->B * p.y + t->TX;
    W.y = t->B * p.x + (1 + t->A) * p.y + t->TY;
    return W;
}
/* imgproc.cpp:401-411 */
vso_point vso_transform_warp_center(const vso_transform* t, vso_point p, double cx, double cy) {
    double px = p.x - cx, py = p.y - cy;
    vso_point W;
    W.x = (1 + t->A) * px - t->B * py + cx + t->TX;
    W.y = t->B * px + (1 + t->A) * py + cy + t->TY;
    return W;
}
static double pt_dist(vso_point a, vso_point b) {
    double dx = a.x - b.x, dy = a.y - b.y;
    return std::sqrt(dx * dx + dy * dy);
}
/* imgproc.cpp:419-437 */
double vso_transform_max_corner_displacement(const vso_transform* t, double width, double height) {
    double cx = width * 0.5, cy = height * 0.5;
    vso_point c[4] = {{0.0, 0.0}, {width, 0.0}, {0.0, height}, {width, height}};
    double max_d = 0.0;
    for (int i = 0; i < 4; i++) max_d = std::max(max_d, pt_dist(vso_transform_warp_center(t, c[i], cx, cy), c[i]));
    return max_d;
}

/* alignment.cpp:438-486 with alignment.hpp:84-87's DeltaPixel */
int vso_select_smallest(const uint16_t* warpdiff, int tx, int ty, float fraction, int32_t* out_idx) {
    struct DeltaPixel { uint16_t abs_delta, tile_x, tile_y; };
    std::vector<DeltaPixel> v;
    v.reserve((size_t)tx * ty);
    for (int j = 0; j < ty; j++)
        for (int k = 0; k < tx; k++) v.push_back(DeltaPixel{warpdiff[(size_t)j * tx + k], (uint16_t)k, (uint16_t)j});
    const size_t selected_count = static_cast<size_t>(v.size() * fraction);
    std::nth_element(v.begin(), v.begin() + selected_count, v.end(),
                     [](const DeltaPixel& lhs, const DeltaPixel& rhs) { return lhs.abs_delta < rhs.abs_delta; });
    v.resize(selected_count);
    for (size_t i = 0; i < v.size(); i++) out_idx[i] = (int32_t)v[i].tile_y * tx + v[i].tile_x;
    return (int)selected_count;
}

/* The same keep-best-fraction step under a documented, STL-independent rule: keep the selected_count
 * tiles that are smallest by (abs_delta, tile index) -- ties on abs_delta go to the lower tile index -- and hand them out in
 * ascending tile order.  Any conforming std::nth_element may produce this set (its comparator sees abs_delta only, so which of
 * the tied elements land in front of nth is unspecified); the order of the survivors, which the reference leaves to the STL,
 * is fixed here.  The product's VS_SELECT_STABLE mode is compared with this bit for bit. */
int vso_select_smallest_stable(const uint16_t* warpdiff, int tx, int ty, float fraction, int32_t* out_idx) {
    const size_t n = (size_t)tx * ty;
    const size_t selected_count = static_cast<size_t>(n * fraction);
    if (selected_count == 0) return 0;
    std::vector<uint32_t> key(n);
    for (size_t i = 0; i < n; i++) key[i] = ((uint32_t)warpdiff[i] << 16) | (uint32_t)(i & 0xffffu);   /* i < 65536 tiles per level */
    std::vector<uint32_t> sorted(key);
    std::sort(sorted.begin(), sorted.end());
    const uint32_t cut = sorted[selected_count - 1];
    size_t m = 0;
    for (size_t i = 0; i < n; i++)
        if (key[i] <= cut) out_idx[m++] = (int32_t)i;
    return (int)m;      /* == selected_count: the keys are distinct */
}

/* Test-input generator: a warpdiff table on which std::nth_element(begin, begin + n*fraction, end) -- this libstdc++'s, the
 * call above -- runs out of its introselect depth budget (2 * lg n partitions) and falls back to heap-select.  McIlroy's
 * adversary ("A Killer Adversary for Quicksort", 1999): the algorithm runs on item ids whose values are decided lazily --
 * unfrozen items count as larger than every frozen one, and when two unfrozen items meet, the one last seen as a pivot
 * candidate is frozen at the next smallest value -- so every median-of-3 pivot ends up among the smallest of its range and a
 * partition peels off a constant number of elements.  The frozen values, replayed as data, make the same comparisons come out
 * the same way.  Returns the number of distinct values used (the rest share the value `n_solid`), or -1 if the table would not fit
 * 16 bits.  The product's on-device replica must flag exactly these inputs (tests/test_select_gpu.py). */
int vso_nth_element_killer(int tx, int ty, float fraction, uint16_t* out) {
    const int n = tx * ty;
    if (n < 1 || n > 65535) return -1;
    const int gas = n + 1;
    std::vector<int> val((size_t)n, gas), item((size_t)n);
    for (int i = 0; i < n; i++) item[i] = i;
    int nsolid = 0, candidate = 0;
    auto cmp = [&](int x, int y) {
        if (val[x] == gas && val[y] == gas) { if (x == candidate) val[x] = nsolid++; else val[y] = nsolid++; }
        if (val[x] == gas) candidate = x; else if (val[y] == gas) candidate = y;
        return val[x] < val[y];
    };
    const size_t selected_count = static_cast<size_t>((size_t)n * fraction);
    std::nth_element(item.begin(), item.begin() + selected_count, item.end(), cmp);
    for (int i = 0; i < n; i++) out[i] = (uint16_t)(val[i] == gas ? nsolid : val[i]);
    return nsolid;
}

/* How many partition rounds libstdc++'s introselect spends on a table before it stops (range <= 3) or gives up: counted by
 * running the same call with a comparator that watches for the median-of-3 pattern is fragile, so this restates the control
 * flow of bits/stl_algo.h __introselect literally (depth budget 2 * lg n, __unguarded_partition_pivot = median of first+1 /
 * mid / last-1 to first, Hoare partition of [first+1, last)) on a copy of the keys and reports 1 when the budget reaches 0
 * with more than 3 elements left -- the condition under which std::nth_element calls __heap_select. */
int vso_nth_element_hits_depth_limit(const uint16_t* warpdiff, int n, float fraction) {
    std::vector<uint16_t> a(warpdiff, warpdiff + n);
    const long nth = (long)static_cast<size_t>((size_t)n * fraction);
    if (n == 0 || nth == n) return 0;
    long first = 0, last = n;
    int depth = 0;
    for (long m = n; m > 1; m >>= 1) depth++;
    depth *= 2;
    while (last - first > 3) {
        if (depth == 0) return 1;
        --depth;
        const long mid = first + (last - first) / 2, ia = first + 1, ib = mid, ic = last - 1;
        long m;
        if (a[ia] < a[ib]) m = (a[ib] < a[ic]) ? ib : ((a[ia] < a[ic]) ? ic : ia);
        else if (a[ia] < a[ic]) m = ia;
        else if (a[ib] < a[ic]) m = ic;
        else m = ib;
        std::swap(a[first], a[m]);
        long i = first + 1, j = last;
        const uint16_t pv = a[first];
        for (;;) {
            while (a[i] < pv) ++i;
            --j;
            while (pv < a[j]) --j;
            if (!(i < j)) break;
            std::swap(a[i], a[j]);
            ++i;
        }
        if (i <= nth) first = i; else last = i;
    }
    return 0;
}

/* alignment.cpp:278-332.  jac planar (n,4): element (i,c) at c*n+i */
void vso_hessian(const float* jacx, int nx, const float* jacy, int ny, double H[16]) {
    for (int i = 0; i < 16; i++) H[i] = 0.0;
    for (int set = 0; set < 2; set++) {
        const float* jac = set == 0 ? jacx : jacy;
        const int m = set == 0 ? nx : ny;
        for (int i = 0; i < m; i++) {
            double j[4] = {jac[0 * (size_t)m + i], jac[1 * (size_t)m + i], jac[2 * (size_t)m + i], jac[3 * (size_t)m + i]};
            for (int r = 0; r < 4; r++)
                for (int c = r; c < 4; c++) H[r * 4 + c] += j[r] * j[c];
        }
    }
    for (int r = 0; r < 4; r++)
        for (int c = r + 1; c < 4; c++) H[c * 4 + r] = H[r * 4 + c];
}

/* alignment.cpp:555-583.  cv::SVD singular values of a symmetric PSD matrix = its
 * eigenvalues (abs), sorted descending.  Mat::inv(DECOMP_SVD) = V diag(1/w) V^T with
 * w <= 2*DBL_EPSILON*sum(w) treated as 0 (OpenCV SVBackSubst threshold). */
double vso_condition_and_invert(double H[16], double Hinv[16]) {
    double ev[4], V[16];
    jacobi_eig4(H, ev, V);
    double max_sv = 0.0, min_sv = 1e300;
    for (int i = 0; i < 4; i++) { double a = std::fabs(ev[i]); max_sv = std::max(max_sv, a); min_sv = std::min(min_sv, a); }
    double condition_number = max_sv / (min_sv + 1e-10);
    if (condition_number > 1e6) {
        double lambda = 1e-6 * max_sv;
        for (int i = 0; i < 4; i++) H[i * 4 + i] += lambda;
        jacobi_eig4(H, ev, V);
    }
    double sum = 0.0;
    for (int i = 0; i < 4; i++) sum += std::fabs(ev[i]);
    double thresh = 2.0 * 2.220446049250313e-16 * sum;
    for (int r = 0; r < 4; r++)
        for (int c = 0; c < 4; c++) {
            double s = 0.0;
            for (int k = 0; k < 4; k++)
                if (std::fabs(ev[k]) > thresh) s += V[r * 4 + k] * V[c * 4 + k] / ev[k];
            Hinv[r * 4 + c] = s;
        }
    return condition_number;
}

/* smoother.cpp:18-65 */
void vso_tvl1_smooth(const double* data, int n, double lambda, int iterations, double* x) {
    const size_t N = (size_t)n;
    if (N == 0) return;
    for (size_t i = 0; i < N; i++) x[i] = data[i];
    for (int iter = 0; iter < iterations; ++iter) {
        for (size_t i = 0; i < N; i++) {
            double alpha = 0.5;
            x[i] = (1.0 - alpha) * x[i] + alpha * data[i];
        }
        for (size_t i = 0; i + 1 < N; i++) {
            double diff = x[i + 1] - x[i];
            double mag = std::fabs(diff);
            if (mag > lambda) {
                double shrink = (mag - lambda) / mag * 0.5;
                x[i] += diff * shrink;
                x[i + 1] -= diff * shrink;
            } else {
                double mid = 0.5 * (x[i] + x[i + 1]);
                x[i] = mid;
                x[i + 1] = mid;
            }
        }
    }
}

}  // extern "C"

/* smoother.cpp:67-127 */
struct vso_smoother {
    int lagBehind, lagAhead; double lambda; int nextToFinalize;
    std::vector<vso_transform> measurements;
};

extern "C" {
vso_smoother* vso_smoother_create(int lag_behind, int lag_ahead, double lambda) {
    return new vso_smoother{lag_behind, lag_ahead, lambda, 0, {}};
}
void vso_smoother_destroy(vso_smoother* s) { delete s; }
int vso_smoother_update(vso_smoother* s, const vso_transform* meas, vso_transform* outFinalized) {
    s->measurements.push_back(*meas);
    const int newestIndex = (int)s->measurements.size() - 1;
    if (s->nextToFinalize + s->lagAhead > newestIndex) return 0;
    int startIndex = std::max(0, s->nextToFinalize - s->lagBehind);
    int endIndex = s->nextToFinalize + s->lagAhead;
    std::vector<double> v[4], sm[4];
    for (int i = startIndex; i <= endIndex; i++) {
        const vso_transform& m = s->measurements[i];
        v[0].push_back(m.A); v[1].push_back(m.B); v[2].push_back(m.TX); v[3].push_back(m.TY);
    }
    for (int k = 0; k < 4; k++) { sm[k].resize(v[k].size()); vso_tvl1_smooth(v[k].data(), (int)v[k].size(), s->lambda, 100, sm[k].data()); }
    int middle = s->nextToFinalize - startIndex;
    outFinalized->A = sm[0][middle]; outFinalized->B = sm[1][middle];
    outFinalized->TX = sm[2][middle]; outFinalized->TY = sm[3][middle];
    s->nextToFinalize++;
    return 1;
}
}

/* ---- VideoAligner ------------------------------------------------------------------------ */
struct vso_aligner {
    /* alignment.hpp:61-94 */
    int CurrFrameIndex = 0, PrevFrameIndex = 1, FramesAccumulated = 0;
    static constexpr int KeyframeIndex = 1, NonKeyframeIndex = 0;
    static constexpr int PhaseLevel = 2;                 /* alignment.hpp:69 */
    std::vector<float> PhaseImage[2];
    int PyramidLevels = -1;
    int LastWidth = -1, LastHeight = -1;
    struct Level {
        int w = 0, h = 0, ts = 0, tx = 0, ty = 0;
        std::vector<uint8_t> img[2];
        std::vector<float> gx, gy;
        std::vector<uint16_t> amx, amy;
        std::vector<float> jx, jy;
        std::vector<uint16_t> wdx, wdy;
        std::vector<uint16_t> selx, sely;
        std::vector<float> seljx, seljy;
    };
    std::vector<Level> L;
    std::vector<int32_t> idx;
    vso_align_debug dbg;
    int select_rule = 0;      /* 0: std::nth_element as the reference (this libstdc++'s order); 1: vso_select_smallest_stable */

    bool ComputePyramid(const void* frame, int width, int height, int stride, int format, const vso_aligner_params& params);
    bool ComputeKeyFrame();
    int Align(const void* frame, int w, int h, int stride, int format, const vso_aligner_params& params, vso_transform& transform);
};

/* alignment.cpp:149-235 */
bool vso_aligner::ComputePyramid(const void* frame, int width0, int height0, int stride, int format,
                                 const vso_aligner_params& params) {
    int width = width0, height = height0;
    if (L.empty() || width != LastWidth || height != LastHeight) {
        CurrFrameIndex = 0; PrevFrameIndex = 1; FramesAccumulated = 0;
        LastWidth = width; LastHeight = height;
        PyramidLevels = 0;
        do { PyramidLevels++; width /= 2; height /= 2; }
        while (width >= params.pyramid_min_width && height >= params.pyramid_min_height);
        L.assign(PyramidLevels, Level());
        width = width0; height = height0;
        for (int i = 0; i < PyramidLevels; i++) {
            if (i > 0) { width /= 2; height /= 2; }
            L[i].w = width; L[i].h = height;
            L[i].img[0].assign((size_t)width * height, 0);
            L[i].img[1].assign((size_t)width * height, 0);
            L[i].gx.assign((size_t)width * height, 0.f);
            L[i].gy.assign((size_t)width * height, 0.f);
        }
    } else {
        PrevFrameIndex = CurrFrameIndex;
        CurrFrameIndex ^= 1;
    }
    /* :212 cvtColor (or a straight copy when the caller already has grayscale) */
    uint8_t* g0 = L[0].img[CurrFrameIndex].data();
    if (format == VSO_FMT_GRAY8) {
        for (int y = 0; y < height0; y++) std::memcpy(g0 + (size_t)y * width0, (const uint8_t*)frame + (size_t)y * stride, width0);
    } else if (format == VSO_FMT_BGR8) {
        vso_bgr_to_gray(frame, width0, height0, stride, 8, 0, g0, width0);
    } else {
        vso_bgr_to_gray(frame, width0, height0, stride, 16, vso_format_bits(format) - 8, g0, width0);
    }
    for (int i = 1; i < PyramidLevels; i++)
        vso_pyr_down(L[i - 1].img[CurrFrameIndex].data(), L[i - 1].w, L[i - 1].h, L[i - 1].w,
                     L[i].img[CurrFrameIndex].data(), L[i].w, L[i].h, L[i].w);
    /* :225-229 PhaseImage = level PhaseLevel (2) as CV_32F, made for every frame whatever phase_correlate says */
    {
        const Level& pl = L[PhaseLevel];
        PhaseImage[CurrFrameIndex].resize((size_t)pl.w * pl.h);
        for (size_t i = 0; i < PhaseImage[CurrFrameIndex].size(); i++) PhaseImage[CurrFrameIndex][i] = (float)pl.img[CurrFrameIndex][i];
    }
    if (FramesAccumulated >= 2) return true;
    return ++FramesAccumulated >= 2;
}

/* alignment.cpp:237-276 */
bool vso_aligner::ComputeKeyFrame() {
    for (int i = 0; i < PyramidLevels; i++) {
        Level& l = L[i];
        vso_grad_xy(l.img[CurrFrameIndex].data(), l.w, l.h, l.w, l.gx.data(), l.gy.data());
        l.ts = vso_tile_size(l.w, l.h);
        l.tx = l.w / l.ts; l.ty = l.h / l.ts;
        size_t n = (size_t)l.tx * l.ty;
        l.amx.assign(n * 2, 0); l.amy.assign(n * 2, 0);
        vso_grad_argmax(l.gx.data(), l.gy.data(), l.w, l.h, l.ts, l.amx.data(), l.amy.data());
        l.jx.assign(n * 4, 0.f); l.jy.assign(n * 4, 0.f);
        vso_sparse_jac(l.gx.data(), l.gy.data(), l.w, l.h, l.amx.data(), l.amy.data(), l.tx, l.ty, l.jx.data(), l.jy.data());
    }
    return true;
}

/* alignment.cpp:334-704 */
int vso_aligner::Align(const void* frame, int w, int h, int stride, int format, const vso_aligner_params& params,
                       vso_transform& transform) {
    transform = vso_transform{0, 0, 0, 0};
    std::memset(&dbg, 0, sizeof(dbg));
    if (!ComputePyramid(frame, w, h, stride, format, params)) { dbg.levels = PyramidLevels; dbg.fail_reason = 1; return 0; }
    dbg.levels = PyramidLevels;
    if (CurrFrameIndex == KeyframeIndex) {
        if (!ComputeKeyFrame()) { LastWidth = -1; return 0; }
    }
    /* :369-388 */
    if (params.phase_correlate) {
        const Level& pl = L[PhaseLevel];
        const std::vector<float>& prev = PhaseImage[PrevFrameIndex];
        const std::vector<float>& curr = PhaseImage[CurrFrameIndex];
        {
            double sx = 0.0, sy = 0.0, response = 0.0;
            vso_phase_correlate(prev.data(), curr.data(), pl.w, pl.h, &sx, &sy, &response);
            dbg.phase_dx = sx; dbg.phase_dy = sy; dbg.phase_response = response;
            if (response > params.phase_correlate_threshold) {
                const float phase_layer_scale = (1 << PhaseLevel) / float(1 << PyramidLevels);
                transform.TX = sx * phase_layer_scale;
                transform.TY = sy * phase_layer_scale;
                if (CurrFrameIndex == KeyframeIndex) { transform.TX = -transform.TX; transform.TY = -transform.TY; }
            }
        }
    }
    for (int i = PyramidLevels - 1; i >= 0; i--) {
        Level& l = L[i];
        const uint8_t* template_image = l.img[NonKeyframeIndex].data();
        const uint8_t* keyframe_image = l.img[KeyframeIndex].data();
        const int image_width = l.w, image_height = l.h;
        const size_t ntiles = (size_t)l.tx * l.ty;
        float p[4];
        vso_ul_params_sparse(&transform, image_width, image_height, p);
        l.wdx.resize(ntiles); l.wdy.resize(ntiles);
        vso_sparse_warpdiff(template_image, keyframe_image, image_width, image_height, image_width, l.amx.data(), l.tx, l.ty, p[0], p[1], p[2], p[3], l.wdx.data());
        vso_sparse_warpdiff(template_image, keyframe_image, image_width, image_height, image_width, l.amy.data(), l.tx, l.ty, p[0], p[1], p[2], p[3], l.wdy.data());

        /* :435-546 selection + gather */
        idx.resize(ntiles);
        int (*const select)(const uint16_t*, int, int, float, int32_t*) = select_rule ? vso_select_smallest_stable : vso_select_smallest;
        int nx = select(l.wdx.data(), l.tx, l.ty, params.smallest_fraction, idx.data());
        l.selx.resize((size_t)nx * 2); l.seljx.resize((size_t)nx * 4);
        for (int j = 0; j < nx; j++) {
            l.selx[j] = l.amx[idx[j]]; l.selx[(size_t)nx + j] = l.amx[ntiles + idx[j]];
            for (int k = 0; k < 4; k++) l.seljx[(size_t)k * nx + j] = l.jx[(size_t)k * ntiles + idx[j]];
        }
        int ny = select(l.wdy.data(), l.tx, l.ty, params.smallest_fraction, idx.data());
        l.sely.resize((size_t)ny * 2); l.seljy.resize((size_t)ny * 4);
        for (int j = 0; j < ny; j++) {
            l.sely[j] = l.amy[idx[j]]; l.sely[(size_t)ny + j] = l.amy[ntiles + idx[j]];
            for (int k = 0; k < 4; k++) l.seljy[(size_t)k * ny + j] = l.jy[(size_t)k * ntiles + idx[j]];
        }
        dbg.tile_size[i] = l.ts; dbg.selected_x[i] = nx; dbg.selected_y[i] = ny;

        double H[16], Hinv[16];
        vso_hessian(l.seljx.data(), nx, l.seljy.data(), ny, H);
        dbg.condition[i] = vso_condition_and_invert(H, Hinv);

        /* :585-598 */
        double cx_img = image_width * 0.5, cy_img = image_height * 0.5;
        vso_point corner[4] = {{0.f, 0.f}, {image_width - 1.f, 0.f}, {0.f, image_height - 1.f}, {image_width - 1.f, image_height - 1.f}};
        vso_point c0[4], c1[4];
        for (int k = 0; k < 4; k++) c1[k] = c0[k] = vso_transform_warp_center(&transform, corner[k], cx_img, cy_img);

        for (int iter = 0; iter < params.max_iters; iter++) {
            dbg.iterations[i]++;
            double b[4];
            vso_ul_params_sparse(&transform, image_width, image_height, p);
            vso_sparse_ica(template_image, keyframe_image, image_width, image_height, image_width, l.selx.data(), nx,
                           l.sely.data(), ny, l.seljx.data(), l.seljy.data(), p[0], p[1], p[2], p[3], b);
            double dt[4];
            for (int r = 0; r < 4; r++) {   /* dt = Hinv * b (:624) */
                double s = 0.0;
                for (int k = 0; k < 4; k++) s += Hinv[r * 4 + k] * b[k];
                dt[r] = s;
            }
            double scale = 1.0 / image_width;            /* :629 */
            vso_transform delta{dt[0] * scale, dt[1] * scale, dt[2], dt[3]};
            transform = vso_transform_compose(&delta, &transform);   /* :639 */

            vso_point c2[4];
            for (int k = 0; k < 4; k++) c2[k] = vso_transform_warp_center(&transform, corner[k], cx_img, cy_img);
            double ud12 = std::max(pt_dist(c2[0], c1[0]), pt_dist(c2[1], c1[1]));
            double ld12 = std::max(pt_dist(c2[2], c1[2]), pt_dist(c2[3], c1[3]));
            double displacement12 = std::max(ud12, ld12);
            for (int k = 0; k < 4; k++) c1[k] = c2[k];
            if (displacement12 < params.threshold) break;
            if (iter >= params.max_iters - 1) { dbg.fail_reason = 2; dbg.fail_level = i; return 0; }
        }
        double ud01 = std::max(pt_dist(c0[0], c1[0]), pt_dist(c0[1], c1[1]));
        double ld01 = std::max(pt_dist(c0[2], c1[2]), pt_dist(c0[3], c1[3]));
        double displacement01 = std::max(ud01, ld01);
        dbg.level_transform[i] = transform;
        if (displacement01 > params.max_displacement) { dbg.fail_reason = 3; dbg.fail_level = i; return 0; }
        if (i > 0) { transform.TX *= 2.0; transform.TY *= 2.0; }
    }
    if (CurrFrameIndex != KeyframeIndex) transform = vso_transform_inverse(&transform);
    return 1;
}

extern "C" {
vso_aligner* vso_aligner_create(void) { return new vso_aligner(); }
int vso_aligner_set_select_rule(vso_aligner* a, int rule) {
    if (!a || (rule != 0 && rule != 1)) return -1;
    a->select_rule = rule;
    return 0;
}
void vso_aligner_destroy(vso_aligner* a) { delete a; }
int vso_aligner_align_next(vso_aligner* a, const void* frame, int w, int h, int stride, int format,
                           const vso_aligner_params* params, vso_transform* out) {
    if (!a || !frame || !out || w < 8 || h < 8) return -1;
    vso_aligner_params p;
    if (params) p = *params; else vso_aligner_params_default(&p);
    /* PhaseLevel = 2 is indexed unconditionally (alignment.cpp:227): >=3 levels required */
    { int lv = 0, ww = w, hh = h; do { lv++; ww /= 2; hh /= 2; } while (ww >= p.pyramid_min_width && hh >= p.pyramid_min_height); if (lv < 3 || lv > 16) return -3; }
    return a->Align(frame, w, h, stride, format, p, *out);
}
const vso_align_debug* vso_aligner_debug(const vso_aligner* a) { return &a->dbg; }
int vso_aligner_level_dims(const vso_aligner* a, int level, int* w, int* h, int* tx, int* ty, int* ts) {
    if (level < 0 || level >= (int)a->L.size()) return -1;
    const auto& l = a->L[level];
    *w = l.w; *h = l.h; *tx = l.tx; *ty = l.ty; *ts = l.ts;
    return 0;
}
const uint8_t* vso_aligner_level_image(const vso_aligner* a, int slot, int level) { return a->L[level].img[slot].data(); }
const uint16_t* vso_aligner_level_argmax(const vso_aligner* a, int level, int set) { return set == 0 ? a->L[level].amx.data() : a->L[level].amy.data(); }
const float* vso_aligner_level_jacobian(const vso_aligner* a, int level, int set) { return set == 0 ? a->L[level].jx.data() : a->L[level].jy.data(); }
}

/* ---- VideoStabilizer (stabilizer.cpp:3-117) -------------------------------------------- */
struct vso_stabilizer {
    vso_stabilizer_params params;
    vso_aligner aligner;
    int frameIndex = 0;
    vso_smoother smoother;
    std::deque<vso_transform> measurementBuffer;
    std::deque<std::vector<uint8_t>> frameBuffer;   /* raw bytes of each buffered frame */
    vso_transform accum{0, 0, 0, 0};
    vso_transform lastMeas{0, 0, 0, 0};
    int lastSuccess = 0;
    explicit vso_stabilizer(const vso_stabilizer_params& p)
        : params(p), smoother{p.lag, p.smoother_memory, p.lambda, 0, {}} {}
};

extern "C" {
int vso_stabilizer_set_select_rule(vso_stabilizer* s, int rule) { return s ? vso_aligner_set_select_rule(&s->aligner, rule) : -1; }
vso_stabilizer* vso_stabilizer_create(const vso_stabilizer_params* p) {
    vso_stabilizer_params d;
    if (p) d = *p; else vso_stabilizer_params_default(&d);
    return new vso_stabilizer(d);
}
void vso_stabilizer_destroy(vso_stabilizer* s) { delete s; }

int vso_stabilizer_process(vso_stabilizer* s, const void* frame, int w, int h, int stride, int format, void* out,
                           int* out_w, int* out_h) {
    if (!s || !frame || format == VSO_FMT_GRAY8 || vso_format_bits(format) == 0) return -1;
    const int depth = vso_format_bits(format);              /* bits the samples use */
    const int bits = depth > 8 ? 16 : 8;                    /* container */
    const size_t esz = bits / 8;
    ++s->frameIndex;
    /* :15 clone */
    std::vector<uint8_t> copy((size_t)w * h * 3 * esz);
    for (int y = 0; y < h; y++)
        std::memcpy(copy.data() + (size_t)y * w * 3 * esz, (const uint8_t*)frame + (size_t)y * stride * esz, (size_t)w * 3 * esz);
    s->frameBuffer.push_back(std::move(copy));

    vso_transform currentMeas{0, 0, 0, 0};
    bool success = vso_aligner_align_next(&s->aligner, frame, w, h, stride, format, &s->params.aligner, &currentMeas) == 1;
    s->lastMeas = currentMeas; s->lastSuccess = success ? 1 : 0;

    bool reset = !success;
    vso_transform earliestSmoothed{0, 0, 0, 0};
    if (s->params.enable_smoother) vso_smoother_update(&s->smoother, &currentMeas, &earliestSmoothed);   /* :35, return ignored */
    if (reset) s->accum = vso_transform{0, 0, 0, 0};
    s->measurementBuffer.push_back(currentMeas);
    bool hasFinalized = s->measurementBuffer.size() > (size_t)s->params.lag;
    int produced = 0;
    if (hasFinalized) {
        vso_transform earliestMeas = s->measurementBuffer.front();
        s->measurementBuffer.pop_front();
        vso_transform jitter;
        if (s->params.enable_smoother) {
            vso_transform inv = vso_transform_inverse(&earliestSmoothed);
            jitter = vso_transform_compose(&earliestMeas, &inv);
        } else {
            jitter = earliestMeas;
        }
        vso_transform newAccum = vso_transform_compose(&s->accum, &jitter);
        double displacement = vso_transform_max_corner_displacement(&newAccum, w, h);
        double decay = 1.0;
        if (displacement > s->params.max_disp) {
            decay = s->params.max_decay;
        } else if (displacement > s->params.min_disp) {
            double f = (displacement - s->params.min_disp) / (s->params.max_disp - s->params.min_disp);
            f = std::max(0.0, std::min(1.0, f));
            decay = s->params.min_decay * (1.0 - f) + s->params.max_decay * f;
        } else {
            decay = s->params.min_decay;
        }
        newAccum.TX *= decay; newAccum.TY *= decay; newAccum.A *= decay; newAccum.B *= decay;
        s->accum = newAccum;
        if (!s->frameBuffer.empty()) {
            std::vector<uint8_t> frameToStabilize = std::move(s->frameBuffer.front());
            s->frameBuffer.pop_front();
            /* :97-99: warpBySimilarityTransform(frame, correction) where cv::warpAffine
             * *inverts* the matrix it is given (imgproc.cpp:472) => sampling map = correction^-1 */
            vso_transform correction = vso_transform_inverse(&newAccum);
            vso_transform sampling = vso_transform_inverse(&correction);
            std::vector<uint8_t> warped((size_t)w * h * 3 * esz);
            /* (VSO_WARP_BILINEAR_CV restates cv::warpAffine itself, inversion included: it takes the correction as the reference hands it over) */
            vso_bgr_image_warp(frameToStabilize.data(), w, h, w * 3, 3, bits, s->params.warp_mode == VSO_WARP_BILINEAR_CV ? &correction : &sampling,
                               s->params.warp_mode, s->params.warp_border, (1 << depth) - 1, warped.data(), w * 3);
            int c = s->params.crop_pixels > 0 ? s->params.crop_pixels : 0;
            int ow = w - 2 * c, oh = h - 2 * c;
            for (int y = 0; y < oh; y++)
                std::memcpy((uint8_t*)out + (size_t)y * ow * 3 * esz, warped.data() + ((size_t)(y + c) * w + c) * 3 * esz, (size_t)ow * 3 * esz);
            *out_w = ow; *out_h = oh;
            produced = 1;
        }
    }
    return produced;
}
void vso_stabilizer_state(const vso_stabilizer* s, vso_transform* last_meas, vso_transform* accum, int* last_success) {
    if (last_meas) *last_meas = s->lastMeas;
    if (accum) *accum = s->accum;
    if (last_success) *last_success = s->lastSuccess;
}
}
