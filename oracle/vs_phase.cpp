/* vs_phase.cpp -- CPU restatement of the phase-correlation initialisation.
 *
 * TEST INFRASTRUCTURE ONLY (see vs_oracle.h).  PARITY UNPINNED: the arithmetic lives in OpenCV
 * (cv::phaseCorrelate, called at alignment.cpp:374 on the CV_32F copies of pyramid level 2 made at
 * alignment.cpp:225-229; apt libopencv-dev, version not pinned by the reference), which is not in this image and has
 * no golden vectors in the reference.  What is restated is OpenCV 4.x's published structure
 * (modules/imgproc/src/phasecorr.cpp):
 *   1. pad both images with zeros to getOptimalDFTSize (smallest 2^a 3^b 5^c >= extent), no window (cv::noArray())
 *   2. forward real DFT of both, single precision
 *   3. P = F1 * conj(F2)                                            (mulSpectrums, conjB = true)
 *   4. C = P / |P|: |P| = float(sqrt(double re^2 + double im^2)), the division done in double with FLT_EPSILON added to
 *      the squared magnitude                                         (magSpectrums + divSpectrums)
 *   5. inverse DFT without scaling, fftShift
 *   6. peak = first maximum in row-major order (minMaxLoc); 5x5 window around it clipped to the image:
 *      centroid = sum(x v, y v) / (sum v + DBL_EPSILON) in double, response = sum v / (M N)
 *   7. shift = (N/2, M/2) - centroid
 * OpenCV's own FFT (radix order, CCS packing, SIMD paths) cannot be reproduced bit for bit; the DFT here is the
 * build's own mixed-radix Stockham transform, specified below so that the GPU kernels (vs_phase.hip) evaluate the same
 * float operations in the same order and agree with this file bit for bit.
 *
 * Transform specification (shared with the product):
 *   - radix plan of n: all factors 5, then all factors 3, then factors 4 while divisible, then one factor 2
 *   - pass with radix r on sub-length n_cur (m = n_cur / r), stride s:  for p < m, q < s:
 *       a_j = src[q + s (p + m j)],  b = DFT_r(a) (the butterflies below),  dst[q + s (r p + k)] = b_k * W[p k (n / n_cur)]
 *     (b_0 is stored unmultiplied), then n_cur = m, s = s r, buffers swap
 *   - W[j] = (float cos(2 pi j / n), float -sin(2 pi j / n)) from double-precision libm
 *   - complex product (x.re w.re - x.im w.im, x.re w.im + x.im w.re), no fused multiply-add
 *   - inverse = conj(forward(conj(x))), unscaled
 *   - 2-D: rows first (real input as complex with zero imaginary part, columns 0 .. n/2 kept), then columns; the
 *     inverse runs columns first, then rows on the Hermitian extension X[n - c] = conj(X[c]), real part kept.
 */
#include <cfloat>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

#include "vs_oracle.h"

namespace {

struct cf { float re, im; };
inline cf cadd(cf a, cf b) { return {a.re + b.re, a.im + b.im}; }
inline cf csub(cf a, cf b) { return {a.re - b.re, a.im - b.im}; }
inline cf cmul(cf x, cf w) { return {x.re * w.re - x.im * w.im, x.re * w.im + x.im * w.re}; }
inline cf conj(cf a) { return {a.re, -a.im}; }

struct Plan {
    int n = 0, passes = 0;
    int radix[32];
    std::vector<cf> tw;
};

Plan make_plan(int n) {
    Plan p;
    p.n = n;
    int r = n;
    while (r % 5 == 0) { p.radix[p.passes++] = 5; r /= 5; }
    while (r % 3 == 0) { p.radix[p.passes++] = 3; r /= 3; }
    while (r % 4 == 0) { p.radix[p.passes++] = 4; r /= 4; }
    if (r % 2 == 0) { p.radix[p.passes++] = 2; r /= 2; }
    if (r != 1) { p.passes = -1; return p; }
    p.tw.resize(n);
    for (int j = 0; j < n; j++) {
        const double a = 2.0 * M_PI * (double)j / (double)n;
        p.tw[j] = {(float)std::cos(a), (float)-std::sin(a)};
    }
    return p;
}

const float kS3 = 0.8660254037844386f;                                   /* sin(2 pi / 3) */
const float kC51 = 0.30901699437494745f, kC52 = -0.8090169943749473f;    /* cos(2 pi / 5), cos(4 pi / 5) */
const float kS51 = 0.9510565162951535f, kS52 = 0.5877852522924731f;      /* sin(2 pi / 5), sin(4 pi / 5) */

inline void butterfly(int r, const cf* a, cf* b) {
    if (r == 2) {
        b[0] = cadd(a[0], a[1]);
        b[1] = csub(a[0], a[1]);
    } else if (r == 4) {
        const cf t0 = cadd(a[0], a[2]), t1 = csub(a[0], a[2]), t2 = cadd(a[1], a[3]), t3 = csub(a[1], a[3]);
        b[0] = cadd(t0, t2);
        b[2] = csub(t0, t2);
        b[1] = {t1.re + t3.im, t1.im - t3.re};
        b[3] = {t1.re - t3.im, t1.im + t3.re};
    } else if (r == 3) {
        const cf t1 = cadd(a[1], a[2]), t2 = csub(a[1], a[2]);
        b[0] = cadd(a[0], t1);
        const cf m = {a[0].re - 0.5f * t1.re, a[0].im - 0.5f * t1.im};
        const float nr = kS3 * t2.im, ni = kS3 * t2.re;
        b[1] = {m.re + nr, m.im - ni};
        b[2] = {m.re - nr, m.im + ni};
    } else {
        const cf t1 = cadd(a[1], a[4]), t2 = cadd(a[2], a[3]), t3 = csub(a[1], a[4]), t4 = csub(a[2], a[3]);
        b[0] = cadd(cadd(a[0], t1), t2);
        const cf m1 = {(a[0].re + kC51 * t1.re) + kC52 * t2.re, (a[0].im + kC51 * t1.im) + kC52 * t2.im};
        const cf m2 = {(a[0].re + kC52 * t1.re) + kC51 * t2.re, (a[0].im + kC52 * t1.im) + kC51 * t2.im};
        const cf n1 = {kS51 * t3.re + kS52 * t4.re, kS51 * t3.im + kS52 * t4.im};
        const cf n2 = {kS52 * t3.re - kS51 * t4.re, kS52 * t3.im - kS51 * t4.im};
        b[1] = {m1.re + n1.im, m1.im - n1.re};
        b[4] = {m1.re - n1.im, m1.im + n1.re};
        b[2] = {m2.re + n2.im, m2.im - n2.re};
        b[3] = {m2.re - n2.im, m2.im + n2.re};
    }
}

/* forward transform of x[0..n), result back in x; tmp is scratch of n */
void fft_forward(const Plan& plan, cf* x, cf* tmp) {
    cf* src = x;
    cf* dst = tmp;
    int n_cur = plan.n, s = 1;
    for (int pass = 0; pass < plan.passes; pass++) {
        const int r = plan.radix[pass], m = n_cur / r, tstep = plan.n / n_cur;
        for (int p = 0; p < m; p++) {
            for (int q = 0; q < s; q++) {
                cf a[5], b[5];
                for (int j = 0; j < r; j++) a[j] = src[q + s * (p + m * j)];
                butterfly(r, a, b);
                dst[q + s * (r * p)] = b[0];
                for (int k = 1; k < r; k++) dst[q + s * (r * p + k)] = cmul(b[k], plan.tw[(size_t)p * k * tstep]);
            }
        }
        cf* t = src; src = dst; dst = t;
        n_cur = m;
        s *= r;
    }
    if (src != x) std::memcpy(x, src, sizeof(cf) * plan.n);
}

void fft_inverse(const Plan& plan, cf* x, cf* tmp) {
    for (int i = 0; i < plan.n; i++) x[i] = conj(x[i]);
    fft_forward(plan, x, tmp);
    for (int i = 0; i < plan.n; i++) x[i] = conj(x[i]);
}

/* half spectrum [M][NC] of a w x h float image zero-padded to N x M */
void spectrum(const float* img, int w, int h, const Plan& pn, const Plan& pm, std::vector<cf>& F) {
    const int N = pn.n, M = pm.n, NC = N / 2 + 1;
    F.assign((size_t)M * NC, cf{0.f, 0.f});
    std::vector<cf> line(std::max(N, M)), tmp(std::max(N, M));
    for (int r = 0; r < M; r++) {
        for (int c = 0; c < N; c++) line[c] = {(r < h && c < w) ? img[(size_t)r * w + c] : 0.f, 0.f};
        fft_forward(pn, line.data(), tmp.data());
        for (int c = 0; c < NC; c++) F[(size_t)r * NC + c] = line[c];
    }
    for (int c = 0; c < NC; c++) {
        for (int r = 0; r < M; r++) line[r] = F[(size_t)r * NC + c];
        fft_forward(pm, line.data(), tmp.data());
        for (int r = 0; r < M; r++) F[(size_t)r * NC + c] = line[r];
    }
}

}  // namespace

extern "C" {

/* cv::getOptimalDFTSize: the smallest n' >= n whose only prime factors are 2, 3 and 5 */
int vso_optimal_dft_size(int n) {
    if (n < 1) return -1;
    for (int c = n;; c++) {
        int r = c;
        while (r % 2 == 0) r /= 2;
        while (r % 3 == 0) r /= 3;
        while (r % 5 == 0) r /= 5;
        if (r == 1) return c;
    }
}

/* radix plan of n (0 passes for n = 1); returns the number of passes or -1 when n has another prime factor */
int vso_fft_plan(int n, int* radix) {
    Plan p = make_plan(n);
    for (int i = 0; i < p.passes; i++) radix[i] = p.radix[i];
    return p.passes;
}

/* 1-D complex transform of interleaved (re, im) floats, in place; inverse is unscaled */
int vso_fft_c2c(float* data, int n, int inverse) {
    Plan p = make_plan(n);
    if (p.passes < 0) return -1;
    std::vector<cf> tmp(n);
    if (inverse) fft_inverse(p, reinterpret_cast<cf*>(data), tmp.data());
    else fft_forward(p, reinterpret_cast<cf*>(data), tmp.data());
    return 0;
}

/* the unshifted, unscaled correlation surface (M x N floats; *M, *N = padded extents); NULL surface = size query */
int vso_phase_surface(const float* a, const float* b, int w, int h, float* surface, int* M_out, int* N_out) {
    if (w < 1 || h < 1) return -1;
    const int N = vso_optimal_dft_size(w), M = vso_optimal_dft_size(h), NC = N / 2 + 1;
    if (M_out) *M_out = M;
    if (N_out) *N_out = N;
    if (!surface) return 0;
    const Plan pn = make_plan(N), pm = make_plan(M);
    std::vector<cf> Fa, Fb;
    spectrum(a, w, h, pn, pm, Fa);
    spectrum(b, w, h, pn, pm, Fb);
    /* cross-power spectrum, normalised */
    std::vector<cf> C((size_t)M * NC);
    for (size_t i = 0; i < C.size(); i++) {
        const cf x = Fa[i], y = Fb[i];
        const float re = x.re * y.re + x.im * y.im;
        const float im = x.im * y.re - x.re * y.im;
        const float mag = (float)std::sqrt((double)re * (double)re + (double)im * (double)im);
        const double denom = (double)mag * (double)mag + (double)FLT_EPSILON;
        C[i] = {(float)(((double)re * (double)mag) / denom), (float)(((double)im * (double)mag) / denom)};
    }
    std::vector<cf> line(std::max(N, M)), tmp(std::max(N, M));
    for (int c = 0; c < NC; c++) {
        for (int r = 0; r < M; r++) line[r] = C[(size_t)r * NC + c];
        fft_inverse(pm, line.data(), tmp.data());
        for (int r = 0; r < M; r++) C[(size_t)r * NC + c] = line[r];
    }
    for (int r = 0; r < M; r++) {
        for (int c = 0; c < NC; c++) line[c] = C[(size_t)r * NC + c];
        for (int c = NC; c < N; c++) line[c] = conj(C[(size_t)r * NC + (N - c)]);
        fft_inverse(pn, line.data(), tmp.data());
        for (int c = 0; c < N; c++) surface[(size_t)r * N + c] = line[c].re;
    }
    return 0;
}

/* peak + weighted centroid + response on an unshifted surface (steps 5b-7) */
void vso_phase_peak(const float* surface, int M, int N, double* dx, double* dy, double* response) {
    /* fftShift moves element i to (i + n/2) mod n; the scan below walks the shifted image row-major */
    const int hx = N / 2, hy = M / 2;
    auto at = [&](int sy, int sx) { return surface[(size_t)((sy - hy + M) % M) * N + ((sx - hx + N) % N)]; };
    int px = 0, py = 0;
    float best = at(0, 0);
    for (int sy = 0; sy < M; sy++)
        for (int sx = 0; sx < N; sx++) {
            const float v = at(sy, sx);
            if (v > best) { best = v; px = sx; py = sy; }
        }
    int minr = py - 2, maxr = py + 2, minc = px - 2, maxc = px + 2;
    if (minr < 0) minr = 0;
    if (minc < 0) minc = 0;
    if (maxr > M - 1) maxr = M - 1;
    if (maxc > N - 1) maxc = N - 1;
    double cx = 0.0, cy = 0.0, sum = 0.0;
    for (int y = minr; y <= maxr; y++)
        for (int x = minc; x <= maxc; x++) {
            const double v = (double)at(y, x);
            cx += (double)x * v;
            cy += (double)y * v;
            sum += v;
        }
    *response = sum / (double)((long long)M * N);
    sum += DBL_EPSILON;
    cx /= sum;
    cy /= sum;
    *dx = (double)N / 2.0 - cx;
    *dy = (double)M / 2.0 - cy;
}

/* cv::phaseCorrelate(a, b, noArray(), &response) on two w x h float images (dense rows) */
int vso_phase_correlate(const float* a, const float* b, int w, int h, double* dx, double* dy, double* response) {
    int M = 0, N = 0;
    if (vso_phase_surface(a, b, w, h, nullptr, &M, &N) != 0) return -1;
    std::vector<float> surface((size_t)M * N);
    vso_phase_surface(a, b, w, h, surface.data(), &M, &N);
    vso_phase_peak(surface.data(), M, N, dx, dy, response);
    return 0;
}

/* the same on u8 images, converted like alignment.cpp:227-228 (convertTo CV_32F: exact) */
int vso_phase_correlate_u8(const uint8_t* a, const uint8_t* b, int w, int h, int stride, double* dx, double* dy, double* response) {
    std::vector<float> fa((size_t)w * h), fb((size_t)w * h);
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            fa[(size_t)y * w + x] = (float)a[(size_t)y * stride + x];
            fb[(size_t)y * w + x] = (float)b[(size_t)y * stride + x];
        }
    return vso_phase_correlate(fa.data(), fb.data(), w, h, dx, dy, response);
}

}  /* extern "C" */
