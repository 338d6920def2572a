// vs_phase.hip -- phase-correlation initialisation on gfx950 (alignment.cpp:225-229, 369-388).
//
// cv::phaseCorrelate on pyramid level 2 of successive frames: zero-pad to a 2^a 3^b 5^c extent, real 2-D DFT, normalised
// cross-power spectrum, unscaled inverse DFT, fftShift, first maximum, 5x5 weighted centroid, response.  OpenCV's own
// arithmetic cannot be pinned (not in the image, version unpinned by the reference); the transform specification is the
// build's own and is stated in oracle/vs_phase.cpp, which these kernels follow operation for operation (fp32, no
// contraction), so the result equals the oracle's bit for bit.
//
// Shape of the work: level 2 is small (480x270 at 1080p, 960x540 at 4K), so one transform line lives in LDS and a
// workgroup owns `nb` lines at a time (mixed-radix Stockham passes ping-pong between two LDS buffers).  A frame's half
// spectrum [M][N/2+1] is computed once (rows, then columns in place) and stays in HBM next to its pyramid slot; a pair
// costs one fused cross-power + column pass, one row pass on the Hermitian extension and one peak/centroid reduction.
// Every stage is one launch over all frames / pairs of the batch.  HBM traffic per frame is ~6 passes over 1 MB (1080p):
// the mode is latency/issue-bound like the Gauss-Newton loop, not bandwidth-bound.
#include <cfloat>
#include <cmath>
#include <vector>

#include "vs_phase.hpp"
#include "vs_internal.hpp"
#include "vs_device.hpp"

namespace {

constexpr int kThreads = 256;

__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ float2 cmul(float2 x, float2 w) { return make_float2(x.x * w.x - x.y * w.y, x.x * w.y + x.y * w.x); }
__device__ __forceinline__ float2 cconj(float2 a) { return make_float2(a.x, -a.y); }

constexpr float kS3 = 0.8660254037844386f;
constexpr float kC51 = 0.30901699437494745f, kC52 = -0.8090169943749473f;
constexpr float kS51 = 0.9510565162951535f, kS52 = 0.5877852522924731f;

// One Stockham pass of radix R over nb lines of length plan.n held back to back in LDS (oracle/vs_phase.cpp "pass").
template <int R>
__device__ __forceinline__ void fft_pass(const float2* __restrict__ src, float2* __restrict__ dst, int n, int nb, int m, int s,
                                         int tstep, const float2* __restrict__ tw) {
    const int per = m * s;
    // idx / per and bf / s by multiplication: n / d == umulhi(n, 2^32 / d + 1) for n, d < 2^16 (d > 1)
    const uint32_t inv_per = 0xffffffffu / (uint32_t)per + 1u, inv_s = 0xffffffffu / (uint32_t)s + 1u;
    for (int idx = threadIdx.x; idx < nb * per; idx += kThreads) {
        const int t = per > 1 ? (int)__umulhi((uint32_t)idx, inv_per) : idx, bf = idx - t * per;
        const int p = s > 1 ? (int)__umulhi((uint32_t)bf, inv_s) : bf, q = bf - p * s;
        VS_BOUNDS_CHECK(t * n + q + s * p + s * m * (R - 1), nb * n, 301);        // the butterfly's last input
        VS_BOUNDS_CHECK(t * n + q + s * (R * p) + s * (R - 1), nb * n, 302);      // ... and last output, inside the nb lines
        VS_BOUNDS_CHECK(p * (R - 1) * tstep, n, 303);                              // twiddle table
        const float2* in = src + t * n + q + s * p;          // (record only: an LDS access past the block does not fault)
        float2* out = dst + t * n + q + s * (R * p);
        float2 a[R], b[R];
#pragma unroll
        for (int j = 0; j < R; j++) a[j] = in[s * m * j];
        if constexpr (R == 2) {
            b[0] = cadd(a[0], a[1]);
            b[1] = csub(a[0], a[1]);
        } else if constexpr (R == 4) {
            const float2 t0 = cadd(a[0], a[2]), t1 = csub(a[0], a[2]), t2 = cadd(a[1], a[3]), t3 = csub(a[1], a[3]);
            b[0] = cadd(t0, t2);
            b[2] = csub(t0, t2);
            b[1] = make_float2(t1.x + t3.y, t1.y - t3.x);
            b[3] = make_float2(t1.x - t3.y, t1.y + t3.x);
        } else if constexpr (R == 3) {
            const float2 t1 = cadd(a[1], a[2]), t2 = csub(a[1], a[2]);
            b[0] = cadd(a[0], t1);
            const float2 mm = make_float2(a[0].x - 0.5f * t1.x, a[0].y - 0.5f * t1.y);
            const float nr = kS3 * t2.y, ni = kS3 * t2.x;
            b[1] = make_float2(mm.x + nr, mm.y - ni);
            b[2] = make_float2(mm.x - nr, mm.y + ni);
        } else {
            const float2 t1 = cadd(a[1], a[4]), t2 = cadd(a[2], a[3]), t3 = csub(a[1], a[4]), t4 = csub(a[2], a[3]);
            b[0] = cadd(cadd(a[0], t1), t2);
            const float2 m1 = make_float2((a[0].x + kC51 * t1.x) + kC52 * t2.x, (a[0].y + kC51 * t1.y) + kC52 * t2.y);
            const float2 m2 = make_float2((a[0].x + kC52 * t1.x) + kC51 * t2.x, (a[0].y + kC52 * t1.y) + kC51 * t2.y);
            const float2 n1 = make_float2(kS51 * t3.x + kS52 * t4.x, kS51 * t3.y + kS52 * t4.y);
            const float2 n2 = make_float2(kS52 * t3.x - kS51 * t4.x, kS52 * t3.y - kS51 * t4.y);
            b[1] = make_float2(m1.x + n1.y, m1.y - n1.x);
            b[4] = make_float2(m1.x - n1.y, m1.y + n1.x);
            b[2] = make_float2(m2.x + n2.y, m2.y - n2.x);
            b[3] = make_float2(m2.x - n2.y, m2.y + n2.x);
        }
        out[0] = b[0];
#pragma unroll
        for (int k = 1; k < R; k++) out[s * k] = cmul(b[k], tw[p * k * tstep]);
    }
}

// Forward transform of nb lines in LDS; returns the buffer holding the result.  Entry and exit are barrier-separated.
// tw_lds: plan.n float2 of LDS that receive the twiddle table (every pass reads r - 1 twiddles per butterfly: from LDS
// that is ~64 cycles instead of an L2 round trip).
__device__ float2* fft_lds(float2* src, float2* dst, const vsp::Plan& plan, int nb, const float2* __restrict__ tw_global,
                           float2* __restrict__ tw) {
    int n_cur = plan.n, s = 1;
    for (int i = threadIdx.x; i < plan.n; i += kThreads) tw[i] = tw_global[i];
    __syncthreads();
    for (int pass = 0; pass < plan.passes; pass++) {
        const int r = plan.radix[pass], m = n_cur / r, tstep = plan.n / n_cur;
        switch (r) {
        case 2: fft_pass<2>(src, dst, plan.n, nb, m, s, tstep, tw); break;
        case 3: fft_pass<3>(src, dst, plan.n, nb, m, s, tstep, tw); break;
        case 4: fft_pass<4>(src, dst, plan.n, nb, m, s, tstep, tw); break;
        default: fft_pass<5>(src, dst, plan.n, nb, m, s, tstep, tw); break;
        }
        __syncthreads();
        float2* t = src; src = dst; dst = t;
        n_cur = m;
        s *= r;
    }
    return src;
}

extern __shared__ float2 lds_lines[];

// rows of the zero-padded float(level-2 image) -> columns 0..N/2 of the row spectrum.  grid (ceil(M/nb), frames)
__global__ __launch_bounds__(kThreads) void vs_k_phase_rows_fwd(const uint8_t* __restrict__ img, size_t img_frame, int w, int h,
                                                                int stride, vsp::Plan pn, int M, int nb,
                                                                const float2* __restrict__ tw, float2* __restrict__ spec,
                                                                size_t spec_frame) {
    const int N = pn.n, NC = N / 2 + 1;
    const int r0 = blockIdx.x * nb, lines = min(nb, M - r0);
    const uint8_t* im = img + (size_t)blockIdx.y * img_frame;
    float2* a = lds_lines;
    float2* b = lds_lines + (size_t)nb * N;
    for (int idx = threadIdx.x; idx < lines * N; idx += kThreads) {
        const int t = idx / N, c = idx - t * N, r = r0 + t;
        a[idx] = make_float2((r < h && c < w) ? (float)im[(size_t)r * stride + c] : 0.0f, 0.0f);
    }
    const float2* res = fft_lds(a, b, pn, lines, tw, lds_lines + (size_t)2 * nb * N);
    float2* out = spec + (size_t)blockIdx.y * spec_frame;
    for (int idx = threadIdx.x; idx < lines * NC; idx += kThreads) {
        const int t = idx / NC, c = idx - t * NC;
        out[(size_t)(r0 + t) * NC + c] = res[t * N + c];
    }
}

// columns of the half spectrum, in place.  grid (ceil(NC/nb), frames)
__global__ __launch_bounds__(kThreads) void vs_k_phase_cols_fwd(float2* __restrict__ spec, size_t spec_frame, vsp::Plan pm, int NC,
                                                                int nb, const float2* __restrict__ tw) {
    const int M = pm.n;
    const int c0 = blockIdx.x * nb, lines = min(nb, NC - c0);
    float2* sp = spec + (size_t)blockIdx.y * spec_frame;
    float2* a = lds_lines;
    float2* b = lds_lines + (size_t)nb * M;
    for (int idx = threadIdx.x; idx < lines * M; idx += kThreads) {
        const int r = idx / lines, t = idx - r * lines;
        a[t * M + r] = sp[(size_t)r * NC + c0 + t];
    }
    const float2* res = fft_lds(a, b, pm, lines, tw, lds_lines + (size_t)2 * nb * M);
    for (int idx = threadIdx.x; idx < lines * M; idx += kThreads) {
        const int r = idx / lines, t = idx - r * lines;
        sp[(size_t)r * NC + c0 + t] = res[t * M + r];
    }
}

// C = F_prev conj(F_cur) / |.|, then the inverse column transform (conj . forward . conj).  grid (ceil(NC/nb), pairs)
__global__ __launch_bounds__(kThreads) void vs_k_phase_cross_cols_inv(const float2* __restrict__ spec, size_t spec_frame,
                                                                      const vsp::Pair* __restrict__ pairs, vsp::Plan pm, int NC,
                                                                      int nb, const float2* __restrict__ tw,
                                                                      float2* __restrict__ G, size_t g_pair) {
    const int M = pm.n;
    const int c0 = blockIdx.x * nb, lines = min(nb, NC - c0);
    const vsp::Pair pr = pairs[blockIdx.y];
    const float2* Fa = spec + (size_t)pr.prev_slot * spec_frame;
    const float2* Fb = spec + (size_t)pr.cur_slot * spec_frame;
    float2* a = lds_lines;
    float2* b = lds_lines + (size_t)nb * M;
    for (int idx = threadIdx.x; idx < lines * M; idx += kThreads) {
        const int r = idx / lines, t = idx - r * lines;
        const float2 x = Fa[(size_t)r * NC + c0 + t], y = Fb[(size_t)r * NC + c0 + t];
        const float re = x.x * y.x + x.y * y.y;                    // mulSpectrums, conjB
        const float im = x.y * y.x - x.x * y.y;
        const float mag = (float)sqrt((double)re * (double)re + (double)im * (double)im);   // magSpectrums
        const double denom = (double)mag * (double)mag + (double)FLT_EPSILON;               // divSpectrums
        const float cre = (float)(((double)re * (double)mag) / denom);
        const float cim = (float)(((double)im * (double)mag) / denom);
        a[t * M + r] = make_float2(cre, -cim);
    }
    const float2* res = fft_lds(a, b, pm, lines, tw, lds_lines + (size_t)2 * nb * M);
    float2* g = G + (size_t)blockIdx.y * g_pair;
    for (int idx = threadIdx.x; idx < lines * M; idx += kThreads) {
        const int r = idx / lines, t = idx - r * lines;
        g[(size_t)r * NC + c0 + t] = cconj(res[t * M + r]);
    }
}

// (value, shifted linear index) of a surface sample; minMaxLoc on the fftShift-ed image = largest value, first in row-major
// order of the shifted image among equals
struct PeakCand { float v; int si; };
__device__ __forceinline__ bool cand_better(float v, int si, float bv, int bsi) { return v > bv || (v == bv && si < bsi); }

// block-wide best candidate (all threads call; result valid in thread 0)
__device__ __forceinline__ PeakCand block_best(float v, int si) {
    __shared__ float s_v[kThreads];
    __shared__ int s_i[kThreads];
    s_v[threadIdx.x] = v;
    s_i[threadIdx.x] = si;
    __syncthreads();
    for (int off = kThreads / 2; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) {
            const float ov = s_v[threadIdx.x + off];
            const int oi = s_i[threadIdx.x + off];
            if (cand_better(ov, oi, s_v[threadIdx.x], s_i[threadIdx.x])) { s_v[threadIdx.x] = ov; s_i[threadIdx.x] = oi; }
        }
        __syncthreads();
    }
    return PeakCand{s_v[0], s_i[0]};
}

// inverse row transform of the Hermitian extension, real part kept: the unshifted, unscaled surface; and the best sample of
// the workgroup's rows as a candidate for the peak search.  grid (ceil(M/nb), pairs)
__global__ __launch_bounds__(kThreads) void vs_k_phase_rows_inv(const float2* __restrict__ G, size_t g_pair, vsp::Plan pn, int M,
                                                                int nb, const float2* __restrict__ tw, float* __restrict__ surf,
                                                                size_t surf_pair, PeakCand* __restrict__ cands) {
    const int N = pn.n, NC = N / 2 + 1;
    const int r0 = blockIdx.x * nb, lines = min(nb, M - r0);
    const float2* g = G + (size_t)blockIdx.y * g_pair;
    float2* a = lds_lines;
    float2* b = lds_lines + (size_t)nb * N;
    for (int idx = threadIdx.x; idx < lines * N; idx += kThreads) {
        const int t = idx / N, c = idx - t * N;
        const float2* row = g + (size_t)(r0 + t) * NC;
        // X[c] for c <= N/2, conj(X[N-c]) above; stored conjugated for the inverse
        a[idx] = c < NC ? cconj(row[c]) : row[N - c];
    }
    const float2* res = fft_lds(a, b, pn, lines, tw, lds_lines + (size_t)2 * nb * N);
    float* out = surf + (size_t)blockIdx.y * surf_pair;
    const int hx = N / 2, hy = M / 2;
    float best = -INFINITY;
    int best_si = 0x7fffffff;
    for (int idx = threadIdx.x; idx < lines * N; idx += kThreads) {
        const int t = idx / N, c = idx - t * N;
        const float v = res[idx].x;
        out[(size_t)(r0 + t) * N + c] = v;
        int sy = r0 + t + hy; if (sy >= M) sy -= M;             // fftShift moves element i to (i + n/2) mod n
        int sx = c + hx; if (sx >= N) sx -= N;
        const int si = sy * N + sx;
        if (cand_better(v, si, best, best_si)) { best = v; best_si = si; }
    }
    const PeakCand w = block_best(best, best_si);
    if (threadIdx.x == 0) cands[(size_t)blockIdx.y * gridDim.x + blockIdx.x] = w;
}

// minMaxLoc over the row-block candidates + weightedCentroid(5x5) + response.  one workgroup per pair
__global__ __launch_bounds__(kThreads) void vs_k_phase_peak(const float* __restrict__ surf, size_t surf_pair, int M, int N,
                                                            const PeakCand* __restrict__ cands, int n_cands,
                                                            vsp::Result* __restrict__ results) {
    const float* sf = surf + (size_t)blockIdx.x * surf_pair;
    const int hx = N / 2, hy = M / 2;
    float best = -INFINITY;
    int best_i = 0x7fffffff;
    for (int i = threadIdx.x; i < n_cands; i += kThreads) {
        const PeakCand c = cands[(size_t)blockIdx.x * n_cands + i];
        if (cand_better(c.v, c.si, best, best_i)) { best = c.v; best_i = c.si; }
    }
    const PeakCand top = block_best(best, best_i);
    if (threadIdx.x == 0) {
        int pi = top.si;
        if (pi == 0x7fffffff) pi = 0;      // a surface without any comparable value (all NaN): minMaxLoc leaves (0, 0)
        const int py = pi / N, px = pi - py * N;
        int minr = py - 2, maxr = py + 2, minc = px - 2, maxc = px + 2;
        if (minr < 0) minr = 0;
        if (minc < 0) minc = 0;
        if (maxr > M - 1) maxr = M - 1;
        if (maxc > N - 1) maxc = N - 1;
        double cx = 0.0, cy = 0.0, sum = 0.0;
        for (int sy = minr; sy <= maxr; sy++) {
            int y = sy - hy; if (y < 0) y += M;
            for (int sx = minc; sx <= maxc; sx++) {
                int x = sx - hx; if (x < 0) x += N;
                const double v = (double)sf[(size_t)y * N + x];
                cx += (double)sx * v;
                cy += (double)sy * v;
                sum += v;
            }
        }
        vsp::Result r;
        r.response = sum / (double)((long long)M * N);
        sum += DBL_EPSILON;
        cx /= sum;
        cy /= sum;
        r.dx = (double)N / 2.0 - cx;
        r.dy = (double)M / 2.0 - cy;
        results[blockIdx.x] = r;
    }
}

}  // namespace

VS_BOUNDS_TU(vs_bounds_fetch_phase)

namespace vsp {

int optimal_dft_size(int n) {
    if (n < 1) return -1;
    for (int c = n;; c++) {
        int r = c;
        while (r % 2 == 0) r /= 2;
        while (r % 3 == 0) r /= 3;
        while (r % 5 == 0) r /= 5;
        if (r == 1) return c;
    }
}

bool make_plan(int n, Plan& p) {
    p.n = n;
    p.passes = 0;
    int r = n;
    while (r % 5 == 0) { p.radix[p.passes++] = 5; r /= 5; }
    while (r % 3 == 0) { p.radix[p.passes++] = 3; r /= 3; }
    while (r % 4 == 0) { p.radix[p.passes++] = 4; r /= 4; }
    if (r % 2 == 0) { p.radix[p.passes++] = 2; r /= 2; }
    return r == 1;
}

static hipError_t upload_twiddles(int n, float2** out, hipStream_t s) {
    std::vector<float2> tw((size_t)n);
    for (int j = 0; j < n; j++) {
        const double a = 2.0 * M_PI * (double)j / (double)n;
        tw[j] = make_float2((float)std::cos(a), (float)-std::sin(a));
    }
    hipError_t e = vsi::dev_alloc((void**)out, sizeof(float2) * (size_t)n);
    if (e != hipSuccess) return e;
    e = hipMemcpyAsync(*out, tw.data(), sizeof(float2) * (size_t)n, hipMemcpyHostToDevice, s);
    if (e != hipSuccess) return e;
    return hipStreamSynchronize(s);      // tw is a local
}

void Context::destroy() {
    if (twN) (void)hipFree(twN);
    if (twM) (void)hipFree(twM);
    if (cands) (void)hipFree(cands);
    cands = nullptr; cands_bytes = 0;
    twN = twM = nullptr;
    w = h = N = M = NC = 0;
}

hipError_t Context::configure(int width, int height, hipStream_t s) {
    if (width == w && height == h && twN) return hipSuccess;
    destroy();
    const int n = optimal_dft_size(width), m = optimal_dft_size(height);
    if (n < 1 || m < 1 || n > kMaxLine || m > kMaxLine) return hipErrorInvalidValue;
    if (!make_plan(n, pn) || !make_plan(m, pm)) return hipErrorInvalidValue;
    hipError_t e = upload_twiddles(n, &twN, s);
    if (e != hipSuccess) return e;
    e = upload_twiddles(m, &twM, s);
    if (e != hipSuccess) return e;
    w = width; h = height; N = n; M = m; NC = n / 2 + 1;
    return hipSuccess;
}

// lines per workgroup: small LDS footprints (several workgroups per CU) beat long ones, but the column transforms read
// `lines` adjacent columns of every row and want whole sectors.  Measured per 240 1080p frames (N = 480, M = 270):
// 8 / 8 lines 0.84 ms, 4 / 8 lines 0.69 ms, 2 / 3 lines 0.84 ms; per 120 4K frames (960, 540): 4 / 7 lines 2.00 ms,
// 2 / 4 lines 1.62 ms, 2 / 8 lines 1.69 ms, 1 / 1 line 2.64 ms.
static int lines_per_block(int n) { return std::max(1, std::min(n > 600 ? 2 : (n > 300 ? 4 : 8), kMaxLine / n)); }
// dynamic LDS of a transform kernel: two line buffers of nb lines + the twiddle table
static size_t lds_bytes(int nb, int n) { return sizeof(float2) * ((size_t)2 * nb + 1) * n; }
template <typename K>
static hipError_t allow_lds(K kernel, size_t bytes) {
    return hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}

hipError_t Context::spectra(const uint8_t* img, size_t img_frame, int stride, int n_frames, float2* spec, hipStream_t s) const {
    if (n_frames <= 0) return hipSuccess;
    const int nbr = lines_per_block(N), nbc = lines_per_block(M);
    hipError_t e = allow_lds(vs_k_phase_rows_fwd, lds_bytes(nbr, N));
    if (e != hipSuccess) return e;
    e = allow_lds(vs_k_phase_cols_fwd, lds_bytes(nbc, M));
    if (e != hipSuccess) return e;
    vs_k_phase_rows_fwd<<<dim3((M + nbr - 1) / nbr, n_frames), kThreads, lds_bytes(nbr, N), s>>>(
        img, img_frame, w, h, stride, pn, M, nbr, twN, spec, spec_frame());
    vs_k_phase_cols_fwd<<<dim3((NC + nbc - 1) / nbc, n_frames), kThreads, lds_bytes(nbc, M), s>>>(
        spec, spec_frame(), pm, NC, nbc, twM);
    return hipGetLastError();
}

hipError_t Context::correlate(const float2* spec, const Pair* pairs_dev, int n_pairs, float2* G, float* surf, Result* results_dev,
                              hipStream_t s) const {
    if (n_pairs <= 0) return hipSuccess;
    const int nbr = lines_per_block(N), nbc = lines_per_block(M);
    const int row_blocks = (M + nbr - 1) / nbr;
    const size_t need = (size_t)n_pairs * row_blocks * sizeof(PeakCand);
    if (need > cands_bytes) {                  // per-row-block peak candidates (grown on demand; the stream is in order)
        if (cands) { hipError_t es = hipStreamSynchronize(s); if (es != hipSuccess) return es; (void)hipFree(cands); }
        cands = nullptr; cands_bytes = 0;
        hipError_t em = vsi::dev_alloc(&cands, need);
        if (em != hipSuccess) return em;
        cands_bytes = need;
    }
    hipError_t e = allow_lds(vs_k_phase_cross_cols_inv, lds_bytes(nbc, M));
    if (e != hipSuccess) return e;
    e = allow_lds(vs_k_phase_rows_inv, lds_bytes(nbr, N));
    if (e != hipSuccess) return e;
    vs_k_phase_cross_cols_inv<<<dim3((NC + nbc - 1) / nbc, n_pairs), kThreads, lds_bytes(nbc, M), s>>>(
        spec, spec_frame(), pairs_dev, pm, NC, nbc, twM, G, spec_frame());
    vs_k_phase_rows_inv<<<dim3(row_blocks, n_pairs), kThreads, lds_bytes(nbr, N), s>>>(
        G, spec_frame(), pn, M, nbr, twN, surf, surface_elems(), (PeakCand*)cands);
    vs_k_phase_peak<<<n_pairs, kThreads, 0, s>>>(surf, surface_elems(), M, N, (const PeakCand*)cands, row_blocks, results_dev);
    return hipGetLastError();
}

}  // namespace vsp
