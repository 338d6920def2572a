// vs_host.cpp -- the host-only part of the C ABI: scalar algebra the reference also keeps on the
// CPU (SimilarityTransform, tile-size rule, L1 smoother).  No device calls in this file.
#include "vs_internal.hpp"

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <exception>
#include <new>
#include <vector>

namespace vsi {
static thread_local char g_err[512] = "";
int set_error(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}
int caught() noexcept {
    try { throw; }
    catch (const std::bad_alloc&) { return set_error(VS_ERR_NOMEM, "host allocation failed (std::bad_alloc)"); }
    catch (const std::exception& e) { return set_error(VS_ERR_NOMEM, "C++ exception stopped at the C boundary: %s", e.what()); }
    catch (...) { return set_error(VS_ERR_NOMEM, "unknown C++ exception stopped at the C boundary"); }
}
}  // namespace vsi

extern "C" {

const char* vs_last_error(void) { return vsi::g_err; }
const char* vs_version(void) { return "video_stabilizer_amd 0.5 (gfx950, ABI 5)"; }
int vs_abi_version(void) { return VS_ABI_VERSION; }
size_t vs_sizeof_align_info(void) { return sizeof(vs_align_info); }

// alignment.hpp:5-41
int vs_format_bits(int format) try {
    switch (format) {
        case VS_FMT_GRAY8: case VS_FMT_BGR8: return 8;
        case VS_FMT_BGR10: return 10;
        case VS_FMT_BGR12: return 12;
        case VS_FMT_BGR16_FULL: return 16;
        default: return 0;
    }
} VS_CATCH_ALL
int vs_format_max_value(int format) try {
    const int b = vs_format_bits(format);
    return b ? (1 << b) - 1 : 0;
} VS_CATCH_ALL

void vs_aligner_params_default(vs_aligner_params* p) {
    p->phase_correlate = 0;
    p->phase_correlate_threshold = 0.5;
    p->threshold = 0.02;
    p->smallest_fraction = 0.8f;
    p->max_iters = 64;
    p->pyramid_min_width = 20;
    p->pyramid_min_height = 20;
    p->max_displacement = 10.0;
}
// stabilizer.hpp:13-30
void vs_stabilizer_params_default(vs_stabilizer_params* p) {
    vs_aligner_params_default(&p->aligner);
    p->lag = 10;
    p->smoother_memory = 5;
    p->lambda = 4.0;
    p->enable_smoother = 1;
    p->crop_pixels = 32;
    p->min_disp = 48.0;
    p->max_disp = 64.0;
    p->min_decay = 0.9;
    p->max_decay = 0.7;
    // the reference warps with cv::warpAffine(INTER_LINEAR, BORDER_CONSTANT) (stabilizer.cpp:97-99 -> imgproc.cpp:472-481): a drop-in
    // user gets that arithmetic by default (OpenCV's fixed-point bilinear restated: VS_WARP_BILINEAR_CV); VS_WARP_BILINEAR (the Halide
    // sampler's float lerp) and the Lanczos2 family (the north star's bgr_image_warp) are one field away
    p->warp_mode = VS_WARP_BILINEAR_CV;
    p->warp_border = VS_BORDER_CONSTANT;
}

// imgproc.cpp:333-359
vs_transform vs_transform_inverse(const vs_transform* t) {
    const double p = 1.0 + t->A, q = t->B;
    const double denom = p * p + q * q;
    vs_transform inv;
    inv.A = (p / denom) - 1.0;
    inv.B = -q / denom;
    inv.TX = (-p * t->TX - q * t->TY) / denom;
    inv.TY = (q * t->TX - p * t->TY) / denom;
    return inv;
}

// imgproc.cpp:361-387: result(p) = t2(t1(p))
vs_transform vs_transform_compose(const vs_transform* t1, const vs_transform* t2) {
    const double p1 = 1.0 + t1->A, q1 = t1->B;
    const double p2 = 1.0 + t2->A, q2 = t2->B;
    vs_transform t3;
    t3.A = (p2 * p1 - q2 * q1) - 1.0;
    t3.B = (p2 * q1 + q2 * p1);
    t3.TX = p2 * t1->TX - q2 * t1->TY + t2->TX;
    t3.TY = q2 * t1->TX + p2 * t1->TY + t2->TY;
    return t3;
}

// imgproc.cpp:389-394
vs_point vs_transform_warp(const vs_transform* t, vs_point p) {
    vs_point o;
    o.x = (1 + t->A) * p.x - t->B * p.y + t->TX;
    o.y = t->B * p.x + (1 + t->A) * p.y + t->TY;
    return o;
}

// imgproc.cpp:401-411
vs_point vs_transform_warp_center(const vs_transform* t, vs_point p, double cx, double cy) {
    const double px = p.x - cx, py = p.y - cy;
    vs_point o;
    o.x = (1 + t->A) * px - t->B * py + cx + t->TX;
    o.y = t->B * px + (1 + t->A) * py + cy + t->TY;
    return o;
}

// imgproc.cpp:419-437
double vs_transform_max_corner_displacement(const vs_transform* t, double width, double height) {
    const double cx = width * 0.5, cy = height * 0.5;
    const vs_point corners[4] = {{0.0, 0.0}, {width, 0.0}, {0.0, height}, {width, height}};
    double max_d = 0.0;
    for (const vs_point& c : corners) {
        vs_point m = vs_transform_warp_center(t, c, cx, cy);
        max_d = std::max(max_d, std::sqrt((m.x - c.x) * (m.x - c.x) + (m.y - c.y) * (m.y - c.y)));
    }
    return max_d;
}

// imgproc.cpp:151-162
int vs_tile_size(int w, int h) try {
    int tile_size = 2;
    for (int i = 4; i <= 20; i += 2) {
        if ((w / i) * (h / i) < 1000) break;
        tile_size = i;
    }
    return tile_size;
} VS_CATCH_ALL

// imgproc.cpp:69-75 / 98-103
void vs_ul_params_sparse(const vs_transform* t, int w, int h, float out4[4]) {
    out4[0] = static_cast<float>(t->A);
    out4[1] = static_cast<float>(t->B);
    out4[2] = static_cast<float>(t->TX - t->A * (w * 0.5f) + t->B * (h * 0.5f));
    out4[3] = static_cast<float>(t->TY - t->B * (w * 0.5f) - t->A * (h * 0.5f));
}
// imgproc.cpp:125-131
void vs_ul_params_warp(const vs_transform* t, int w, int h, float out4[4]) {
    const double cx = (w - 1) * 0.5, cy = (h - 1) * 0.5;
    out4[0] = static_cast<float>(t->A);
    out4[1] = static_cast<float>(t->B);
    out4[2] = static_cast<float>(t->TX - t->A * cx + t->B * cy);
    out4[3] = static_cast<float>(t->TY - t->B * cx - t->A * cy);
}

// imgproc.cpp:457-466 (the forward matrix of warpBySimilarityTransform) + cv::warpAffine's own inversion of a matrix given without
// WARP_INVERSE_MAP (OpenCV 4.x imgwarp.cpp: D = M0 M4 - M1 M3; D = D != 0 ? 1./D : 0; ... -- restated from the published source, in its
// operation order; this file compiles with -ffp-contract=off, so every product and sum rounds where OpenCV's x86-64 baseline build rounds)
void vs_cv_inverse_matrix(const vs_transform* t, int w, int h, double M[6]) {
    const double cx = (w - 1) * 0.5, cy = (h - 1) * 0.5;
    const double tx_ul = t->TX - t->A * cx + t->B * cy;
    const double ty_ul = t->TY - t->B * cx - t->A * cy;
    M[0] = 1.0 + t->A; M[1] = -t->B; M[2] = tx_ul;
    M[3] = t->B;       M[4] = 1.0 + t->A; M[5] = ty_ul;
    double D = M[0] * M[4] - M[1] * M[3];
    D = D != 0 ? 1. / D : 0;
    const double A11 = M[4] * D, A22 = M[0] * D;
    M[0] = A11; M[1] *= -D;
    M[3] *= -D; M[4] = A22;
    const double b1 = -M[0] * M[2] - M[1] * M[5];
    const double b2 = -M[3] * M[2] - M[4] * M[5];
    M[2] = b1; M[5] = b2;
}

// smoother.cpp:18-65
void vs_tvl1_smooth(const double* data, int n, double lambda, int iterations, double* x) {
    if (n <= 0) return;
    std::copy(data, data + n, x);
    for (int it = 0; it < iterations; ++it) {
        for (int i = 0; i < n; i++) x[i] = (1.0 - 0.5) * x[i] + 0.5 * data[i];
        for (int i = 0; i + 1 < n; i++) {
            const double diff = x[i + 1] - x[i];
            const double mag = std::fabs(diff);
            if (mag > lambda) {
                const double shrink = (mag - lambda) / mag * 0.5;
                x[i] += diff * shrink;
                x[i + 1] -= diff * shrink;
            } else {
                const double mid = 0.5 * (x[i] + x[i + 1]);
                x[i] = mid;
                x[i + 1] = mid;
            }
        }
    }
}

}  // extern "C"

// smoother.cpp:67-127
struct vs_smoother {
    int lag_behind, lag_ahead;
    double lambda;
    int next_to_finalize = 0;
    std::vector<vs_transform> measurements;
};

extern "C" {

vs_smoother* vs_smoother_create(int lag_behind, int lag_ahead, double lambda) try {
    vs_smoother* s = new vs_smoother();
    s->lag_behind = lag_behind;
    s->lag_ahead = lag_ahead;
    s->lambda = lambda;
    return s;
} VS_CATCH_ALL_NULL
void vs_smoother_destroy(vs_smoother* s) { delete s; }

int vs_smoother_update(vs_smoother* s, const vs_transform* meas, vs_transform* out_finalized) try {
    s->measurements.push_back(*meas);
    const int newest = (int)s->measurements.size() - 1;
    if (s->next_to_finalize + s->lag_ahead > newest) return 0;
    const int start = std::max(0, s->next_to_finalize - s->lag_behind);
    const int end = s->next_to_finalize + s->lag_ahead;
    const int n = end - start + 1;
    // The four parameters are smoothed independently (smoother.cpp:98-116) by the same sweep: run them as the four lanes of one
    // vector.  Every lane performs exactly vs_tvl1_smooth's IEEE operations in its order (the branch becomes a per-lane select; no
    // contraction: -ffp-contract=off), so the result is the scalar routine's bit for bit -- the sweep is a dependent chain along the
    // window, and four chains in flight cost what one did (13 -> ~4 us per frame: with 100 iterations per frame this was the host
    // side's largest item, a third of a stabilizer batch at 1080p).
    constexpr int kMaxWin = 64;
    const int middle = s->next_to_finalize - start;
    if (n <= kMaxWin) {
        double data[kMaxWin][4], x[kMaxWin][4];
        for (int i = 0; i < n; i++) {
            const vs_transform& m = s->measurements[start + i];
            data[i][0] = m.A; data[i][1] = m.B; data[i][2] = m.TX; data[i][3] = m.TY;
            for (int k = 0; k < 4; k++) x[i][k] = data[i][k];
        }
        const double lam = s->lambda;
        for (int it = 0; it < 100; ++it) {
            // (element i+1's relaxation toward the data is folded into the sweep, ahead of the first edge that reads it, and the
            //  running right-hand value is carried in a register: same operations on the same values, a shorter dependent chain)
            double cur[4];
            for (int k = 0; k < 4; k++) cur[k] = (1.0 - 0.5) * x[0][k] + 0.5 * data[0][k];
            for (int i = 0; i + 1 < n; i++) {
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const double left = cur[k];
                    const double right = (1.0 - 0.5) * x[i + 1][k] + 0.5 * data[i + 1][k];
                    const double diff = right - left;
                    const double mag = std::fabs(diff);
                    if (mag > lam) {
                        const double shrink = (mag - lam) / mag * 0.5;
                        x[i][k] = left + diff * shrink;
                        cur[k] = right - diff * shrink;
                    } else {
                        const double mid = 0.5 * (left + right);
                        x[i][k] = mid;
                        cur[k] = mid;
                    }
                }
            }
            for (int k = 0; k < 4; k++) x[n - 1][k] = cur[k];
        }
        out_finalized->A = x[middle][0]; out_finalized->B = x[middle][1]; out_finalized->TX = x[middle][2]; out_finalized->TY = x[middle][3];
    } else {
        std::vector<double> in(4 * n), out(4 * n);
        for (int i = 0; i < n; i++) {
            const vs_transform& m = s->measurements[start + i];
            in[i] = m.A; in[n + i] = m.B; in[2 * n + i] = m.TX; in[3 * n + i] = m.TY;
        }
        for (int k = 0; k < 4; k++) vs_tvl1_smooth(&in[k * n], n, s->lambda, 100, &out[k * n]);
        out_finalized->A = out[middle];
        out_finalized->B = out[n + middle];
        out_finalized->TX = out[2 * n + middle];
        out_finalized->TY = out[3 * n + middle];
    }
    s->next_to_finalize++;
    return 1;
} VS_CATCH_ALL

}  // extern "C"
