// halide_abi_test.cpp -- the link-level drop-in (include/vs_halide_abi.h, libvs_halide_abi.so) exercised the way the reference
// uses its generated pipelines: seven wrappers with the reference's names and call shapes (imgproc.hpp:8-32, 67-95 /
// imgproc.cpp:26-202: reallocate outputs on a shape mismatch, convert the centre-based transform, choose the tile size, call
// the AOT symbol with buffers, return r == 0) on a minimal owning buffer class that converts to halide_buffer_t* like
// Halide::Runtime::Buffer does.  Everything here is written against this repository's own struct; the AOT calls below are
// the sixteen C symbols the reference's imgproc.cpp links against.  Outputs are compared byte for byte with the CPU
// restatement (oracle/: test infrastructure).
//   usage: halide_abi_test args   -> argument validation only (no device needed)
//          halide_abi_test gpu    -> + the seven wrappers against the oracle
#include <cmath>
#include <cstdio>
#include <cstring>
#include <random>
#include <string>
#include <vector>

#include "../../include/vs_halide_abi.h"
#include "../../include/vs_amd.h"
#include "../../oracle/vs_oracle.h"

static int fails = 0;
#define CHECK(cond) do { if (!(cond)) { std::printf("[FAIL] %s:%d %s\n", __FILE__, __LINE__, #cond); fails++; } } while (0)

template <typename T> struct TypeOf;
template <> struct TypeOf<uint8_t> { static vs_halide_type_t get() { return {VS_HALIDE_TYPE_UINT, 8, 1}; } };
template <> struct TypeOf<uint16_t> { static vs_halide_type_t get() { return {VS_HALIDE_TYPE_UINT, 16, 1}; } };
template <> struct TypeOf<float> { static vs_halide_type_t get() { return {VS_HALIDE_TYPE_FLOAT, 32, 1}; } };
template <> struct TypeOf<double> { static vs_halide_type_t get() { return {VS_HALIDE_TYPE_FLOAT, 64, 1}; } };

// dense, owning, planar (dim 0 innermost) -- the subset of Halide::Runtime::Buffer<T> the wrappers use
template <typename T>
class Buffer {
public:
    Buffer() { init({}); }
    explicit Buffer(int e0) { init({e0}); }
    Buffer(int e0, int e1) { init({e0, e1}); }
    Buffer(int e0, int e1, int e2) { init({e0, e1, e2}); }
    Buffer(const Buffer& o) { *this = o; }
    Buffer& operator=(const Buffer& o) {
        store = o.store; dims = o.dims; buf = o.buf; buf.dim = dims.data(); buf.host = (uint8_t*)store.data();
        return *this;
    }
    struct Dim { int e; int extent() const { return e; } };
    int dimensions() const { return buf.dimensions; }
    Dim dim(int i) const { return Dim{dims[i].extent}; }
    int width() const { return dims.size() > 0 ? dims[0].extent : 1; }
    int height() const { return dims.size() > 1 ? dims[1].extent : 1; }
    T* data() { return store.data(); }
    const T* data() const { return store.data(); }
    size_t size() const { return store.size(); }
    operator vs_halide_buffer_t*() { return &buf; }
    vs_halide_buffer_t* raw() { return &buf; }
private:
    void init(std::initializer_list<int> e) {
        dims.clear();
        int s = 1;
        for (int x : e) { dims.push_back(vs_halide_dimension_t{0, x, s, 0}); s *= x; }
        store.assign(e.size() ? (size_t)s : 0, T());
        std::memset(&buf, 0, sizeof(buf));
        buf.host = (uint8_t*)store.data(); buf.type = TypeOf<T>::get(); buf.dimensions = (int)dims.size(); buf.dim = dims.data();
    }
    std::vector<T> store;
    std::vector<vs_halide_dimension_t> dims;
    vs_halide_buffer_t buf;
};

struct SimilarityTransform { double A = 0, B = 0, TX = 0, TY = 0; };     // imgproc.hpp:40-46

// ---- the seven wrappers, shaped as imgproc.cpp:26-202 -------------------------------------------------------------
bool SparseJacobian(Buffer<float>& grad_x, Buffer<float>& grad_y, Buffer<uint16_t>& local_max_x, Buffer<uint16_t>& local_max_y,
                    Buffer<float>& output_x, Buffer<float>& output_y) {
    if (output_x.dimensions() != 3 || output_x.dim(0).extent() != local_max_x.dim(0).extent() ||
        output_x.dim(1).extent() != local_max_x.dim(1).extent()) {
        output_x = Buffer<float>(local_max_x.dim(0).extent(), local_max_x.dim(1).extent(), 4);
        output_y = Buffer<float>(local_max_x.dim(0).extent(), local_max_x.dim(1).extent(), 4);
    }
    return sparse_jac(grad_x, grad_y, local_max_x, local_max_y, output_x, output_y) == 0;
}
bool SparseICA(Buffer<uint8_t>& input_template, Buffer<uint8_t>& input_keyframe, Buffer<uint16_t>& selected_pixels_x,
               Buffer<uint16_t>& selected_pixels_y, Buffer<float>& selected_jacobians_x, Buffer<float>& selected_jacobians_y,
               const SimilarityTransform& transform, Buffer<double>& output) {
    if (output.dimensions() != 1 || output.dim(0).extent() != 4) output = Buffer<double>(4);
    return sparse_ica(input_template, input_keyframe, selected_pixels_x, selected_pixels_y, selected_jacobians_x, selected_jacobians_y,
                      static_cast<float>(transform.A), static_cast<float>(transform.B),
                      static_cast<float>(transform.TX - transform.A * (input_template.width() * 0.5f) + transform.B * (input_template.height() * 0.5f)),
                      static_cast<float>(transform.TY - transform.B * (input_template.width() * 0.5f) - transform.A * (input_template.height() * 0.5f)),
                      output) == 0;
}
bool SparseWarpDiff(Buffer<uint8_t>& input_template, Buffer<uint8_t>& input_keyframe, Buffer<uint16_t>& local_max,
                    const SimilarityTransform& transform, Buffer<uint16_t>& output) {
    if (output.dimensions() != 2 || output.dim(0).extent() != local_max.dim(0).extent() || output.dim(1).extent() != local_max.dim(1).extent())
        output = Buffer<uint16_t>(local_max.dim(0).extent(), local_max.dim(1).extent());
    return sparse_warpdiff(input_template, input_keyframe, local_max, static_cast<float>(transform.A), static_cast<float>(transform.B),
                           static_cast<float>(transform.TX - transform.A * (input_template.width() * 0.5f) + transform.B * (input_template.height() * 0.5f)),
                           static_cast<float>(transform.TY - transform.B * (input_template.width() * 0.5f) - transform.A * (input_template.height() * 0.5f)),
                           output) == 0;
}
bool PyrDown(Buffer<uint8_t>& input, Buffer<uint8_t>& output) { return pyr_down(input, output) == 0; }
bool ImageWarp(Buffer<uint8_t>& input, const SimilarityTransform& transform, Buffer<float>& output) {
    double cx = (input.width() - 1) * 0.5, cy = (input.height() - 1) * 0.5;
    double tx_ul = transform.TX - transform.A * cx + transform.B * cy, ty_ul = transform.TY - transform.B * cx - transform.A * cy;
    return image_warp(input, transform.A, transform.B, tx_ul, ty_ul, output) == 0;
}
bool GradXY(Buffer<uint8_t>& input, Buffer<float>& output_x, Buffer<float>& output_y) { return grad_xy(input, output_x, output_y) == 0; }
bool GradArgMax(Buffer<float>& grad_x, Buffer<float>& grad_y, int& tile_size, Buffer<uint16_t>& local_max_x, Buffer<uint16_t>& local_max_y) {
    const int min_tiles = 1000, max_tile_size = 20;
    tile_size = 2;
    for (int i = 4; i <= max_tile_size; i += 2) {
        if ((grad_x.width() / i) * (grad_y.height() / i) < min_tiles) break;
        tile_size = i;
    }
    int width_tiles = grad_x.width() / tile_size, height_tiles = grad_y.height() / tile_size;
    if (local_max_x.dimensions() != 3 || local_max_x.dim(0).extent() != width_tiles || local_max_x.dim(1).extent() != height_tiles) {
        local_max_x = Buffer<uint16_t>(width_tiles, height_tiles, 2);
        local_max_y = Buffer<uint16_t>(width_tiles, height_tiles, 2);
    }
    int r = -1;
    if (tile_size >= 19) r = grad_argmax_20(grad_x, grad_y, local_max_x, local_max_y);
    else if (tile_size >= 17) r = grad_argmax_18(grad_x, grad_y, local_max_x, local_max_y);
    else if (tile_size >= 15) r = grad_argmax_16(grad_x, grad_y, local_max_x, local_max_y);
    else if (tile_size >= 13) r = grad_argmax_14(grad_x, grad_y, local_max_x, local_max_y);
    else if (tile_size >= 11) r = grad_argmax_12(grad_x, grad_y, local_max_x, local_max_y);
    else if (tile_size >= 9) r = grad_argmax_10(grad_x, grad_y, local_max_x, local_max_y);
    else if (tile_size >= 7) r = grad_argmax_8(grad_x, grad_y, local_max_x, local_max_y);
    else if (tile_size >= 5) r = grad_argmax_6(grad_x, grad_y, local_max_x, local_max_y);
    else if (tile_size >= 3) r = grad_argmax_4(grad_x, grad_y, local_max_x, local_max_y);
    else r = grad_argmax_2(grad_x, grad_y, local_max_x, local_max_y);
    return r == 0;
}

// ---- argument validation: Halide's error codes, no device touched ---------------------------------------------------
static void TestArguments() {
    Buffer<uint8_t> in(64, 48), out(32, 24);
    Buffer<float> gx(64, 48), gy(64, 48);
    CHECK(pyr_down(nullptr, out) == VS_HALIDE_ERR_BUFFER_NULL);
    CHECK(grad_xy(in, gx, nullptr) == VS_HALIDE_ERR_BUFFER_NULL);
    {   // a bounds-query style call (host == NULL) is refused, not dereferenced
        Buffer<uint8_t> q(64, 48);
        q.raw()->host = nullptr;
        CHECK(pyr_down(q, out) == VS_HALIDE_ERR_HOST_NULL);
    }
    {   // element type and rank are the generator's (generators.cpp:59-61, 205-208)
        Buffer<float> wrong(32, 24);
        CHECK(pyr_down(in, wrong) == VS_HALIDE_ERR_BAD_TYPE);
        Buffer<uint8_t> three(32, 24, 1);
        CHECK(pyr_down(in, three) == VS_HALIDE_ERR_BAD_DIMENSIONS);
        Buffer<uint8_t> lanes(32, 24);
        lanes.raw()->type.lanes = 4;
        CHECK(pyr_down(in, lanes) == VS_HALIDE_ERR_BAD_TYPE);
    }
    {   // a cropped buffer (non-zero min), a transposed one (stride of dim 0 != 1), a padded float plane
        Buffer<uint8_t> crop(32, 24);
        crop.raw()->dim[0].min = 4;
        CHECK(pyr_down(in, crop) == VS_HALIDE_ERR_CONSTRAINT);
        Buffer<uint8_t> tr(32, 24);
        tr.raw()->dim[0].stride = 24; tr.raw()->dim[1].stride = 1;
        CHECK(pyr_down(in, tr) == VS_HALIDE_ERR_CONSTRAINT);
        Buffer<float> padded(64, 48);
        padded.raw()->dim[1].stride = 80;
        CHECK(grad_xy(in, padded, gy) == VS_HALIDE_ERR_CONSTRAINT);
    }
    {   // extents that do not fit together
        Buffer<uint8_t> big(40, 24);
        CHECK(pyr_down(in, big) == VS_HALIDE_ERR_OUT_OF_BOUNDS);
        Buffer<float> small(60, 48);
        CHECK(grad_xy(in, small, gy) == VS_HALIDE_ERR_OUT_OF_BOUNDS);
        Buffer<uint16_t> lmx(16, 12, 2), lmy(16, 12, 2), lm3(16, 12, 3);
        CHECK(grad_argmax_4(gx, gy, lmx, lm3) == VS_HALIDE_ERR_OUT_OF_BOUNDS);
        CHECK(grad_argmax_2(gx, gy, lmx, lmy) == VS_HALIDE_ERR_OUT_OF_BOUNDS);       // 64 / 2 x 48 / 2 tiles expected
        Buffer<double> out5(5);
        Buffer<uint16_t> sx(10, 2), sy(10, 2);
        Buffer<float> jx(10, 4), jy(10, 4);
        CHECK(sparse_ica(in, in, sx, sy, jx, jy, 0.f, 0.f, 0.f, 0.f, out5) == VS_HALIDE_ERR_OUT_OF_BOUNDS);
    }
}

// ---- the wrappers against the oracle ----------------------------------------------------------------------------------
static void fill_image(Buffer<uint8_t>& b, unsigned seed, int shift = 0) {
    // smooth blobs + edges so that every tile has a gradient maximum (random noise would do, but ties are rarer this way)
    std::mt19937 rng(seed);
    const int w = b.width(), h = b.height();
    std::vector<float> f((size_t)w * h, 0.f);
    for (int k = 0; k < 60; k++) {
        int x0 = (int)(rng() % w), y0 = (int)(rng() % h), rw = 4 + (int)(rng() % 40), rh = 4 + (int)(rng() % 40);
        float v = (float)(rng() % 200) - 60.f;
        for (int y = std::max(0, y0); y < std::min(h, y0 + rh); y++)
            for (int x = std::max(0, x0 + shift); x < std::min(w, x0 + rw + shift); x++) f[(size_t)y * w + x] += v;
    }
    for (int i = 0; i < w * h; i++) b.data()[i] = (uint8_t)std::min(255.f, std::max(0.f, 90.f + f[i] + (float)(rng() % 7)));
}
template <typename T> static bool same(const Buffer<T>& a, const std::vector<T>& b) {
    return a.size() == b.size() && std::memcmp(a.data(), b.data(), b.size() * sizeof(T)) == 0;
}

static void TestWrappers(int w, int h) {
    Buffer<uint8_t> key(w, h), tmpl(w, h);
    fill_image(key, 7u + w);
    fill_image(tmpl, 7u + w, 2);
    // PyrDown
    Buffer<uint8_t> half(w / 2, h / 2);
    CHECK(PyrDown(key, half));
    std::vector<uint8_t> ref_half((size_t)(w / 2) * (h / 2));
    vso_pyr_down(key.data(), w, h, w, ref_half.data(), w / 2, h / 2, w / 2);
    CHECK(same(half, ref_half));
    // GradXY
    Buffer<float> gx(w, h), gy(w, h);
    CHECK(GradXY(key, gx, gy));
    std::vector<float> rgx((size_t)w * h), rgy((size_t)w * h);
    vso_grad_xy(key.data(), w, h, w, rgx.data(), rgy.data());
    CHECK(same(gx, rgx) && same(gy, rgy));
    // GradArgMax (outputs allocated by the wrapper, tile size chosen by the wrapper)
    int ts = 0;
    Buffer<uint16_t> lmx, lmy;
    CHECK(GradArgMax(gx, gy, ts, lmx, lmy));
    CHECK(ts == vso_tile_size(w, h));
    const int tx = w / ts, ty = h / ts;
    std::vector<uint16_t> rlx((size_t)tx * ty * 2), rly((size_t)tx * ty * 2);
    vso_grad_argmax(rgx.data(), rgy.data(), w, h, ts, rlx.data(), rly.data());
    CHECK(same(lmx, rlx) && same(lmy, rly));
    // SparseJacobian (outputs allocated by the wrapper)
    Buffer<float> jx, jy;
    CHECK(SparseJacobian(gx, gy, lmx, lmy, jx, jy));
    std::vector<float> rjx((size_t)tx * ty * 4), rjy((size_t)tx * ty * 4);
    vso_sparse_jac(rgx.data(), rgy.data(), w, h, rlx.data(), rly.data(), tx, ty, rjx.data(), rjy.data());
    CHECK(same(jx, rjx) && same(jy, rjy));
    // SparseWarpDiff with a centre-based transform (the wrapper converts it as imgproc.cpp:98-103 does)
    SimilarityTransform T;
    T.A = 0.004; T.B = -0.003; T.TX = 1.75; T.TY = -0.6;
    Buffer<uint16_t> wd;
    CHECK(SparseWarpDiff(tmpl, key, lmx, T, wd));
    vso_transform vt{T.A, T.B, T.TX, T.TY};
    float p[4];
    vso_ul_params_sparse(&vt, w, h, p);
    std::vector<uint16_t> rwd((size_t)tx * ty);
    vso_sparse_warpdiff(tmpl.data(), key.data(), w, h, w, rlx.data(), tx, ty, p[0], p[1], p[2], p[3], rwd.data());
    CHECK(same(wd, rwd));
    // SparseICA on the first n tiles of each set (planar (n, 2) / (n, 4) as alignment.cpp:526-545 lays them out)
    const int n = std::min(tx * ty, 700);
    Buffer<uint16_t> sx(n, 2), sy(n, 2);
    Buffer<float> sjx(n, 4), sjy(n, 4);
    for (int i = 0; i < n; i++) {
        for (int c = 0; c < 2; c++) { sx.data()[c * n + i] = rlx[(size_t)c * tx * ty + i]; sy.data()[c * n + i] = rly[(size_t)c * tx * ty + i]; }
        for (int c = 0; c < 4; c++) { sjx.data()[c * n + i] = rjx[(size_t)c * tx * ty + i]; sjy.data()[c * n + i] = rjy[(size_t)c * tx * ty + i]; }
    }
    Buffer<double> b;                                                    // allocated by the wrapper
    CHECK(SparseICA(tmpl, key, sx, sy, sjx, sjy, T, b));
    double rb[4];
    vso_sparse_ica(tmpl.data(), key.data(), w, h, w, sx.data(), n, sy.data(), n, sjx.data(), sjy.data(), p[0], p[1], p[2], p[3], rb);
    for (int c = 0; c < 4; c++) CHECK(std::fabs(b.data()[c] - rb[c]) <= 1e-12 * (std::fabs(rb[c]) + 1.0));   // fp64 tree vs serial sum
    // ImageWarp
    Buffer<float> warped(w, h);
    CHECK(ImageWarp(key, T, warped));
    float q[4];
    vso_ul_params_warp(&vt, w, h, q);
    std::vector<float> rw((size_t)w * h);
    vso_image_warp(key.data(), w, h, w, q[0], q[1], q[2], q[3], rw.data(), w, h);
    CHECK(same(warped, rw));
}

int main(int argc, char** argv) {
    const std::string mode = argc > 1 ? argv[1] : "args";
    TestArguments();
    if (mode == "gpu") {
        if (vs_device_count() < 1) { std::printf("no HIP device\n"); return 2; }
        TestWrappers(640, 480);      // tile size 16
        TestWrappers(322, 246);      // odd sizes: remainder pixels, tile size 8
        TestWrappers(96, 80);        // tile size 2
    }
    if (fails == 0) std::printf("ALL PASS (%s)\n", mode.c_str());
    return fails ? 1 : 0;
}
