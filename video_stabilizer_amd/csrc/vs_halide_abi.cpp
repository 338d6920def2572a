// vs_halide_abi.cpp -- the sixteen Halide AOT symbols of the reference (imgproc.cpp:9-24) on the HIP kernels of libvs_amd.so.
// Plain host C++: argument validation + one vs_* call each (include/vs_halide_abi.h has the contract).  Built into its own
// library, libvs_halide_abi.so, because the symbol names are the reference's generic ones (pyr_down, image_warp, ...).
#include <initializer_list>

#include "../../include/vs_halide_abi.h"
#include "../../include/vs_amd.h"

namespace {

struct View {          // a validated buffer: dense element strides in dim 0, rows possibly padded
    uint8_t* host;
    int extent[4];
    int stride[4];
};

// Halide's own argument checks, in the order the generated code makes them: null buffer, null host, element type, rank
int check(const vs_halide_buffer_t* b, int code, int bits, int dims, View& v) {
    if (!b) return VS_HALIDE_ERR_BUFFER_NULL;
    if (!b->host) return VS_HALIDE_ERR_HOST_NULL;        // includes bounds-query calls, which this library does not serve
    if (b->type.code != code || b->type.bits != bits || b->type.lanes != 1) return VS_HALIDE_ERR_BAD_TYPE;
    if (b->dimensions != dims || !b->dim) return VS_HALIDE_ERR_BAD_DIMENSIONS;
    v.host = b->host;
    for (int i = 0; i < 4; i++) { v.extent[i] = 1; v.stride[i] = 0; }
    for (int i = 0; i < dims; i++) {
        if (b->dim[i].min != 0 || b->dim[i].extent < 0) return VS_HALIDE_ERR_CONSTRAINT;
        v.extent[i] = b->dim[i].extent;
        v.stride[i] = b->dim[i].stride;
    }
    if (v.stride[0] != 1) return VS_HALIDE_ERR_CONSTRAINT;
    return VS_HALIDE_OK;
}
// planes the kernels address densely: stride of dim k = product of the extents below it
int dense(const View& v, int dims) {
    long long s = 1;
    for (int i = 0; i < dims; i++) {
        if (v.stride[i] != s) return VS_HALIDE_ERR_CONSTRAINT;
        s *= v.extent[i];
    }
    return VS_HALIDE_OK;
}
int rows_ok(const View& v) { return v.stride[1] >= v.extent[0] ? VS_HALIDE_OK : VS_HALIDE_ERR_CONSTRAINT; }
int status(int r) { return r < 0 ? VS_HALIDE_ERR_GENERIC : VS_HALIDE_OK; }

#define VS_H(expr) do { int _e = (expr); if (_e) return _e; } while (0)

int grad_argmax_ts(int ts, vs_halide_buffer_t* gx, vs_halide_buffer_t* gy, vs_halide_buffer_t* lmx, vs_halide_buffer_t* lmy) {
    View vgx, vgy, vx, vy;
    VS_H(check(gx, VS_HALIDE_TYPE_FLOAT, 32, 2, vgx)); VS_H(check(gy, VS_HALIDE_TYPE_FLOAT, 32, 2, vgy));
    VS_H(check(lmx, VS_HALIDE_TYPE_UINT, 16, 3, vx)); VS_H(check(lmy, VS_HALIDE_TYPE_UINT, 16, 3, vy));
    VS_H(dense(vgx, 2)); VS_H(dense(vgy, 2)); VS_H(dense(vx, 3)); VS_H(dense(vy, 3));
    const int w = vgx.extent[0], h = vgx.extent[1];
    if (vgy.extent[0] != w || vgy.extent[1] != h) return VS_HALIDE_ERR_OUT_OF_BOUNDS;
    // generators.cpp:275-293: tile (x, y) reads gradients at (x * ts + r.x, y * ts + r.y) -- the outputs' extents decide how
    // many tiles are computed, and they must lie inside the gradient planes (imgproc.cpp:163-172 allocates w / ts x h / ts x 2)
    for (const View* o : {&vx, &vy})
        if (o->extent[2] != 2 || o->extent[0] != w / ts || o->extent[1] != h / ts) return VS_HALIDE_ERR_OUT_OF_BOUNDS;
    return status(vs_grad_argmax((const float*)vgx.host, (const float*)vgy.host, w, h, ts, (uint16_t*)vx.host, (uint16_t*)vy.host,
                                 VS_MEM_HOST, nullptr));
}

}  // namespace

extern "C" {

int pyr_down(vs_halide_buffer_t* input, vs_halide_buffer_t* output) {
    View in, out;
    VS_H(check(input, VS_HALIDE_TYPE_UINT, 8, 2, in)); VS_H(check(output, VS_HALIDE_TYPE_UINT, 8, 2, out));
    VS_H(rows_ok(in)); VS_H(rows_ok(out));
    // generators.cpp:70 clamps reads to the input, so any output extent up to (w/2, h/2) -- what alignment.cpp:185-188 allocates -- is served
    if (out.extent[0] > (in.extent[0] + 1) / 2 || out.extent[1] > (in.extent[1] + 1) / 2) return VS_HALIDE_ERR_OUT_OF_BOUNDS;
    return status(vs_pyr_down(in.host, in.extent[0], in.extent[1], in.stride[1], out.host, out.extent[0], out.extent[1], out.stride[1],
                              VS_MEM_HOST, nullptr));
}

int grad_xy(vs_halide_buffer_t* input, vs_halide_buffer_t* output_x, vs_halide_buffer_t* output_y) {
    View in, gx, gy;
    VS_H(check(input, VS_HALIDE_TYPE_UINT, 8, 2, in));
    VS_H(check(output_x, VS_HALIDE_TYPE_FLOAT, 32, 2, gx)); VS_H(check(output_y, VS_HALIDE_TYPE_FLOAT, 32, 2, gy));
    VS_H(rows_ok(in)); VS_H(dense(gx, 2)); VS_H(dense(gy, 2));
    for (const View* o : {&gx, &gy})
        if (o->extent[0] != in.extent[0] || o->extent[1] != in.extent[1]) return VS_HALIDE_ERR_OUT_OF_BOUNDS;
    return status(vs_grad_xy(in.host, in.extent[0], in.extent[1], in.stride[1], (float*)gx.host, (float*)gy.host, VS_MEM_HOST, nullptr));
}

#define VS_ARGMAX(TS) \
    int grad_argmax_##TS(vs_halide_buffer_t* gx, vs_halide_buffer_t* gy, vs_halide_buffer_t* lmx, vs_halide_buffer_t* lmy) { \
        return grad_argmax_ts(TS, gx, gy, lmx, lmy); \
    }
VS_ARGMAX(2) VS_ARGMAX(4) VS_ARGMAX(6) VS_ARGMAX(8) VS_ARGMAX(10) VS_ARGMAX(12) VS_ARGMAX(14) VS_ARGMAX(16) VS_ARGMAX(18) VS_ARGMAX(20)
#undef VS_ARGMAX

int sparse_jac(vs_halide_buffer_t* grad_x, vs_halide_buffer_t* grad_y, vs_halide_buffer_t* local_max_x, vs_halide_buffer_t* local_max_y,
               vs_halide_buffer_t* output_x, vs_halide_buffer_t* output_y) {
    View gx, gy, lx, ly, ox, oy;
    VS_H(check(grad_x, VS_HALIDE_TYPE_FLOAT, 32, 2, gx)); VS_H(check(grad_y, VS_HALIDE_TYPE_FLOAT, 32, 2, gy));
    VS_H(check(local_max_x, VS_HALIDE_TYPE_UINT, 16, 3, lx)); VS_H(check(local_max_y, VS_HALIDE_TYPE_UINT, 16, 3, ly));
    VS_H(check(output_x, VS_HALIDE_TYPE_FLOAT, 32, 3, ox)); VS_H(check(output_y, VS_HALIDE_TYPE_FLOAT, 32, 3, oy));
    VS_H(dense(gx, 2)); VS_H(dense(gy, 2)); VS_H(dense(lx, 3)); VS_H(dense(ly, 3)); VS_H(dense(ox, 3)); VS_H(dense(oy, 3));
    const int w = gx.extent[0], h = gx.extent[1], tx = lx.extent[0], ty = lx.extent[1];
    if (gy.extent[0] != w || gy.extent[1] != h || lx.extent[2] != 2 || ly.extent[0] != tx || ly.extent[1] != ty || ly.extent[2] != 2)
        return VS_HALIDE_ERR_OUT_OF_BOUNDS;
    for (const View* o : {&ox, &oy})
        if (o->extent[0] != tx || o->extent[1] != ty || o->extent[2] != 4) return VS_HALIDE_ERR_OUT_OF_BOUNDS;
    return status(vs_sparse_jac((const float*)gx.host, (const float*)gy.host, w, h, (const uint16_t*)lx.host, (const uint16_t*)ly.host, tx, ty,
                                (float*)ox.host, (float*)oy.host, VS_MEM_HOST, nullptr));
}

int sparse_warpdiff(vs_halide_buffer_t* input_template, vs_halide_buffer_t* input_keyframe, vs_halide_buffer_t* local_max,
                    float A, float B, float TX, float TY, vs_halide_buffer_t* output) {
    View tm, ky, lm, out;
    VS_H(check(input_template, VS_HALIDE_TYPE_UINT, 8, 2, tm)); VS_H(check(input_keyframe, VS_HALIDE_TYPE_UINT, 8, 2, ky));
    VS_H(check(local_max, VS_HALIDE_TYPE_UINT, 16, 3, lm)); VS_H(check(output, VS_HALIDE_TYPE_UINT, 16, 2, out));
    VS_H(rows_ok(tm)); VS_H(rows_ok(ky)); VS_H(dense(lm, 3)); VS_H(dense(out, 2));
    // the kernel takes one row stride for both images (the reference's two are pyramid levels of equal geometry)
    if (tm.extent[0] != ky.extent[0] || tm.extent[1] != ky.extent[1] || tm.stride[1] != ky.stride[1]) return VS_HALIDE_ERR_OUT_OF_BOUNDS;
    if (lm.extent[2] != 2 || out.extent[0] != lm.extent[0] || out.extent[1] != lm.extent[1]) return VS_HALIDE_ERR_OUT_OF_BOUNDS;
    return status(vs_sparse_warpdiff(tm.host, ky.host, ky.extent[0], ky.extent[1], ky.stride[1], (const uint16_t*)lm.host, lm.extent[0],
                                     lm.extent[1], A, B, TX, TY, (uint16_t*)out.host, VS_MEM_HOST, nullptr));
}

int sparse_ica(vs_halide_buffer_t* input_template, vs_halide_buffer_t* input_keyframe, vs_halide_buffer_t* selected_pixels_x,
               vs_halide_buffer_t* selected_pixels_y, vs_halide_buffer_t* selected_jacobians_x, vs_halide_buffer_t* selected_jacobians_y,
               float A, float B, float TX, float TY, vs_halide_buffer_t* output) {
    View tm, ky, sx, sy, jx, jy, out;
    VS_H(check(input_template, VS_HALIDE_TYPE_UINT, 8, 2, tm)); VS_H(check(input_keyframe, VS_HALIDE_TYPE_UINT, 8, 2, ky));
    VS_H(check(selected_pixels_x, VS_HALIDE_TYPE_UINT, 16, 2, sx)); VS_H(check(selected_pixels_y, VS_HALIDE_TYPE_UINT, 16, 2, sy));
    VS_H(check(selected_jacobians_x, VS_HALIDE_TYPE_FLOAT, 32, 2, jx)); VS_H(check(selected_jacobians_y, VS_HALIDE_TYPE_FLOAT, 32, 2, jy));
    VS_H(check(output, VS_HALIDE_TYPE_FLOAT, 64, 1, out));
    VS_H(rows_ok(tm)); VS_H(rows_ok(ky)); VS_H(dense(sx, 2)); VS_H(dense(sy, 2)); VS_H(dense(jx, 2)); VS_H(dense(jy, 2));
    if (tm.extent[0] != ky.extent[0] || tm.extent[1] != ky.extent[1] || tm.stride[1] != ky.stride[1]) return VS_HALIDE_ERR_OUT_OF_BOUNDS;
    const int nx = sx.extent[0], ny = sy.extent[0];
    if (sx.extent[1] != 2 || sy.extent[1] != 2 || jx.extent[0] != nx || jx.extent[1] != 4 || jy.extent[0] != ny || jy.extent[1] != 4 ||
        out.extent[0] != 4)
        return VS_HALIDE_ERR_OUT_OF_BOUNDS;
    return status(vs_sparse_ica(tm.host, ky.host, ky.extent[0], ky.extent[1], ky.stride[1], (const uint16_t*)sx.host, nx,
                                (const uint16_t*)sy.host, ny, (const float*)jx.host, (const float*)jy.host, A, B, TX, TY, (double*)out.host,
                                VS_MEM_HOST, nullptr));
}

int image_warp(vs_halide_buffer_t* input, float A, float B, float TX, float TY, vs_halide_buffer_t* output) {
    View in, out;
    VS_H(check(input, VS_HALIDE_TYPE_UINT, 8, 2, in)); VS_H(check(output, VS_HALIDE_TYPE_FLOAT, 32, 2, out));
    VS_H(rows_ok(in)); VS_H(dense(out, 2));
    return status(vs_image_warp(in.host, in.extent[0], in.extent[1], in.stride[1], A, B, TX, TY, (float*)out.host, out.extent[0],
                                out.extent[1], VS_MEM_HOST, nullptr));
}

}  // extern "C"
