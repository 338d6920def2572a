// imgproc.hpp -- source-level drop-in for the reference's imgproc.hpp (imgproc.hpp:8-95), implemented purely
// on the C ABI of libvs_amd.so (include/vs_amd.h).  Same function names, argument order and `bool` protocol
// (true iff the underlying call succeeded; outputs passed by non-const reference and (re)allocated by the
// callee when their shape does not match -- imgproc.cpp:34-40,56-60,87-92,166-172).
//
// The operator functions are templates over the buffer class (see below): the small owning vs::Buffer<T> of this header
// (dense, dim0 stride 1, "planar" like Halide: element (x,y,c) at c*w*h + y*w + x) or Halide::Runtime::Buffer<T> itself.
// With OpenCV present a caller wraps cv::Mat data with vs::Buffer<T>::view().
#pragma once

#include <cmath>
#include <cstdint>
#include <cstring>
#include <sstream>
#include <stdexcept>
#include <string>
#include <type_traits>
#include <utility>
#include <vector>

#include "../../include/vs_amd.h"

namespace vs {

// The engine writes sizeof(vs_align_info) bytes per frame into caller memory and the enums have moved between releases: a
// translation unit built against another include/vs_amd.h than the library must stop before the first engine call.
inline void check_abi() {
    if (vs_abi_version() != VS_ABI_VERSION || vs_sizeof_align_info() != sizeof(vs_align_info))
        throw std::runtime_error("libvs_amd.so ABI " + std::to_string(vs_abi_version()) + " does not match the headers this program was built with (ABI " +
                                 std::to_string(VS_ABI_VERSION) + ")");
}

template <typename T>
class Buffer {
public:
    Buffer() = default;
    Buffer(int w, int h = 1, int c = 1) : w_(w), h_(h), c_(c), own_((size_t)w * h * c), p_(own_.data()), dims_(c > 1 ? 3 : (h > 1 ? 2 : 1)) {}
    static Buffer view(T* data, int w, int h, int c = 1) { Buffer b; b.w_ = w; b.h_ = h; b.c_ = c; b.p_ = data; b.dims_ = c > 1 ? 3 : 2; return b; }
    int dimensions() const { return dims_; }
    int width() const { return w_; }
    int height() const { return h_; }
    int channels() const { return c_; }
    T* data() { return p_; }
    const T* data() const { return p_; }
    T& operator()(int x, int y = 0, int c = 0) { return p_[((size_t)c * h_ + y) * w_ + x]; }
    const T& operator()(int x, int y = 0, int c = 0) const { return p_[((size_t)c * h_ + y) * w_ + x]; }
    size_t size() const { return (size_t)w_ * h_ * c_; }
private:
    int w_ = 0, h_ = 0, c_ = 1;
    std::vector<T> own_;
    T* p_ = nullptr;
    int dims_ = 0;
};

}  // namespace vs

// imgproc.hpp:34-38
struct Point {
    double x = 0.0, y = 0.0;
    double distance(const Point& p) const { double dx = x - p.x, dy = y - p.y; return std::sqrt(dx * dx + dy * dy); }
};

// imgproc.hpp:40-65
struct SimilarityTransform {
    double A = 0.0, B = 0.0, TX = 0.0, TY = 0.0;

    std::string toString() const {   // imgproc.cpp:327-331
        std::stringstream ss;
        ss << "A=" << A << ", B=" << B << ", TX=" << TX << ", TY=" << TY;
        return ss.str();
    }
    SimilarityTransform inverse() const { return from(vs_transform_inverse(&c())); }
    Point warp(Point p) const { vs_point q = vs_transform_warp(&c(), vs_point{p.x, p.y}); return Point{q.x, q.y}; }
    Point warp(Point p, double cx, double cy) const { vs_point q = vs_transform_warp_center(&c(), vs_point{p.x, p.y}, cx, cy); return Point{q.x, q.y}; }
    double maxCornerDisplacement(double width, double height) const { return vs_transform_max_corner_displacement(&c(), width, height); }
    // "this" = T1, param = T2: T3 = T2(T1(p))
    SimilarityTransform compose(const SimilarityTransform& w2) const { return from(vs_transform_compose(&c(), &w2.c())); }

    const vs_transform& c() const { return *reinterpret_cast<const vs_transform*>(this); }
    static SimilarityTransform from(const vs_transform& t) { SimilarityTransform s; s.A = t.A; s.B = t.B; s.TX = t.TX; s.TY = t.TY; return s; }
};
static_assert(sizeof(SimilarityTransform) == sizeof(vs_transform), "layout must match vs_transform");

// ---- the seven operator functions of imgproc.hpp:8-32 ---------------------------------------------------------------
// Written against the buffer INTERFACE, not a buffer class: anything with data() / width() / height() / dimensions() that is
// constructible from (w, h[, c]) and assignable -- vs::Buffer<T> above, and Halide::Runtime::Buffer<T> where a caller still has
// Halide's headers, so that the reference's own call sites (alignment.cpp:220-276, 435-546) compile unchanged.  The element
// types are the reference's (static_assert), the layout is dense planar (checked at run time where the class exposes
// dim(i).stride(), as Halide's does: freshly allocated Halide buffers are dense).
namespace vs {
template <class B> using elem_t = std::remove_cv_t<std::remove_pointer_t<decltype(std::declval<B&>().data())>>;
template <class B, class = void> struct has_dim : std::false_type {};
template <class B> struct has_dim<B, std::void_t<decltype(std::declval<const B&>().dim(0).stride())>> : std::true_type {};
template <class B, class = void> struct has_min : std::false_type {};
template <class B> struct has_min<B, std::void_t<decltype(std::declval<const B&>().dim(0).min())>> : std::true_type {};
template <class B, class = void> struct has_host_dirty : std::false_type {};
template <class B> struct has_host_dirty<B, std::void_t<decltype(std::declval<B&>().set_host_dirty())>> : std::true_type {};
// a buffer the kernels can address: host memory, dense planar strides, every dimension starting at 0 (a cropped or strided
// Halide buffer is refused, never written as if it were dense)
template <class B>
inline bool dense(const B& b) {
    if constexpr (has_dim<B>::value) {
        long want = 1;
        for (int d = 0; d < b.dimensions(); d++) {
            if (b.dim(d).stride() != want) return false;
            if constexpr (has_min<B>::value) { if (b.dim(d).min() != 0) return false; }
            want *= b.dim(d).extent();
        }
    }
    return b.data() != nullptr;
}
template <class... Bs> inline bool all_dense(const Bs&... bs) { return (dense(bs) && ...); }
// outputs were written through their host pointer: tell a buffer class that tracks it (Halide::Runtime::Buffer does)
template <class B> inline void wrote(B& b) { if constexpr (has_host_dirty<B>::value) b.set_host_dirty(); }
template <class... Bs> inline bool done(bool ok, Bs&... outs) { if (ok) (wrote(outs), ...); return ok; }
template <class B>
inline int channels_of(const B& b) {
    if constexpr (has_dim<B>::value) return b.dimensions() > 2 ? (int)b.dim(2).extent() : 1;
    else return b.channels();
}
}  // namespace vs

// imgproc.cpp:108-114
template <class BIn, class BOut>
inline bool PyrDown(BIn& input, BOut& output) {
    static_assert(std::is_same<vs::elem_t<BIn>, uint8_t>::value && std::is_same<vs::elem_t<BOut>, uint8_t>::value, "PyrDown: u8 -> u8");
    if (!vs::all_dense(input, output)) return false;
    return vs::done(vs_pyr_down(input.data(), input.width(), input.height(), input.width(), output.data(), output.width(), output.height(),
                                output.width(), VS_MEM_HOST, nullptr) == 0, output);
}
// imgproc.cpp:135-142
template <class BIn, class BOut>
inline bool GradXY(BIn& input, BOut& output_x, BOut& output_y) {
    static_assert(std::is_same<vs::elem_t<BIn>, uint8_t>::value && std::is_same<vs::elem_t<BOut>, float>::value, "GradXY: u8 -> f32, f32");
    if (!vs::all_dense(input, output_x, output_y)) return false;
    if (output_x.width() != input.width() || output_x.height() != input.height() || output_y.width() != input.width() ||
        output_y.height() != input.height()) return false;
    return vs::done(vs_grad_xy(input.data(), input.width(), input.height(), input.width(), output_x.data(), output_y.data(), VS_MEM_HOST,
                               nullptr) == 0, output_x, output_y);
}
// imgproc.cpp:144-202 (tile-size rule + (re)allocation + dispatch)
template <class BGrad, class BMax>
inline bool GradArgMax(BGrad& grad_x, BGrad& grad_y, int& tile_size, BMax& local_max_x, BMax& local_max_y) {
    static_assert(std::is_same<vs::elem_t<BGrad>, float>::value && std::is_same<vs::elem_t<BMax>, uint16_t>::value, "GradArgMax: f32, f32 -> u16, u16");
    tile_size = vs_tile_size(grad_x.width(), grad_y.height());
    const int wt = grad_x.width() / tile_size, ht = grad_y.height() / tile_size;
    if (local_max_x.dimensions() != 3 || local_max_x.width() != wt || local_max_x.height() != ht) {
        local_max_x = BMax(wt, ht, 2);
        local_max_y = BMax(wt, ht, 2);
    }
    // every buffer is checked, the outputs after their (re)allocation: an existing output of the right shape may still be a
    // cropped or strided view
    if (!vs::all_dense(grad_x, grad_y, local_max_x, local_max_y)) return false;
    if (local_max_y.dimensions() != 3 || local_max_y.width() != wt || local_max_y.height() != ht || vs::channels_of(local_max_x) != 2 ||
        vs::channels_of(local_max_y) != 2 || grad_y.width() != grad_x.width() || grad_y.height() != grad_x.height()) return false;
    return vs::done(vs_grad_argmax(grad_x.data(), grad_y.data(), grad_x.width(), grad_x.height(), tile_size, local_max_x.data(),
                                   local_max_y.data(), VS_MEM_HOST, nullptr) == 0, local_max_x, local_max_y);
}
// imgproc.cpp:26-44
template <class BGrad, class BMax, class BJac>
inline bool SparseJacobian(BGrad& grad_x, BGrad& grad_y, BMax& local_max_x, BMax& local_max_y, BJac& output_x, BJac& output_y) {
    static_assert(std::is_same<vs::elem_t<BGrad>, float>::value && std::is_same<vs::elem_t<BMax>, uint16_t>::value &&
                  std::is_same<vs::elem_t<BJac>, float>::value, "SparseJacobian: f32, f32, u16, u16 -> f32, f32");
    if (output_x.dimensions() != 3 || output_x.width() != local_max_x.width() || output_x.height() != local_max_x.height()) {
        output_x = BJac(local_max_x.width(), local_max_x.height(), 4);
        output_y = BJac(local_max_x.width(), local_max_x.height(), 4);
    }
    if (!vs::all_dense(grad_x, grad_y, local_max_x, local_max_y, output_x, output_y)) return false;
    if (output_y.dimensions() != 3 || output_y.width() != local_max_x.width() || output_y.height() != local_max_x.height() ||
        vs::channels_of(output_x) != 4 || vs::channels_of(output_y) != 4) return false;
    return vs::done(vs_sparse_jac(grad_x.data(), grad_y.data(), grad_x.width(), grad_x.height(), local_max_x.data(), local_max_y.data(),
                                  local_max_x.width(), local_max_x.height(), output_x.data(), output_y.data(), VS_MEM_HOST, nullptr) == 0,
                    output_x, output_y);
}
// imgproc.cpp:46-78: selected_pixels (n,2) u16, selected_jacobians (n,4) f32
template <class BImg, class BPix, class BJac, class BOut>
inline bool SparseICA(BImg& input_template, BImg& input_keyframe, BPix& selected_pixels_x, BPix& selected_pixels_y, BJac& selected_jacobians_x,
                      BJac& selected_jacobians_y, const SimilarityTransform& transform, BOut& output) {
    static_assert(std::is_same<vs::elem_t<BImg>, uint8_t>::value && std::is_same<vs::elem_t<BPix>, uint16_t>::value &&
                  std::is_same<vs::elem_t<BJac>, float>::value && std::is_same<vs::elem_t<BOut>, double>::value,
                  "SparseICA: u8, u8, u16, u16, f32, f32 -> f64");
    if (output.dimensions() != 1 || output.width() != 4) output = BOut(4);
    if (!vs::all_dense(input_template, input_keyframe, selected_pixels_x, selected_pixels_y, selected_jacobians_x, selected_jacobians_y, output))
        return false;
    float p[4];
    vs_ul_params_sparse(&transform.c(), input_template.width(), input_template.height(), p);
    return vs::done(vs_sparse_ica(input_template.data(), input_keyframe.data(), input_keyframe.width(), input_keyframe.height(), input_keyframe.width(),
                         selected_pixels_x.data(), selected_pixels_x.width(), selected_pixels_y.data(), selected_pixels_y.width(),
                         selected_jacobians_x.data(), selected_jacobians_y.data(), p[0], p[1], p[2], p[3], output.data(), VS_MEM_HOST,
                         nullptr) == 0, output);
}
// imgproc.cpp:80-106
template <class BImg, class BMax>
inline bool SparseWarpDiff(BImg& input_template, BImg& input_keyframe, BMax& local_max, const SimilarityTransform& transform, BMax& output) {
    static_assert(std::is_same<vs::elem_t<BImg>, uint8_t>::value && std::is_same<vs::elem_t<BMax>, uint16_t>::value, "SparseWarpDiff: u8, u8, u16 -> u16");
    if (output.dimensions() != 2 || output.width() != local_max.width() || output.height() != local_max.height())
        output = BMax(local_max.width(), local_max.height());
    if (!vs::all_dense(input_template, input_keyframe, local_max, output)) return false;
    float p[4];
    vs_ul_params_sparse(&transform.c(), input_template.width(), input_template.height(), p);
    return vs::done(vs_sparse_warpdiff(input_template.data(), input_keyframe.data(), input_keyframe.width(), input_keyframe.height(),
                              input_keyframe.width(), local_max.data(), local_max.width(), local_max.height(), p[0], p[1], p[2], p[3],
                              output.data(), VS_MEM_HOST, nullptr) == 0, output);
}
// imgproc.cpp:116-133
template <class BIn, class BOut>
inline bool ImageWarp(BIn& input, const SimilarityTransform& transform, BOut& output) {
    static_assert(std::is_same<vs::elem_t<BIn>, uint8_t>::value && std::is_same<vs::elem_t<BOut>, float>::value, "ImageWarp: u8 -> f32");
    if (!vs::all_dense(input, output)) return false;
    float p[4];
    vs_ul_params_warp(&transform.c(), input.width(), input.height(), p);
    return vs::done(vs_image_warp(input.data(), input.width(), input.height(), input.width(), p[0], p[1], p[2], p[3], output.data(),
                                  output.width(), output.height(), VS_MEM_HOST, nullptr) == 0, output);
}

// imgproc.cpp:446-484 warpBySimilarityTransform on an interleaved BGR image (h x w x 3, u8).  OpenCV's warpAffine
// without WARP_INVERSE_MAP inverts the matrix it is given (imgproc.cpp:472), i.e. it samples the source at
// transform^-1; bgr_image_warp takes the sampling map, so it receives transform.inverse().  The reference's
// interpolation is OpenCV's fixed-point bilinear with a black border: VS_WARP_BILINEAR_CV restates exactly that (cv::warpAffine's own
// matrix inversion included, so that mode receives `transform` itself) and is the default; VS_WARP_BILINEAR is the Halide sampler's float
// lerp (generators.cpp:148-163), the VS_WARP_LANCZOS2 family bgr_image_warp.
inline bool warpBySimilarityTransform(const uint8_t* src_bgr, int w, int h, const SimilarityTransform& transform, uint8_t* dst_bgr,
                                      int mode = VS_WARP_BILINEAR_CV, int border = VS_BORDER_CONSTANT) {
    const SimilarityTransform sampling = mode == VS_WARP_BILINEAR_CV ? transform : transform.inverse();
    return vs_bgr_image_warp(src_bgr, w, h, w * 3, 3, 8, &sampling.c(), mode, border, 255, dst_bgr, w * 3, VS_MEM_HOST, nullptr) == 0;
}

// ---- optional cv::Mat overloads: compiled only where the caller's own OpenCV headers exist -----------------------
// With them the reference's call sites compile unchanged: warpBySimilarityTransform(frame, correction)
// (stabilizer.cpp:98), aligner.AlignNextFrame(inputFrame, currentMeas, params) (stabilizer.cpp:19),
// stabilizer.processFrame(frame) (video_test.cpp:106).  Define VS_FACADE_NO_OPENCV to leave them out.
#if !defined(VS_FACADE_NO_OPENCV) && defined(__has_include)
#if __has_include(<opencv2/core.hpp>)
#include <opencv2/core.hpp>
#define VS_FACADE_HAVE_OPENCV 1
namespace vs {
// the reference's adapters throw std::runtime_error on a wrong Mat type (imgproc.cpp:207-209,239-241)
inline void require_bgr8(const cv::Mat& m, const char* who) {
    if (m.empty() || m.type() != CV_8UC3) throw std::runtime_error(std::string(who) + ": expected a non-empty CV_8UC3 (BGR) cv::Mat");
}
inline int mat_stride_elems(const cv::Mat& m) { return (int)(m.step / m.elemSize1()); }
}  // namespace vs

// imgproc.hpp:97 / imgproc.cpp:446-484.  Same defaults as the reference's call: cv::warpAffine's bilinear, black border.
inline cv::Mat warpBySimilarityTransform(const cv::Mat& src, const SimilarityTransform& transform, int mode = VS_WARP_BILINEAR_CV,
                                         int border = VS_BORDER_CONSTANT) {
    vs::require_bgr8(src, "warpBySimilarityTransform");
    cv::Mat dst(src.rows, src.cols, CV_8UC3);
    const SimilarityTransform sampling = mode == VS_WARP_BILINEAR_CV ? transform : transform.inverse();
    if (vs_bgr_image_warp(src.data, src.cols, src.rows, vs::mat_stride_elems(src), 3, 8, &sampling.c(), mode, border, 255, dst.data,
                          vs::mat_stride_elems(dst), VS_MEM_HOST, nullptr) != 0)
        throw std::runtime_error(std::string("vs_bgr_image_warp: ") + vs_last_error());
    return dst;
}
#endif
#endif

