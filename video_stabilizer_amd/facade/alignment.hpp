// alignment.hpp -- drop-in for the reference's alignment.hpp:5-99 on the C ABI.
#pragma once
#include "imgproc.hpp"

// alignment.hpp:5-41 -- same fields, same defaults
struct VideoAlignerParams {
    bool phase_correlate = false;
    double phase_correlate_threshold = 0.5;
    double threshold = 0.02;
    float smallest_fraction = 0.8f;
    int max_iters = 64;
    int pyramid_min_width = 20;
    int pyramid_min_height = 20;
    double max_displacement = 10.0;

    vs_aligner_params c() const {
        vs_aligner_params p;
        p.phase_correlate = phase_correlate ? 1 : 0;
        p.phase_correlate_threshold = phase_correlate_threshold;
        p.threshold = threshold;
        p.smallest_fraction = smallest_fraction;
        p.max_iters = max_iters;
        p.pyramid_min_width = pyramid_min_width;
        p.pyramid_min_height = pyramid_min_height;
        p.max_displacement = max_displacement;
        return p;
    }
};

// alignment.hpp:51-99.  The frame is an interleaved BGR u8 image (what the reference receives as a CV_8UC3
// cv::Mat and converts with cvtColor, alignment.cpp:212); the gray overload skips that conversion.
class VideoAligner {
public:
    explicit VideoAligner(int device = 0) : device_(device) {}
    ~VideoAligner() { if (h_) vs_aligner_destroy(h_); }
    VideoAligner(const VideoAligner&) = delete;
    VideoAligner& operator=(const VideoAligner&) = delete;

    // Returns false if track is lost or a kernel fails (alignment.hpp:54-58)
    bool AlignNextFrame(const uint8_t* bgr, int width, int height, SimilarityTransform& transform,
                        const VideoAlignerParams& params = VideoAlignerParams()) {
        return align(bgr, width, height, width * 3, VS_FMT_BGR8, transform, params);
    }
    bool AlignNextFrameGray(const uint8_t* gray, int width, int height, SimilarityTransform& transform,
                            const VideoAlignerParams& params = VideoAlignerParams()) {
        return align(gray, width, height, width, VS_FMT_GRAY8, transform, params);
    }
#ifdef VS_FACADE_HAVE_OPENCV
    // alignment.hpp:55-58: the reference's own signature
    bool AlignNextFrame(const cv::Mat& frame, SimilarityTransform& transform, const VideoAlignerParams& params = VideoAlignerParams()) {
        vs::require_bgr8(frame, "VideoAligner::AlignNextFrame");
        return align(frame.data, frame.cols, frame.rows, vs::mat_stride_elems(frame), VS_FMT_BGR8, transform, params);
    }
#endif
    vs_aligner* handle() { return h_; }

private:
    bool align(const uint8_t* data, int w, int h, int stride, int fmt, SimilarityTransform& transform, const VideoAlignerParams& params) {
        transform = SimilarityTransform();   // alignment.cpp:344
        const vs_aligner_params p = params.c();
        if (!h_) { vs::check_abi(); h_ = vs_aligner_create(&p, device_); }
        if (!h_) return false;
        vs_transform t{0, 0, 0, 0};
        const int r = vs_aligner_align_next(h_, data, w, h, stride, fmt, VS_MEM_HOST, &p, &t);
        // like the reference, a failed call leaves the estimate it had reached in `transform` (identity for the first frame)
        if (r >= 0) transform = SimilarityTransform::from(t);
        return r == 1;
    }
    vs_aligner* h_ = nullptr;
    int device_;
};
