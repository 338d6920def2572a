// stabilizer.hpp -- drop-in for the reference's stabilizer.hpp:13-56 on the C ABI.
#pragma once
#include <vector>
#include "alignment.hpp"
#include "smoother.hpp"

// stabilizer.hpp:13-30 -- same fields, same defaults (+ the two sampler knobs of this build)
struct VideoStabilizerParams {
    VideoAlignerParams aligner;
    int lag = 10;
    int smoother_memory = 5;
    double lambda = 4.0;
    bool enable_smoother = true;
    int crop_pixels = 32;
    double min_disp = 48.0, max_disp = 64.0;
    double min_decay = 0.9, max_decay = 0.7;
    int warp_mode = VS_WARP_BILINEAR_CV;  // cv::warpAffine(INTER_LINEAR) as the reference calls it (imgproc.cpp:472), fixed point; VS_WARP_LANCZOS2* = bgr_image_warp
    int warp_border = VS_BORDER_CONSTANT;
};

// stabilizer.hpp:32-56.  processFrame returns an empty vector until `lag` frames have arrived
// (stabilizer.cpp:46-52), then the stabilized, cropped BGR frame (out_width x out_height x 3).
class VideoStabilizer {
public:
    explicit VideoStabilizer(const VideoStabilizerParams& params = VideoStabilizerParams(), int device = 0) {
        vs_stabilizer_params p;
        vs_stabilizer_params_default(&p);
        p.aligner = params.aligner.c();
        p.lag = params.lag; p.smoother_memory = params.smoother_memory; p.lambda = params.lambda;
        p.enable_smoother = params.enable_smoother ? 1 : 0; p.crop_pixels = params.crop_pixels;
        p.min_disp = params.min_disp; p.max_disp = params.max_disp; p.min_decay = params.min_decay; p.max_decay = params.max_decay;
        p.warp_mode = params.warp_mode; p.warp_border = params.warp_border;
        crop_ = params.crop_pixels > 0 ? params.crop_pixels : 0;
        vs::check_abi();
        h_ = vs_stabilizer_create(&p, device);
        if (!h_) throw std::runtime_error(std::string("vs_stabilizer_create: ") + vs_last_error());
    }
    ~VideoStabilizer() { vs_stabilizer_destroy(h_); }
    VideoStabilizer(const VideoStabilizer&) = delete;
    VideoStabilizer& operator=(const VideoStabilizer&) = delete;

    std::vector<uint8_t> processFrame(const uint8_t* bgr, int width, int height, int& out_width, int& out_height) {
        std::vector<uint8_t> out((size_t)(width - 2 * crop_) * (height - 2 * crop_) * 3);
        const int r = vs_stabilizer_process(h_, bgr, width, height, width * 3, VS_FMT_BGR8, VS_MEM_HOST, out.data(), &out_width, &out_height);
        if (r < 0) throw std::runtime_error(std::string("vs_stabilizer_process: ") + vs_last_error());
        if (r == 0) { out.clear(); out_width = out_height = 0; }
        return out;
    }
#ifdef VS_FACADE_HAVE_OPENCV
    // stabilizer.hpp:39: an empty cv::Mat until `lag` frames have arrived, then the stabilized, cropped frame
    cv::Mat processFrame(const cv::Mat& inputFrame) {
        vs::require_bgr8(inputFrame, "VideoStabilizer::processFrame");
        cv::Mat out(inputFrame.rows - 2 * crop_, inputFrame.cols - 2 * crop_, CV_8UC3);
        int ow = 0, oh = 0;
        const int r = vs_stabilizer_process(h_, inputFrame.data, inputFrame.cols, inputFrame.rows, vs::mat_stride_elems(inputFrame), VS_FMT_BGR8,
                                            VS_MEM_HOST, out.data, &ow, &oh);
        if (r < 0) throw std::runtime_error(std::string("vs_stabilizer_process: ") + vs_last_error());
        return r == 1 ? out : cv::Mat();
    }
#endif
private:
    vs_stabilizer* h_ = nullptr;
    int crop_ = 0;
};
