// facade_cvmat_test.cpp -- the cv::Mat overloads of the drop-in facade, compiled and run against a cv::Mat TEST DOUBLE
// (tests/cpp/stubs/opencv2/core.hpp: this image has no OpenCV).  These are the signatures the reference's callers use:
//   VideoAligner::AlignNextFrame(const cv::Mat&, SimilarityTransform&, const VideoAlignerParams&)   alignment.hpp:55-58
//   cv::Mat VideoStabilizer::processFrame(const cv::Mat&)                                            stabilizer.hpp:39
//   cv::Mat warpBySimilarityTransform(const cv::Mat&, const SimilarityTransform&)                    imgproc.hpp:97
// with the argument shapes of their call sites (stabilizer.cpp:19, :97-99, video_test.cpp:106).  Every result must equal the
// pointer overload's bit for bit; a wrong Mat type throws std::runtime_error like the reference's adapters
// (imgproc.cpp:207-209, 239-241); a Mat with padded rows (step > cols * 3) is read through its step.
//   usage: facade_cvmat_test cpu   -> the type checks only (they come before any device call)
//          facade_cvmat_test gpu   -> + the three overloads against the pointer forms
#include <cmath>
#include <cstdio>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../video_stabilizer_amd/facade/stabilizer.hpp"

#ifndef VS_FACADE_HAVE_OPENCV
#error "build with -I tests/cpp/stubs: the cv::Mat overloads must be compiled"
#endif

static int fails = 0;
#define CHECK(cond) do { if (!(cond)) { std::printf("[FAIL] %s:%d %s\n", __FILE__, __LINE__, #cond); fails++; } } while (0)

template <class F>
static bool throws_runtime_error(F&& f) {
    try { f(); } catch (const std::runtime_error&) { return true; } catch (...) { return false; }
    return false;
}

static cv::Mat texture(int w, int h, double dx, double dy, size_t pad_bytes = 0, std::vector<unsigned char>* backing = nullptr) {
    // smooth synthetic BGR frame with three different channels, shifted by (dx, dy); pad_bytes > 0: rows padded (step > cols*3)
    cv::Mat m;
    if (pad_bytes == 0) m = cv::Mat(h, w, CV_8UC3);
    else {
        backing->assign(((size_t)w * 3 + pad_bytes) * h, 0xAB);         // the padding carries a pattern nothing may read as pixels
        m = cv::Mat(h, w, CV_8UC3, backing->data(), (size_t)w * 3 + pad_bytes);
    }
    for (int y = 0; y < h; y++) {
        unsigned char* row = m.ptr(y);
        for (int x = 0; x < w; x++) {
            const double u = x + dx, v = y + dy;
            const double g = 128 + 50 * std::sin(u * 0.11) * std::cos(v * 0.07) + 40 * std::sin((u + v) * 0.045) + 30 * std::cos(u * 0.031 - v * 0.052);
            for (int c = 0; c < 3; c++) {
                int q = (int)std::floor(g + 9.0 * c * std::sin(u * 0.02 * (c + 1)) + 0.5);
                row[3 * x + c] = (unsigned char)(q < 0 ? 0 : (q > 255 ? 255 : q));
            }
        }
    }
    return m;
}
static std::vector<uint8_t> dense_copy(const cv::Mat& m) {
    std::vector<uint8_t> v((size_t)m.rows * m.cols * 3);
    for (int y = 0; y < m.rows; y++) std::memcpy(v.data() + (size_t)y * m.cols * 3, m.ptr(y), (size_t)m.cols * 3);
    return v;
}
static bool same(const cv::Mat& m, const std::vector<uint8_t>& v) {
    if ((size_t)m.rows * m.cols * 3 != v.size()) return false;
    for (int y = 0; y < m.rows; y++) if (std::memcmp(m.ptr(y), v.data() + (size_t)y * m.cols * 3, (size_t)m.cols * 3)) return false;
    return true;
}

static void TestWrongMatTypesThrow() {
    cv::Mat gray(48, 64, CV_8UC1), bgra(48, 64, CV_8UC4), wide(48, 64, CV_16UC3), empty;
    SimilarityTransform t;
    VideoAligner aligner;
    for (const cv::Mat* m : {&gray, &bgra, &wide, &empty}) {
        CHECK(throws_runtime_error([&] { aligner.AlignNextFrame(*m, t); }));
        CHECK(throws_runtime_error([&] { (void)warpBySimilarityTransform(*m, t); }));
    }
    CHECK(aligner.handle() == nullptr);                                // the check comes first: no handle, no device touched
}

static void TestOverloadsEqualThePointerForms() {
    const int w = 640, h = 480;
    // ---- AlignNextFrame(const cv::Mat&, ...) as stabilizer.cpp:19 calls it, dense and padded-step Mats ----
    std::vector<unsigned char> back0, back1;
    cv::Mat a = texture(w, h, 0, 0), b = texture(w, h, 2.5, -1.75);
    cv::Mat ap = texture(w, h, 0, 0, 20, &back0), bp = texture(w, h, 2.5, -1.75, 20, &back1);
    CHECK(!ap.isContinuous() && ap.step == (size_t)w * 3 + 20);
    const auto av = dense_copy(a), bv = dense_copy(b);
    VideoAlignerParams params;
    SimilarityTransform tp1, tp2, tm1, tm2, ts1, ts2;
    { VideoAligner al; CHECK(!al.AlignNextFrame(av.data(), w, h, tp1, params)); CHECK(al.AlignNextFrame(bv.data(), w, h, tp2, params)); }
    { VideoAligner al; CHECK(!al.AlignNextFrame(a, tm1, params)); CHECK(al.AlignNextFrame(b, tm2, params)); }
    { VideoAligner al; CHECK(!al.AlignNextFrame(ap, ts1, params)); CHECK(al.AlignNextFrame(bp, ts2, params)); }
    CHECK(tm2.A == tp2.A && tm2.B == tp2.B && tm2.TX == tp2.TX && tm2.TY == tp2.TY);
    CHECK(ts2.A == tp2.A && ts2.B == tp2.B && ts2.TX == tp2.TX && ts2.TY == tp2.TY);
    CHECK(tm1.A == 0 && tm1.TX == 0 && std::fabs(std::fabs(tm2.TX) - 2.5) < 0.3);
    std::printf("AlignNextFrame(cv::Mat): %s\n", tm2.toString().c_str());

    // ---- warpBySimilarityTransform(const cv::Mat&, correction) as stabilizer.cpp:97-99 calls it ----
    SimilarityTransform corr; corr.A = 0.004; corr.B = -0.003; corr.TX = 3.25; corr.TY = -1.5;
    std::vector<uint8_t> want((size_t)w * h * 3);
    CHECK(warpBySimilarityTransform(av.data(), w, h, corr, want.data()));
    cv::Mat wm = warpBySimilarityTransform(a, corr), wp = warpBySimilarityTransform(ap, corr);
    CHECK(wm.rows == h && wm.cols == w && wm.type() == CV_8UC3 && same(wm, want));
    CHECK(same(wp, want));                                             // the padding bytes (0xAB) never entered a sample
    std::vector<uint8_t> want_l((size_t)w * h * 3);
    CHECK(warpBySimilarityTransform(av.data(), w, h, corr, want_l.data(), VS_WARP_LANCZOS2, VS_BORDER_CLAMP));
    CHECK(same(warpBySimilarityTransform(ap, corr, VS_WARP_LANCZOS2, VS_BORDER_CLAMP), want_l) && want_l != want);

    // ---- processFrame(const cv::Mat&) as video_test.cpp:106 calls it: empty Mat for the first `lag` frames ----
    VideoStabilizerParams sp; sp.lag = 3; sp.smoother_memory = 1; sp.crop_pixels = 8;
    VideoStabilizer st_ptr(sp), st_mat(sp), st_pad(sp);
    int produced = 0;
    for (int i = 0; i < 7; i++) {
        std::vector<unsigned char> back;
        cv::Mat f = texture(w, h, 0.4 * i, -0.3 * i), fp = texture(w, h, 0.4 * i, -0.3 * i, 12, &back);
        const auto fv = dense_copy(f);
        int ow = 0, oh = 0;
        const auto out_ptr = st_ptr.processFrame(fv.data(), w, h, ow, oh);
        cv::Mat out_mat = st_mat.processFrame(f), out_pad = st_pad.processFrame(fp);
        if (i < 3) CHECK(out_ptr.empty() && out_mat.empty() && out_pad.empty());
        else {
            CHECK(!out_mat.empty() && out_mat.cols == w - 16 && out_mat.rows == h - 16 && out_mat.type() == CV_8UC3);
            CHECK(same(out_mat, out_ptr) && same(out_pad, out_ptr));
            produced++;
        }
    }
    CHECK(produced == 4);
    // a wrong Mat type in the middle of a clip throws and leaves the stabilizer usable
    cv::Mat gray(h, w, CV_8UC1);
    CHECK(throws_runtime_error([&] { (void)st_mat.processFrame(gray); }));
    CHECK(!st_mat.processFrame(texture(w, h, 2.8, -2.1)).empty());
}

int main(int argc, char** argv) {
    const std::string mode = argc > 1 ? argv[1] : "cpu";
    TestWrongMatTypesThrow();
    if (mode == "gpu") TestOverloadsEqualThePointerForms();
    std::printf(fails ? "FAILED (%d)\n" : "ALL PASS\n", fails);
    return fails ? 1 : 0;
}
