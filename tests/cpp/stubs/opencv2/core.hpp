// TEST DOUBLE, not OpenCV: the handful of cv::Mat members the facade's cv::Mat overloads touch (video_stabilizer_amd/facade/
// imgproc.hpp, alignment.hpp, stabilizer.hpp), so that those overloads -- the signatures the reference's callers use
// (alignment.hpp:55-58, stabilizer.hpp:34-39, imgproc.hpp:97) -- are compiled and run in an image without OpenCV.
// Semantics follow OpenCV's documented cv::Mat: row-major, `step` bytes from row to row (>= cols * elemSize()), `type()` packs
// depth and channel count as CV_MAKETYPE does, a default-constructed Mat is empty(), copies share the pixel buffer.
// Only tests/cpp/facade_cvmat_test.cpp includes this (through -I tests/cpp/stubs); nothing in the product does.
#pragma once
#include <cstddef>
#include <cstdint>
#include <memory>
#include <vector>

#define CV_8U 0
#define CV_16U 2
#define CV_32F 5
#define CV_CN_SHIFT 3
#define CV_MAKETYPE(depth, cn) (((depth) & 7) + (((cn) - 1) << CV_CN_SHIFT))
#define CV_8UC1 CV_MAKETYPE(CV_8U, 1)
#define CV_8UC3 CV_MAKETYPE(CV_8U, 3)
#define CV_8UC4 CV_MAKETYPE(CV_8U, 4)
#define CV_16UC3 CV_MAKETYPE(CV_16U, 3)
#define CV_32FC1 CV_MAKETYPE(CV_32F, 1)

namespace cv {
class Mat {
public:
    int rows = 0, cols = 0;
    unsigned char* data = nullptr;
    size_t step = 0;                       // bytes per row (cv::Mat::step converts to size_t the same way)

    Mat() = default;
    Mat(int r, int c, int type) : rows(r), cols(c), type_(type) {
        step = (size_t)c * elemSize();
        own_ = std::make_shared<std::vector<unsigned char>>(step * (size_t)r);
        data = own_->data();
    }
    // user-allocated data with an explicit row step (cv::Mat(rows, cols, type, data, step)): not owned
    Mat(int r, int c, int type, void* d, size_t st) : rows(r), cols(c), data((unsigned char*)d), step(st), type_(type) {}
    int type() const { return type_; }
    int channels() const { return (type_ >> CV_CN_SHIFT) + 1; }
    size_t elemSize1() const { const int d = type_ & 7; return d == CV_8U ? 1 : (d == CV_16U ? 2 : 4); }
    size_t elemSize() const { return elemSize1() * (size_t)channels(); }
    bool empty() const { return data == nullptr || rows == 0 || cols == 0; }
    bool isContinuous() const { return step == (size_t)cols * elemSize(); }
    unsigned char* ptr(int y) { return data + (size_t)y * step; }
    const unsigned char* ptr(int y) const { return data + (size_t)y * step; }
private:
    int type_ = 0;
    std::shared_ptr<std::vector<unsigned char>> own_;
};
}  // namespace cv
