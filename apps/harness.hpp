// harness.hpp -- what the four harness programs share: a clip resident in HBM and the jitter score of a device clip.
// The clip is uploaded once; every parameter combination of a grid search then runs stabilizer + scoring without a
// frame crossing PCIe again (the reference keeps the decoded clip in host memory for the same reason,
// grid_search_align.cpp:121-124).
#pragma once
#include <hip/hip_runtime_api.h>
#include <algorithm>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>
#include "jitter.hpp"
#include "video_io.hpp"
#include "vs_amd.h"

namespace vsh {

inline void hip_check(hipError_t e, const char* what) {
    if (e != hipSuccess) throw std::runtime_error(std::string(what) + ": " + hipGetErrorString(e));
}
inline void vs_check(int r, const char* what) {
    if (r < 0) throw std::runtime_error(std::string(what) + ": " + vs_last_error());
}
// sample depth -> frame format; depths the library has no format for are refused (the aligner's luma shift and the warp's
// saturation value follow the format)
inline int vs_format_of(const vsio::Format& f) {
    switch (f.bits) {
        case 8: return VS_FMT_BGR8;
        case 10: return VS_FMT_BGR10;
        case 12: return VS_FMT_BGR12;
        case 16: return VS_FMT_BGR16_FULL;
        default: throw std::runtime_error("unsupported sample depth " + std::to_string(f.bits) + " (8, 10, 12 or 16 bits)");
    }
}

struct DeviceBuffer {
    void* ptr = nullptr;
    size_t bytes = 0;
    DeviceBuffer() = default;
    explicit DeviceBuffer(size_t n) { reset(n); }
    DeviceBuffer(const DeviceBuffer&) = delete;
    DeviceBuffer& operator=(const DeviceBuffer&) = delete;
    ~DeviceBuffer() { if (ptr) (void)hipFree(ptr); }
    void reset(size_t n) {
        if (ptr) { (void)hipFree(ptr); ptr = nullptr; }
        bytes = n;
        if (n) hip_check(hipMalloc(&ptr, n), "hipMalloc");
    }
};

struct DeviceClip {
    vsio::Format fmt;
    int frames = 0;
    DeviceBuffer buf;
    size_t frame_elems() const { return fmt.bgr_elems(); }
    int device = 0;               // the device the buffer lives on
    void upload(const vsio::Clip& c, int dev = 0) {
        hip_check(hipSetDevice(dev), "hipSetDevice");
        device = dev;
        fmt = c.fmt;
        frames = (int)c.frames;
        buf.reset(c.data.size());
        hip_check(hipMemcpy(buf.ptr, c.data.data(), c.data.size(), hipMemcpyHostToDevice), "hipMemcpy H2D");
    }
};

// Jitter score of n device-resident frames (w x h interleaved BGR, frame i at d_frames + i*frame_stride elements),
// measured with the reference's default aligner parameters so the instrument does not depend on the parameters under
// test.  Pairs the aligner gives up on contribute the estimate it had reached.
inline double measure_jitter(vs_aligner* a, const void* d_frames, size_t frame_stride, int n, int w, int h, int format) {
    if (n < 2) return 0.0;                                   // grid_search_align.cpp:29
    std::vector<vs_transform> t((size_t)n);
    std::vector<int32_t> status((size_t)n);
    vs_check(vs_aligner_reset(a), "vs_aligner_reset");
    vs_check(vs_aligner_align_batch(a, d_frames, frame_stride, n, w, h, w * 3, format, VS_MEM_DEVICE, nullptr, t.data(), status.data()),
             "vs_aligner_align_batch");
    return vsjit::jitter(t.data(), n, w, h);
}

// `video [-j N] [--device D | --devices a,b,...|all] [--frames M]` as the grid searches take it (grid_search_align.cpp:62-90)
struct GridArgs {
    std::string video;
    int jobs = 4;        // worker threads, each with its own stabilizer + scoring handles; worker t works on device slot t mod G
    int device = 0;
    std::vector<int> devices;   // the device slots (--devices); empty = {device}.  A device may be listed more than once: every
                                // slot has its own copy of the clip, like a GPU of its own (how the one-GPU box rehearses the split)
    size_t max_frames = 0;
    bool parse(int argc, char** argv) {
        for (int i = 1; i < argc; i++) {
            const std::string a = argv[i];
            auto value = [&](int& dst) { if (i + 1 >= argc) return false; dst = std::atoi(argv[++i]); return true; };
            int v = 0;
            if (a == "-j" || a == "--jobs") { if (!value(v)) return false; jobs = std::max(1, v); }
            else if (a == "--device") { if (!value(v)) return false; device = v; }
            else if (a == "--devices") {
                if (i + 1 >= argc) return false;
                const std::string list = argv[++i];
                devices.clear();
                if (list == "all") {
                    for (int d = 0; d < vs_device_count(); d++) devices.push_back(d);
                } else {
                    size_t pos = 0;
                    while (pos <= list.size()) {
                        const size_t comma = std::min(list.find(',', pos), list.size());
                        const std::string tok = list.substr(pos, comma - pos);
                        if (tok.empty() || tok.find_first_not_of("0123456789") != std::string::npos) return false;
                        devices.push_back(std::atoi(tok.c_str()));
                        pos = comma + 1;
                    }
                }
                if (devices.empty()) return false;
            }
            else if (a == "--frames") { if (!value(v)) return false; max_frames = (size_t)std::max(0, v); }
            else if (a == "--dump-ratios") dump = true;
            else video = a;
        }
        return !video.empty();
    }
    bool dump = false;          // print every combination's ratio in index order after the search (tests compare device splits)
    std::vector<int> slots() const { return devices.empty() ? std::vector<int>{device} : devices; }
};

}  // namespace vsh
