// vs_eval_jitter -- median inter-frame motion of clips (the role of the reference's eval_jitter.cpp:21-75).
//   vs_eval_jitter [--device D] [--frames M] clip1 [clip2 ...]   ->   "<path>\tmedian_jitter_px=<value>" per clip
// The flow field is the similarity measured by this library instead of Farneback's (see jitter.hpp).
#include <iostream>
#include "harness.hpp"

int main(int argc, char** argv) {
    int device = 0;
    size_t max_frames = 0;
    std::vector<std::string> paths;
    for (int i = 1; i < argc; i++) {
        const std::string a = argv[i];
        if (a == "--device" && i + 1 < argc) device = std::atoi(argv[++i]);
        else if (a == "--frames" && i + 1 < argc) max_frames = (size_t)std::max(0, std::atoi(argv[++i]));
        else paths.push_back(a);
    }
    if (paths.empty()) { std::cerr << "Usage: " << argv[0] << " [--device D] [--frames M] video1 [video2 ...]\n"; return 1; }
    try {
        vs_aligner* a = vs_aligner_create(nullptr, device);
        if (!a) { std::cerr << "vs_aligner_create: " << vs_last_error() << "\n"; return 1; }
        std::cerr << vsjit::score_note() << std::endl;
        for (const auto& path : paths) {
            vsio::Clip clip;
            std::string err;
            if (!vsio::load_clip(path, clip, err, max_frames)) { std::cerr << "Cannot open video: " << path << " (" << err << ")\n"; continue; }
            if (clip.frames == 0) { std::cerr << "Empty video: " << path << "\n"; continue; }
            vsh::DeviceClip d;
            d.upload(clip);
            const double j = vsh::measure_jitter(a, d.buf.ptr, d.frame_elems(), d.frames, d.fmt.w, d.fmt.h, vsh::vs_format_of(d.fmt));
            std::cout << path << "\tmedian_jitter_px=" << j << "\n";
        }
        vs_aligner_destroy(a);
    } catch (const std::exception& e) {
        std::cerr << "Error: " << e.what() << "\n";
        return 1;
    }
    return 0;
}
