// vs_latency: what one AlignNextFrame costs from C++ -- the reference's own call pattern (alignment.cpp:206: one frame per
// call), frames already in device memory.  No Python, no bindings: the C ABI of include/vs_amd.h and the HIP runtime only.
//   usage: vs_latency [width height frames levels_min_width passes select_mode]      (default 1920 1080 48 256 24 1)
//   select_mode: 1 = VS_SELECT_DEVICE (libstdc++'s order), 2 = VS_SELECT_STABLE (the documented STL-independent rule), 0 = host
// The clip is walked `passes` times and the best pass is reported: the card's shader clock needs tens of milliseconds of work to
// settle (with VS_LATENCY_VERBOSE=1 every pass is printed: ~0.24 ms per call in the first passes, flat after ~100 ms).
#include <hip/hip_runtime.h>

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../include/vs_amd.h"
#include "synth_clip.hpp"

using namespace vssynth;

int main(int argc, char** argv) {
    const int w = argc > 1 ? atoi(argv[1]) : 1920, h = argc > 2 ? atoi(argv[2]) : 1080, n = argc > 3 ? atoi(argv[3]) : 48;
    const int min_w = argc > 4 ? atoi(argv[4]) : 256;
    const int passes = argc > 5 ? atoi(argv[5]) : 24;
    const int select_mode = argc > 6 ? atoi(argv[6]) : VS_SELECT_DEVICE;
    const bool verbose = getenv("VS_LATENCY_VERBOSE") != nullptr;
    if (vs_device_count() < 1) { std::fprintf(stderr, "Error: no HIP device\n"); return 1; }
    vs_aligner_params p;
    vs_aligner_params_default(&p);
    p.pyramid_min_width = min_w;
    vs_aligner* a = vs_aligner_create(&p, 0);
    if (!a) { std::fprintf(stderr, "Error: %s\n", vs_last_error()); return 1; }
    if (vs_aligner_set_select_mode(a, select_mode) != VS_OK) { std::fprintf(stderr, "Error: %s\n", vs_last_error()); return 1; }
    const size_t fs = (size_t)w * h * 3;
    uint8_t* dev = nullptr;
    if (hipMalloc((void**)&dev, fs * n) != hipSuccess) { std::fprintf(stderr, "Error: hipMalloc\n"); return 1; }
    std::vector<uint8_t> host(fs);
    const std::vector<float> tex = base_texture(w + 2 * kMargin, h + 2 * kMargin);
    for (int i = 0; i < n; i++) {
        make_clip_frame(host, tex, w, h, i);          // slow pan + independent jitter per frame (the synthetic clips of bench.py)
        if (hipMemcpy(dev + fs * i, host.data(), fs, hipMemcpyHostToDevice) != hipSuccess) return 1;
    }
    double best = 1e30;
    int good = 0;
    long iters = 0;
    for (int rep = 0; rep < passes; rep++) {
        vs_aligner_reset(a);
        good = 0;
        const auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < n; i++) {
            vs_transform t;
            int st = 0;
            if (vs_aligner_align_batch(a, dev + fs * i, fs, 1, w, h, 3 * w, VS_FMT_BGR8, VS_MEM_DEVICE, nullptr, &t, &st) < 0) {
                std::fprintf(stderr, "Error: %s\n", vs_last_error());
                return 1;
            }
            good += st;
            vs_align_info inf;
            if (rep == 0 && vs_aligner_get_info(a, 0, &inf) == VS_OK) for (int l = 0; l < inf.levels; l++) iters += inf.iterations[l];
        }
        const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() / n;
        if (ms < best) best = ms;
        if (verbose) std::fprintf(stderr, "pass %d: %.4f ms per call\n", rep, ms);
    }
    std::printf("{\"w\": %d, \"h\": %d, \"frames\": %d, \"aligned\": %d, \"select_mode\": %d, \"gn_iterations_per_frame\": %.2f, \"ms_per_call\": %.4f}\n", w, h, n, good,
                select_mode, (double)iters / (n > 1 ? n - 1 : 1), best);
    vs_aligner_destroy(a);
    (void)hipFree(dev);
    return 0;
}
