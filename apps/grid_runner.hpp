// grid_runner.hpp -- the worker pool of the two grid searches (grid_search_align.cpp:150-205,
// grid_search_smoother.cpp:190-262): every parameter combination stabilizes the whole clip with a fresh stabilizer and
// is scored by output jitter / input jitter.  Here the clip, the stabilized output and the scoring all stay on the GPU;
// a worker thread owns its stabilizer, its scoring aligner and its output buffer (handles are independent).
// Several GPUs (--devices a,b,...|all): the clip is uploaded once per device slot, worker t works on slot t mod G with handles
// created on that slot's device -- the reference's independence model (grid_search_align.cpp:159-210: one stabilizer per worker,
// halide_set_num_threads(1); here one stream per handle), no exchange between devices, one shared work counter on the host.
#pragma once
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <memory>
#include <iostream>
#include <mutex>
#include <thread>
#include "harness.hpp"

namespace vsh {

struct GridCombo {
    std::string label;
    vs_stabilizer_params params;
};
struct GridResult {
    int best = -1;
    double best_ratio = 1e9;
    size_t evaluated = 0, skipped = 0;
    std::vector<double> ratios;          // per combination (NaN: skipped) -- the same whatever the device split
    std::vector<double> slot_seconds;    // per device slot: the time its workers spent inside combinations (summed over its workers)
    std::vector<size_t> slot_done;       // per device slot: combinations evaluated
};

inline GridResult run_grid(const std::vector<std::unique_ptr<DeviceClip>>& clips, double input_jitter, const std::vector<GridCombo>& combos, int jobs) {
    GridResult res;
    const int G = (int)clips.size();
    res.ratios.assign(combos.size(), std::nan(""));
    res.slot_seconds.assign((size_t)G, 0.0);
    res.slot_done.assign((size_t)G, 0);
    const DeviceClip& clip = *clips[0];
    const size_t total = combos.size();
    std::atomic<size_t> next{0}, done{0}, skipped{0};
    std::mutex mu;
    const auto t0 = std::chrono::steady_clock::now();
    const int w = clip.fmt.w, h = clip.fmt.h, fmt = vs_format_of(clip.fmt), n = clip.frames;
    const size_t esz = clip.fmt.bits > 8 ? 2 : 1;
    std::string failure;

    auto worker = [&](int t) {
        const int slot = t % G;
        const DeviceClip& clip = *clips[(size_t)slot];
        const int device = clip.device;
        try {
            hip_check(hipSetDevice(device), "hipSetDevice");
            vs_aligner* scorer = vs_aligner_create(nullptr, device);
            if (!scorer) throw std::runtime_error(std::string("vs_aligner_create: ") + vs_last_error());
            DeviceBuffer out;
            std::vector<int32_t> has_output((size_t)n);
            for (;;) {
                const size_t idx = next.fetch_add(1);
                if (idx >= total) break;
                const GridCombo& c = combos[idx];
                const auto c0 = std::chrono::steady_clock::now();
                vs_stabilizer* stab = vs_stabilizer_create(&c.params, device);
                if (!stab) {      // a parameter set this build rejects (phase_correlate): report and move on
                    skipped++;
                    std::lock_guard<std::mutex> lk(mu);
                    std::cout << "[skipped] " << c.label << "  (" << vs_last_error() << ")" << std::endl;
                    continue;
                }
                const int crop = c.params.crop_pixels > 0 ? c.params.crop_pixels : 0;
                const int ow = w - 2 * crop, oh = h - 2 * crop;
                const size_t out_elems = (size_t)ow * oh * 3;
                if (out.bytes < (size_t)n * out_elems * esz) out.reset((size_t)n * out_elems * esz);
                int rw = 0, rh = 0;
                const int produced = vs_stabilizer_process_batch(stab, clip.buf.ptr, clip.frame_elems(), n, w, h, w * 3, fmt, VS_MEM_DEVICE,
                                                                 out.ptr, out_elems, has_output.data(), &rw, &rh);
                vs_stabilizer_destroy(stab);
                if (produced == VS_ERR_UNSUPPORTED) {   // a mode this build does not have: report and move on
                    skipped++;
                    std::lock_guard<std::mutex> lk(mu);
                    std::cout << "[skipped] " << c.label << "  (" << vs_last_error() << ")" << std::endl;
                    continue;
                }
                vs_check(produced, "vs_stabilizer_process_batch");
                if (produced < 2) { skipped++; continue; }          // grid_search_align.cpp:173
                int first = 0;
                while (first < n && !has_output[first]) first++;     // outputs start after `lag` frames and are contiguous
                const double out_jitter = measure_jitter(scorer, static_cast<const uint8_t*>(out.ptr) + (size_t)first * out_elems * esz,
                                                         out_elems, produced, ow, oh, fmt);
                const double ratio = out_jitter / input_jitter;
                const size_t finished = ++done;
                std::lock_guard<std::mutex> lk(mu);
                res.ratios[idx] = ratio;
                res.slot_seconds[(size_t)slot] += std::chrono::duration<double>(std::chrono::steady_clock::now() - c0).count();
                res.slot_done[(size_t)slot]++;
                const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
                std::cout << "[" << finished << "/" << total << "] " << c.label << "  outJit=" << out_jitter << "  ratio=" << ratio
                          << "  elapsed=" << sec << "s";
                if (G > 1) std::cout << "  slot=" << slot << " dev=" << device;
                // ties go to the lower combination index, so the winner does not depend on which worker finished first
                if (ratio < res.best_ratio || (ratio == res.best_ratio && (int)idx < res.best)) {
                    res.best_ratio = ratio;
                    res.best = (int)idx;
                    std::cout << "  ** new best **";
                }
                std::cout << std::endl;
            }
            vs_aligner_destroy(scorer);
        } catch (const std::exception& e) {
            std::lock_guard<std::mutex> lk(mu);
            failure = e.what();
            next = total;     // stop handing out work
        }
    };
    std::vector<std::thread> threads;
    jobs = std::max(jobs, G);                    // at least one worker per device slot
    for (int i = 0; i < jobs; i++) threads.emplace_back(worker, i);
    for (auto& t : threads) t.join();
    if (!failure.empty()) throw std::runtime_error(failure);
    res.evaluated = done;
    res.skipped = skipped;
    if (G > 1) {
        const double wall = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        std::cout << "Device split: " << G << " slots, " << jobs << " workers, wall " << wall << " s" << std::endl;
        for (int g = 0; g < G; g++)
            std::cout << "  slot " << g << " (device " << clips[(size_t)g]->device << "): " << res.slot_done[(size_t)g] << " combinations, "
                      << res.slot_seconds[(size_t)g] << " worker-seconds" << std::endl;
    }
    return res;
}
// every combination's ratio in index order, full precision: what a device split must reproduce (--dump-ratios)
inline void dump_ratios(const GridResult& r) {
    for (size_t i = 0; i < r.ratios.size(); i++) std::printf("RATIO %zu %.17g\n", i, r.ratios[i]);
}

// load the clip, put it in HBM -- once per device slot --, score the input (grid_search_align.cpp:121-132)
inline bool prepare(const GridArgs& args, std::vector<std::unique_ptr<DeviceClip>>& clips, double& input_jitter) {
    vsio::Clip host;
    std::string err;
    if (!vsio::load_clip(args.video, host, err, args.max_frames)) { std::cerr << "Cannot open " << args.video << " (" << err << ")" << std::endl; return false; }
    if (host.frames < 2) { std::cerr << "Video too short." << std::endl; return false; }
    const std::vector<int> slots = args.slots();
    for (int d : slots) if (d < 0 || vs_device_count() <= d) { std::cerr << "No HIP device " << d << std::endl; return false; }
    clips.clear();
    for (int d : slots) {
        clips.emplace_back(new DeviceClip());
        clips.back()->upload(host, d);
    }
    const DeviceClip& clip = *clips[0];
    hip_check(hipSetDevice(clip.device), "hipSetDevice");
    vs_aligner* a = vs_aligner_create(nullptr, clip.device);
    if (!a) { std::cerr << "vs_aligner_create: " << vs_last_error() << std::endl; return false; }
    input_jitter = measure_jitter(a, clip.buf.ptr, clip.frame_elems(), clip.frames, clip.fmt.w, clip.fmt.h, vs_format_of(clip.fmt));
    vs_aligner_destroy(a);
    std::cout << "Input median jitter: " << input_jitter << " px" << std::endl;
    return true;
}

}  // namespace vsh
