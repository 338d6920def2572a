// vs_many_clips -- BASELINE configs[3] from C++: a batch of independent clips split over the GPUs of one node, host side in C++
// on the C ABI (north_star: "the host stays C++"; the reference's only many-clip precedent is the worker pool of
// grid_search_align.cpp:159-210 -- independent workers, nothing shared but a work counter).
//
//   vs_many_clips [--devices a,b,...|all] [--clips 64] [--frames 120] [--size 1920x1080] [--steps 3] [--min-width 256]
//                 [--warp-mode separable|contracted|exact|cv] [--solver shared|exclusive] [--exact] [--no-warp] [--no-rccl]
// (--warp-mode: the member of the Lanczos2 sampler family, default separable = bench.py's `value`; --exact = --warp-mode exact.
//  VS_MANY_CLIPS_TEST_FAIL_SLOT=g in the environment makes slot g fail on purpose: the exit code must say so, tests/test_apps_gpu.py.)
//
// One thread per device slot (a device may be listed twice: the one-GPU box rehearses the split that way).  Clip i belongs to
// slot i mod G -- the same static round robin as bench.py's ranks --; a slot's clips sit back to back in its HBM.  The timed step
// is bench.py's: vs_aligner_align_clips over all the slot's clips (VS_BATCH_SHARED: the small-footprint solver build), then ONE
// vs_bgr_image_warp_batch of every frame by its measured transform on a second stream, so that the warp of step k runs under the
// alignment of step k + 1.  No exchange between slots: the threads meet at a barrier before and after the timed steps, and the
// line reports every slot's own seconds beside the whole job's (the slowest slot's) -- the C++ point of comparison for the
// Python ranks of `bench.py --gpus N` (c4_strong.per_rank_seconds), where a host-side knee would show first.
// Aggregate: the whole job's numbers are the sum of the slots' frame counters and the maximum of their seconds.  With distinct devices
// they are reduced where the north star puts them -- "RCCL over xGMI only for aggregate throughput reporting": one communicator per
// slot (ncclCommInitAll, one process, one thread per GPU), one ncclAllReduce(sum) of {frames, aligned} and one ncclAllReduce(max) of
// {seconds} after the closing barrier, 24 bytes per slot, nothing on the data path -- and cross-checked against the host-side sum.
// A device listed twice (the one-GPU rehearsal) or a failing RCCL set-up falls back to the host-side aggregate, and the line says so
// ("if RCCL init fails on the box, fall back to host-side aggregation and say so").  --no-rccl skips the attempt.
// The clips are copies of ONE seeded synthetic clip (apps/synth_clip.hpp): every slot aligns real texture and converges like
// bench.py's clips do; for a throughput measurement distinct camera paths per clip add nothing.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <unistd.h>

#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../include/vs_amd.h"
#include "synth_clip.hpp"

namespace {

struct Barrier {                       // (std::barrier is C++20)
    explicit Barrier(int n) : n_(n) {}
    void wait() {
        std::unique_lock<std::mutex> lk(mu_);
        const int gen = gen_;
        if (++count_ == n_) { count_ = 0; gen_++; cv_.notify_all(); }
        else cv_.wait(lk, [&] { return gen != gen_; });
    }
    std::mutex mu_; std::condition_variable cv_; int n_, count_ = 0, gen_ = 0;
};

struct SlotResult { double seconds = 0; long long frames = 0, aligned = 0; std::string error; };

bool parse_devices(const std::string& list, std::vector<int>& out) {
    out.clear();
    if (list == "all") { for (int d = 0; d < vs_device_count(); d++) out.push_back(d); return !out.empty(); }
    size_t pos = 0;
    while (pos <= list.size()) {
        const size_t comma = std::min(list.find(',', pos), list.size());
        const std::string tok = list.substr(pos, comma - pos);
        if (tok.empty() || tok.find_first_not_of("0123456789") != std::string::npos) return false;
        out.push_back(std::atoi(tok.c_str()));
        pos = comma + 1;
    }
    return !out.empty();
}

}  // namespace

int main(int argc, char** argv) {
    std::vector<int> devices = {0};
    int clips = 64, frames = 120, w = 1920, h = 1080, steps = 3, min_width = 256;
    bool warp = true, use_rccl = true;
    std::string warp_mode = "separable", solver = "shared";
    for (int i = 1; i < argc; i++) {
        const std::string a = argv[i];
        auto next = [&]() -> const char* { return i + 1 < argc ? argv[++i] : nullptr; };
        const char* v = nullptr;
        if (a == "--devices") { if (!(v = next()) || !parse_devices(v, devices)) { std::fprintf(stderr, "Error: --devices a,b,...|all\n"); return 1; } }
        else if (a == "--clips") { if (!(v = next())) return 1; clips = std::atoi(v); }
        else if (a == "--frames") { if (!(v = next())) return 1; frames = std::atoi(v); }
        else if (a == "--steps") { if (!(v = next())) return 1; steps = std::atoi(v); }
        else if (a == "--min-width") { if (!(v = next())) return 1; min_width = std::atoi(v); }
        else if (a == "--size") { if (!(v = next()) || std::sscanf(v, "%dx%d", &w, &h) != 2) { std::fprintf(stderr, "Error: --size WxH\n"); return 1; } }
        else if (a == "--exact") warp_mode = "exact";
        else if (a == "--warp-mode") { if (!(v = next())) return 1; warp_mode = v; if (warp_mode == "sep") warp_mode = "separable"; if (warp_mode == "fast") warp_mode = "contracted"; }
        else if (a == "--solver") { if (!(v = next())) return 1; solver = v; }
        else if (a == "--no-warp") warp = false;
        else if (a == "--no-rccl") use_rccl = false;
        else { std::fprintf(stderr, "Usage: %s [--devices a,b,...|all] [--clips N] [--frames M] [--size WxH] [--steps K] [--min-width P] [--warp-mode separable|contracted|exact] [--no-warp] [--no-rccl]\n", argv[0]); return 1; }
    }
    const int G = (int)devices.size();
    if (clips < 1 || frames < 2 || steps < 1 || w < 64 || h < 64) { std::fprintf(stderr, "Error: bad sizes\n"); return 1; }
    if (warp_mode != "separable" && warp_mode != "contracted" && warp_mode != "exact" && warp_mode != "cv") { std::fprintf(stderr, "Error: --warp-mode separable|contracted|exact|cv\n"); return 1; }
    if (solver != "shared" && solver != "exclusive") { std::fprintf(stderr, "Error: --solver shared|exclusive\n"); return 1; }
    const char* fail_env = std::getenv("VS_MANY_CLIPS_TEST_FAIL_SLOT");
    const int fail_slot = fail_env ? std::atoi(fail_env) : -1;
    for (int d : devices) if (d < 0 || d >= vs_device_count()) { std::fprintf(stderr, "Error: no HIP device %d\n", d); return 1; }
    if (vs_abi_version() != VS_ABI_VERSION) { std::fprintf(stderr, "Error: libvs_amd ABI %d, built for %d\n", vs_abi_version(), VS_ABI_VERSION); return 1; }

    // ---- one synthetic clip on the host ----
    const size_t fs = (size_t)w * h * 3;
    std::vector<uint8_t> host(fs * frames);
    {
        const std::vector<float> tex = vssynth::base_texture(w + 2 * vssynth::kMargin, h + 2 * vssynth::kMargin);
        const int nthreads = (int)std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
        std::vector<std::thread> th;
        for (int t = 0; t < nthreads; t++)
            th.emplace_back([&, t] {
                std::vector<uint8_t> f(fs);
                for (int i = t; i < frames; i += nthreads) { vssynth::make_clip_frame(f, tex, w, h, i); std::memcpy(&host[fs * i], f.data(), fs); }
            });
        for (auto& t : th) t.join();
    }

    // ---- RCCL communicators for the report (distinct devices only) ----
    std::vector<ncclComm_t> comms((size_t)G, nullptr);
    std::string aggregate = "host";
    std::string rccl_note;
    if (!use_rccl) rccl_note = "--no-rccl";
    else {
        bool distinct = true;
        for (int i = 0; i < G; i++) for (int j = i + 1; j < G; j++) if (devices[(size_t)i] == devices[(size_t)j]) distinct = false;
        if (!distinct) rccl_note = "a device is listed more than once: one RCCL rank per GPU only";
        else {
            // (RCCL prints its version banner to fd 1 while it comes up: stdout carries the one JSON line, so fd 1 points at stderr meanwhile)
            std::fflush(stdout);
            const int saved = dup(1);
            (void)dup2(2, 1);
            const ncclResult_t r = ncclCommInitAll(comms.data(), G, devices.data());
            std::fflush(stdout);
            (void)dup2(saved, 1);
            (void)close(saved);
            if (r == ncclSuccess) aggregate = "rccl";
            else { rccl_note = std::string("ncclCommInitAll: ") + ncclGetErrorString(r); for (auto& c : comms) c = nullptr; }
        }
    }
    struct Agg { unsigned long long frames, aligned; double seconds; };
    std::vector<Agg> agg((size_t)G, Agg{0, 0, 0.0});
    std::vector<std::string> agg_err((size_t)G);

    std::vector<SlotResult> res((size_t)G);
    Barrier barrier(G);
    // (cv = the reference's own per-frame warp: cv::warpAffine's fixed-point bilinear, constant border, the measured transform as the forward map)
    const int mode = warp_mode == "cv" ? VS_WARP_BILINEAR_CV : (warp_mode == "exact" ? VS_WARP_LANCZOS2 : (warp_mode == "contracted" ? VS_WARP_LANCZOS2_FAST : VS_WARP_LANCZOS2_SEP));
    const int border = warp_mode == "cv" ? VS_BORDER_CONSTANT : VS_BORDER_CLAMP;
    auto slot = [&](int g) {
        SlotResult& r = res[(size_t)g];
        const int dev = devices[(size_t)g];
        int mine = 0;
        for (int c = g; c < clips; c += G) mine++;                  // clip i -> slot i mod G
        uint8_t *in = nullptr, *out = nullptr;
        vs_aligner* a = nullptr;
        hipStream_t ws = nullptr;
        auto fail = [&](const std::string& what) { if (r.error.empty()) r.error = what; };
        const int n = mine * frames;
        std::vector<vs_transform> t((size_t)std::max(n, 1));
        std::vector<int32_t> st((size_t)std::max(n, 1));
        if (mine > 0) {
            if (hipSetDevice(dev) != hipSuccess) fail("hipSetDevice");
            if (r.error.empty() && hipMalloc((void**)&in, fs * n) != hipSuccess) fail("hipMalloc (clips)");
            if (r.error.empty() && warp && hipMalloc((void**)&out, fs * n) != hipSuccess) fail("hipMalloc (outputs)");
            if (r.error.empty() && hipMemcpy(in, host.data(), fs * frames, hipMemcpyHostToDevice) != hipSuccess) fail("hipMemcpy H2D");
            for (int c = 1; r.error.empty() && c < mine; c++)
                if (hipMemcpy(in + fs * frames * c, in, fs * frames, hipMemcpyDeviceToDevice) != hipSuccess) fail("hipMemcpy D2D");
            vs_aligner_params p;
            vs_aligner_params_default(&p);
            p.pyramid_min_width = min_width;
            if (r.error.empty() && !(a = vs_aligner_create(&p, dev))) fail(std::string("vs_aligner_create: ") + vs_last_error());
            if (r.error.empty() && warp && solver == "shared") vs_aligner_set_batch_mode(a, VS_BATCH_SHARED);
            if (r.error.empty() && hipStreamCreateWithFlags(&ws, hipStreamNonBlocking) != hipSuccess) fail("hipStreamCreate");
        }
        auto step = [&]() -> int {
            const int good = vs_aligner_align_clips(a, in, fs, mine, frames, w, h, 3 * w, VS_FMT_BGR8, VS_MEM_DEVICE, nullptr, t.data(), st.data());
            if (good < 0) { fail(std::string("vs_aligner_align_clips: ") + vs_last_error()); return -1; }
            if (warp && vs_bgr_image_warp_batch(in, fs, n, w, h, 3 * w, 3, 8, t.data(), mode, border, 255, out, fs, 3 * w, VS_MEM_DEVICE, ws) < 0) {
                fail(std::string("vs_bgr_image_warp_batch: ") + vs_last_error());
                return -1;
            }
            return good;
        };
        if (g == fail_slot) fail("slot " + std::to_string(g) + " fails on purpose (VS_MANY_CLIPS_TEST_FAIL_SLOT)");
        if (mine > 0 && r.error.empty()) { step(); (void)hipDeviceSynchronize(); }          // warm: allocations, clocks
        barrier.wait();
        const auto t0 = std::chrono::steady_clock::now();
        for (int k = 0; mine > 0 && r.error.empty() && k < steps; k++) {
            const int good = step();
            if (good < 0) break;
            r.frames += n; r.aligned += good;
        }
        if (mine > 0) (void)hipDeviceSynchronize();
        r.seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        barrier.wait();
        if (aggregate == "rccl") {
            // every slot contributes its counters and its seconds; every slot gets the whole job's (slot 0's copy is reported)
            unsigned long long* d_cnt = nullptr; double* d_sec = nullptr;
            const unsigned long long h_cnt[2] = {(unsigned long long)r.frames, (unsigned long long)r.aligned};
            hipStream_t cs = nullptr;
            bool ok = hipSetDevice(dev) == hipSuccess && hipStreamCreateWithFlags(&cs, hipStreamNonBlocking) == hipSuccess &&
                      hipMalloc((void**)&d_cnt, 16) == hipSuccess && hipMalloc((void**)&d_sec, 8) == hipSuccess &&
                      hipMemcpyAsync(d_cnt, h_cnt, 16, hipMemcpyHostToDevice, cs) == hipSuccess &&
                      hipMemcpyAsync(d_sec, &r.seconds, 8, hipMemcpyHostToDevice, cs) == hipSuccess;
            ncclResult_t nr = ncclSuccess;
            if (ok) nr = ncclAllReduce(d_cnt, d_cnt, 2, ncclUint64, ncclSum, comms[(size_t)g], cs);
            if (ok && nr == ncclSuccess) nr = ncclAllReduce(d_sec, d_sec, 1, ncclDouble, ncclMax, comms[(size_t)g], cs);
            unsigned long long o_cnt[2] = {0, 0};
            double o_sec = 0.0;
            ok = ok && nr == ncclSuccess && hipMemcpyAsync(o_cnt, d_cnt, 16, hipMemcpyDeviceToHost, cs) == hipSuccess &&
                 hipMemcpyAsync(&o_sec, d_sec, 8, hipMemcpyDeviceToHost, cs) == hipSuccess && hipStreamSynchronize(cs) == hipSuccess;
            if (ok) agg[(size_t)g] = Agg{o_cnt[0], o_cnt[1], o_sec};
            else agg_err[(size_t)g] = nr != ncclSuccess ? ncclGetErrorString(nr) : "HIP error around the all-reduce";
            if (d_cnt) (void)hipFree(d_cnt);
            if (d_sec) (void)hipFree(d_sec);
            if (cs) (void)hipStreamDestroy(cs);
        }
        if (ws) { (void)vs_stream_retire(ws); (void)hipStreamDestroy(ws); }
        if (a) vs_aligner_destroy(a);
        if (in) (void)hipFree(in);
        if (out) (void)hipFree(out);
    };
    std::vector<std::thread> threads;
    const auto j0 = std::chrono::steady_clock::now();
    for (int g = 0; g < G; g++) threads.emplace_back(slot, g);
    for (auto& t : threads) t.join();
    const double job = std::chrono::duration<double>(std::chrono::steady_clock::now() - j0).count();

    double slowest = 0;
    long long total_frames = 0, total_aligned = 0;
    for (const SlotResult& r : res) {
        if (!r.error.empty()) { std::fprintf(stderr, "Error: %s\n", r.error.c_str()); return 1; }
        slowest = std::max(slowest, r.seconds);
        total_frames += r.frames; total_aligned += r.aligned;
    }
    for (ncclComm_t c : comms) if (c) (void)ncclCommDestroy(c);
    if (aggregate == "rccl") {
        // the collective's answer must be the host-side sum on every slot; anything else is reported and the host-side figures stand
        for (int g = 0; g < G; g++) {
            const Agg& a = agg[(size_t)g];
            if (!agg_err[(size_t)g].empty()) { aggregate = "host"; rccl_note = "all-reduce failed on slot " + std::to_string(g) + ": " + agg_err[(size_t)g]; break; }
            if ((long long)a.frames != total_frames || (long long)a.aligned != total_aligned || a.seconds != slowest) {
                std::fprintf(stderr, "Error: the RCCL aggregate on slot %d (%llu frames, %llu aligned, %.6f s) differs from the host-side one (%lld, %lld, %.6f)\n", g,
                             a.frames, a.aligned, a.seconds, total_frames, total_aligned, slowest);
                return 1;
            }
        }
    }
    std::printf("{\"program\": \"vs_many_clips\", \"host\": \"C++ threads, one per device slot\", \"devices\": [");
    for (int g = 0; g < G; g++) std::printf("%s%d", g ? ", " : "", devices[(size_t)g]);
    std::printf("], \"clips\": %d, \"frames_per_clip\": %d, \"width\": %d, \"height\": %d, \"steps\": %d, \"warp\": \"%s\", \"scaling\": \"strong\", ", clips, frames, w, h,
                steps, warp ? (warp_mode == "cv" ? "cv::warpAffine fixed-point bilinear, constant border" : (warp_mode == "exact" ? "lanczos2" : (warp_mode == "contracted" ? "lanczos2 contracted" : "lanczos2 separable"))) : "none");
    std::printf("\"solver\": \"%s\", ", solver.c_str());
    std::printf("\"value\": %.2f, \"unit\": \"aligned frames/s\", \"frames_per_s\": %.2f, \"seconds\": %.5f, \"per_slot_seconds\": [", total_aligned / slowest,
                total_frames / slowest, slowest);
    for (int g = 0; g < G; g++) std::printf("%s%.5f", g ? ", " : "", res[(size_t)g].seconds);
    std::printf("], \"per_slot_clips\": [");
    for (int g = 0; g < G; g++) { int m = 0; for (int c = g; c < clips; c += G) m++; std::printf("%s%d", g ? ", " : "", m); }
    std::printf("], \"aligned_per_step\": %lld, \"aggregate\": \"%s\", \"aggregate_note\": \"%s\", \"setup_and_run_seconds\": %.2f}\n", total_aligned / steps,
                aggregate == "rccl" ? "rccl: ncclAllReduce sum {frames, aligned} + max {seconds} over one communicator per GPU, equal to the host-side sums" : "host-side sums",
                rccl_note.c_str(), job);
    return 0;
}
