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

// Seeded texture with detail at every scale (three octaves of value noise + flat rectangles: the same recipe as the Python synthetic clips),
// so that every tile of every pyramid level has a gradient to lock onto.
static uint64_t mix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}
static double lattice(int octave, int iy, int ix) { return (double)(mix64(octave * 0x9E3779B1ull + iy * 0x1F123BB5ull + ix * 0x5BD1E995ull) >> 40) / (double)(1 << 24); }

static std::vector<float> base_texture(int tw, int th) {
    std::vector<float> tex((size_t)tw * th, 40.f);
    const int periods[3] = {64, 16, 4};
    const double amps[3] = {96, 48, 24};
    for (int o = 0; o < 3; o++)
        for (int y = 0; y < th; y++) {
            const int y0 = y / periods[o];
            const double wy = (double)(y % periods[o]) / periods[o];
            for (int x = 0; x < tw; x++) {
                const int x0 = x / periods[o];
                const double wx = (double)(x % periods[o]) / periods[o];
                const double top = lattice(o, y0, x0) * (1 - wx) + lattice(o, y0, x0 + 1) * wx, bot = lattice(o, y0 + 1, x0) * (1 - wx) + lattice(o, y0 + 1, x0 + 1) * wx;
                tex[(size_t)y * tw + x] += (float)(amps[o] * (top * (1 - wy) + bot * wy));
            }
        }
    uint64_t r = 12345;
    for (int k = 0; k < 200; k++) {
        const int rw = 8 + (int)((r = mix64(r)) % 192), rh = 8 + (int)((r = mix64(r)) % 192);
        const int rx = (int)((r = mix64(r)) % (uint64_t)(tw - rw)), ry = (int)((r = mix64(r)) % (uint64_t)(th - rh));
        const float val = (float)((r = mix64(r)) % 256);
        for (int y = ry; y < ry + rh; y++)
            for (int x = rx; x < rx + rw; x++) tex[(size_t)y * tw + x] = val;
    }
    for (auto& v : tex) v = v < 0 ? 0 : (v > 255 ? 255 : v);
    return tex;
}

static const int kMargin = 128;
// The texture seen through a camera under the similarity (1+A, B, TX, TY) about the frame centre, bilinear.
static void make_frame(std::vector<uint8_t>& f, const std::vector<float>& tex, int w, int h, double A, double B, double TX, double TY) {
    const int tw = w + 2 * kMargin, th = h + 2 * kMargin;
    const double cx = w * 0.5, cy = h * 0.5;
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            const double px = x - cx, py = y - cy;
            const double sx = (1 + A) * px - B * py + cx + TX + kMargin, sy = B * px + (1 + A) * py + cy + TY + kMargin;
            int x0 = (int)std::floor(sx), y0 = (int)std::floor(sy);
            x0 = x0 < 0 ? 0 : (x0 > tw - 2 ? tw - 2 : x0);
            y0 = y0 < 0 ? 0 : (y0 > th - 2 ? th - 2 : y0);
            const float fx = (float)(sx - x0), fy = (float)(sy - y0);
            const float* r0 = &tex[(size_t)y0 * tw + x0];
            const float* r1 = r0 + tw;
            const float g = (r0[0] * (1 - fx) + r0[1] * fx) * (1 - fy) + (r1[0] * (1 - fx) + r1[1] * fx) * fy;
            const uint8_t q = (uint8_t)(g + 0.5f);
            uint8_t* p = &f[((size_t)y * w + x) * 3];
            p[0] = q; p[1] = q; p[2] = q;
        }
}

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
        // slow pan + independent jitter per frame: +-4 px, +-0.002 rotation, +-0.001 scale (the synthetic clips of bench.py)
        auto u = [&](int k) { return (double)(mix64(977 * i + k) >> 11) / (double)(1ull << 53) * 2 - 1; };
        make_frame(host, tex, w, h, 0.001 * u(0), 0.002 * u(1), 0.5 * i + 4.0 * u(2), 4.0 * u(3));
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
