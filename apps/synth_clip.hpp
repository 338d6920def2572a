// synth_clip.hpp -- the seeded synthetic clip of the C++ harness programs (vs_latency, vs_many_clips): the recipe of the Python
// synthetic clips (video_stabilizer_amd/synth.py, SURVEY 8(d): three octaves of value noise + flat rectangles, a slow pan with
// independent per-frame jitter), written for the host in plain C++ so that the programs need nothing but the C ABI and the HIP runtime.
#pragma once
#include <cmath>
#include <cstdint>
#include <vector>

namespace vssynth {

// Seeded texture with detail at every scale (three octaves of value noise + flat rectangles: the same recipe as the Python synthetic clips),
// so that every tile of every pyramid level has a gradient to lock onto.
inline uint64_t mix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}
inline double lattice(int octave, int iy, int ix) { return (double)(mix64(octave * 0x9E3779B1ull + iy * 0x1F123BB5ull + ix * 0x5BD1E995ull) >> 40) / (double)(1 << 24); }

inline std::vector<float> base_texture(int tw, int th) {
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

constexpr int kMargin = 128;
// The texture seen through a camera under the similarity (1+A, B, TX, TY) about the frame centre, bilinear.
inline void make_frame(std::vector<uint8_t>& f, const std::vector<float>& tex, int w, int h, double A, double B, double TX, double TY) {
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


// frame i of the clip `seed`: slow pan + independent jitter per frame: +-4 px, +-0.002 rotation, +-0.001 scale
inline void make_clip_frame(std::vector<uint8_t>& f, const std::vector<float>& tex, int w, int h, int i, uint64_t seed = 0) {
    auto u = [&](int k) { return (double)(mix64(977 * (uint64_t)i + (uint64_t)k + 7919 * seed) >> 11) / (double)(1ull << 53) * 2 - 1; };
    make_frame(f, tex, w, h, 0.001 * u(0), 0.002 * u(1), 0.5 * i + 4.0 * u(2), 4.0 * u(3));
}

}  // namespace vssynth
