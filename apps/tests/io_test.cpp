// CPU unit test of apps/video_io.hpp and apps/jitter.hpp (built by apps/Makefile, run by tests/test_apps_cpu.py).
// usage: io_test <scratch_dir>   -> prints "ok <checks>" and exits 0, or the first failed check and exits 1.
#include <cmath>
#include <iostream>
#include <random>
#include "jitter.hpp"
#include "video_io.hpp"

static int checks = 0;
#define CHECK(cond) do { checks++; if (!(cond)) { std::cerr << "FAILED line " << __LINE__ << ": " #cond << std::endl; return 1; } } while (0)

template <typename T>
static std::vector<uint8_t> smooth_clip(int w, int h, int frames, int maxv) {
    std::vector<uint8_t> bytes((size_t)w * h * 3 * frames * sizeof(T));
    T* p = reinterpret_cast<T*>(bytes.data());
    for (int f = 0; f < frames; f++)
        for (int y = 0; y < h; y++)
            for (int x = 0; x < w; x++)
                for (int c = 0; c < 3; c++)
                    *p++ = (T)((maxv * ((x + 2 * f) * (c + 1) + y * (3 - c))) / ((w + 2 * frames) * (c + 1) + h * (3 - c)));
    return bytes;
}

template <typename T>
static int roundtrip(const std::string& path, vsio::Chroma chroma, int bits, int w, int h, int tol, bool gray_only) {
    const int frames = 3, maxv = (1 << bits) - 1;
    std::vector<uint8_t> src = smooth_clip<T>(w, h, frames, maxv);
    if (gray_only) {
        T* p = reinterpret_cast<T*>(src.data());
        for (size_t i = 0; i < (size_t)w * h * frames; i++) p[3 * i + 1] = p[3 * i + 2] = p[3 * i];
    }
    vsio::Format f;
    f.w = w; f.h = h; f.bits = bits; f.chroma = chroma; f.fps_num = 25; f.fps_den = 1;
    const size_t fb = f.bgr_elems() * sizeof(T);
    {
        vsio::Writer wr;
        CHECK(wr.open(path, f));
        for (int i = 0; i < frames; i++) CHECK(wr.write(src.data() + i * fb));
    }
    vsio::Clip clip;
    std::string err;
    CHECK(vsio::load_clip(path, clip, err));
    CHECK(clip.frames == (size_t)frames && clip.fmt.w == w && clip.fmt.h == h && clip.fmt.bits == bits);
    if (!vsio::ends_with(path, ".bgr")) CHECK(clip.fmt.fps_num == 25 && clip.fmt.fps_den == 1 && clip.fmt.chroma == chroma);   // raw has no header
    CHECK(clip.data.size() == src.size());
    const T* a = reinterpret_cast<const T*>(src.data());
    const T* b = reinterpret_cast<const T*>(clip.data.data());
    int worst = 0;
    for (size_t i = 0; i < src.size() / sizeof(T); i++) worst = std::max(worst, std::abs((int)a[i] - (int)b[i]));
    if (worst > tol) { std::cerr << path << ": round-trip error " << worst << " > " << tol << std::endl; return 1; }
    checks++;
    // the frame cap stops early
    CHECK(vsio::load_clip(path, clip, err, 2) && clip.frames == 2);
    return 0;
}

// io_test --probe <clip>: open the clip and read it to the end like vs_video_test does; prints what happened.  Exit code 0 = read to the end,
// 2 = refused with an error message; anything else (a signal) is a bug -- tests/test_apps_cpu.py feeds it damaged files.
static int probe(const std::string& path) {
    vsio::Reader r;
    if (!r.open(path)) { std::printf("refused: %s\n", r.error.c_str()); return 2; }
    if ((size_t)r.fmt.w * r.fmt.h > (size_t)4096 * 4096) { std::printf("header ok: %dx%d %d bit (too large to read here)\n", r.fmt.w, r.fmt.h, r.fmt.bits); return 0; }
    std::vector<uint16_t> frame(r.fmt.bgr_elems());
    int n = 0;
    while (r.next(frame.data())) n++;
    if (!r.error.empty()) { std::printf("refused after %d frames: %s\n", n, r.error.c_str()); return 2; }
    std::printf("read %d frames of %dx%d %d bit\n", n, r.fmt.w, r.fmt.h, r.fmt.bits);
    return 0;
}

int main(int argc, char** argv) {
    if (argc > 2 && std::string(argv[1]) == "--probe") return probe(argv[2]);
    const std::string dir = argc > 1 ? argv[1] : "/tmp";

    // colour conversion: grey axis, primaries inside range, exact neutral chroma
    int y, cb, cr, b, g, r;
    vsio::bgr_to_ycbcr(0, 0, 0, 8, y, cb, cr);        CHECK(y == 16 && cb == 128 && cr == 128);
    vsio::bgr_to_ycbcr(255, 255, 255, 8, y, cb, cr);  CHECK(y == 235 && cb == 128 && cr == 128);
    vsio::ycbcr_to_bgr(16, 128, 128, 8, b, g, r);     CHECK(b == 0 && g == 0 && r == 0);
    vsio::ycbcr_to_bgr(235, 128, 128, 8, b, g, r);    CHECK(b == 255 && g == 255 && r == 255);
    vsio::ycbcr_to_bgr(64, 512, 512, 10, b, g, r);    CHECK(b == 0 && g == 0 && r == 0);
    vsio::bgr_to_ycbcr(1023, 1023, 1023, 10, y, cb, cr); CHECK(cb == 512 && cr == 512);
    vsio::ycbcr_to_bgr(y, cb, cr, 10, b, g, r);       CHECK(b == 1023 && g == 1023 && r == 1023);
    for (int v = 0; v < 256; v++) {                    // every grey level survives 8-bit 4:4:4 within 1
        vsio::bgr_to_ycbcr(v, v, v, 8, y, cb, cr);
        vsio::ycbcr_to_bgr(y, cb, cr, 8, b, g, r);
        CHECK(std::abs(b - v) <= 1 && std::abs(g - v) <= 1 && std::abs(r - v) <= 1);
    }

    // container round trips; odd sizes exercise the (w+1)/2 chroma planes
    if (roundtrip<uint8_t>(dir + "/t444.y4m", vsio::Chroma::C444, 8, 37, 21, 3, false)) return 1;
    if (roundtrip<uint8_t>(dir + "/t420.y4m", vsio::Chroma::C420, 8, 37, 21, 6, false)) return 1;
    if (roundtrip<uint8_t>(dir + "/t422.y4m", vsio::Chroma::C422, 8, 36, 20, 6, false)) return 1;
    if (roundtrip<uint8_t>(dir + "/tmono.y4m", vsio::Chroma::Mono, 8, 37, 21, 1, true)) return 1;
    if (roundtrip<uint16_t>(dir + "/t420p10.y4m", vsio::Chroma::C420, 10, 38, 22, 24, false)) return 1;
    if (roundtrip<uint16_t>(dir + "/t444p10.y4m", vsio::Chroma::C444, 10, 37, 21, 4, false)) return 1;
    if (roundtrip<uint8_t>(dir + "/traw_37x21.bgr", vsio::Chroma::C444, 8, 37, 21, 0, false)) return 1;

    // header variants other tools write: no C tag (4:2:0), 420mpeg2 / 420paldv, FRAME parameters, extra tags
    {
        const int w = 6, h = 4;
        FILE* f = std::fopen((dir + "/hdr.y4m").c_str(), "wb");
        CHECK(f != nullptr);
        std::fprintf(f, "YUV4MPEG2 W%d H%d F30000:1001 Ip A1:1 XYSCSS=420JPEG\n", w, h);
        for (int k = 0; k < 2; k++) {
            std::fputs(k ? "FRAME Ip\n" : "FRAME\n", f);
            for (int i = 0; i < w * h; i++) std::fputc(16 + 20 * k + i, f);
            for (int i = 0; i < 2 * (w / 2) * (h / 2); i++) std::fputc(128, f);
        }
        std::fclose(f);
        vsio::Clip clip;
        std::string err;
        CHECK(vsio::load_clip(dir + "/hdr.y4m", clip, err));
        CHECK(clip.frames == 2 && clip.fmt.chroma == vsio::Chroma::C420 && clip.fmt.fps_num == 30000 && clip.fmt.fps_den == 1001);
        int bb, gg, rr;
        vsio::ycbcr_to_bgr(16 + 20 + 5, 128, 128, 8, bb, gg, rr);
        CHECK(clip.frame(1)[5 * 3] == bb && clip.frame(1)[5 * 3 + 2] == rr);
        vsio::Reader rd;
        for (const char* tag : {"C420mpeg2", "C420paldv", "C420p10", "C444p12", "Cmono"}) {
            f = std::fopen((dir + "/tag.y4m").c_str(), "wb");
            std::fprintf(f, "YUV4MPEG2 W8 H8 F25:1 %s\n", tag);
            std::fclose(f);
            CHECK(rd.open(dir + "/tag.y4m"));
        }
        CHECK(rd.fmt.chroma == vsio::Chroma::Mono && rd.fmt.bits == 8);
        // errors: interlaced, unknown colourspace, not a y4m, truncated frame, raw without size, unknown extension
        f = std::fopen((dir + "/bad.y4m").c_str(), "wb"); std::fprintf(f, "YUV4MPEG2 W8 H8 It\n"); std::fclose(f);
        CHECK(!rd.open(dir + "/bad.y4m") && !rd.error.empty());
        f = std::fopen((dir + "/bad.y4m").c_str(), "wb"); std::fprintf(f, "YUV4MPEG2 W8 H8 C411\n"); std::fclose(f);
        CHECK(!rd.open(dir + "/bad.y4m"));
        f = std::fopen((dir + "/bad.y4m").c_str(), "wb"); std::fprintf(f, "RIFF....\n"); std::fclose(f);
        CHECK(!rd.open(dir + "/bad.y4m"));
        f = std::fopen((dir + "/bad.y4m").c_str(), "wb"); std::fprintf(f, "YUV4MPEG2 W8 H8 C444\nFRAME\nshort"); std::fclose(f);
        CHECK(rd.open(dir + "/bad.y4m"));
        std::vector<uint8_t> px(8 * 8 * 3);
        CHECK(!rd.next(px.data()) && !rd.error.empty());
        f = std::fopen((dir + "/nosize.bgr").c_str(), "wb"); std::fclose(f);
        CHECK(!rd.open(dir + "/nosize.bgr"));
        CHECK(rd.open(dir + "/nosize.bgr", 4, 4) && !rd.next(px.data()) && rd.error.empty());   // empty clip: clean EOF
        CHECK(!rd.open(dir + "/clip.mp4"));
        int sw = 0, sh = 0;
        CHECK(vsio::size_from_name("a/b/clip_take2_1920x1080.bgr", sw, sh) && sw == 1920 && sh == 1080);
        CHECK(!vsio::size_from_name("clip.bgr", sw, sh));
    }

    // median: odd, even (mean of the middle pair), empty (eval_jitter.cpp:8-19)
    {
        std::vector<double> a = {5, 1, 3}, e = {4, 1, 3, 2}, z;
        CHECK(vsjit::median(a) == 3.0);
        CHECK(vsjit::median(e) == 2.5);
        CHECK(vsjit::median(z) == 0.0);
    }
    // flow statistic: a translation moves every pixel by |t|; a rotation about the centre moves the median pixel by
    // B * (median distance from the centre); identity does not move anything
    {
        vs_transform t{0, 0, 3, -4};
        CHECK(std::fabs(vsjit::flow_median(t, 640, 480) - 5.0) < 1e-6);
        vs_transform id{0, 0, 0, 0};
        CHECK(vsjit::flow_median(id, 640, 480) == 0.0);
        vs_transform rot{0, 0.01, 0, 0};
        std::vector<float> d;
        for (int j = 0; j < 33; j++)
            for (int i = 0; i < 33; i++)
                d.push_back((float)(0.01 * std::hypot((i + 0.5) * 640 / 33 - 320, (j + 0.5) * 480 / 33 - 240)));
        std::nth_element(d.begin(), d.begin() + d.size() / 2, d.end());
        CHECK(std::fabs(vsjit::flow_median(rot, 640, 480) - d[d.size() / 2]) < 1e-6);
        // clip score: median over pairs, frame 0 skipped
        vs_transform seq[5] = {{0, 0, 100, 100}, {0, 0, 1, 0}, {0, 0, 0, 3}, {0, 0, 2, 0}, {0, 0, 0, 4}};
        CHECK(std::fabs(vsjit::jitter(seq, 5, 64, 64) - 2.5) < 1e-6);
        CHECK(vsjit::jitter(seq, 1, 64, 64) == 0.0);
    }
    std::cout << "ok " << checks << std::endl;
    return 0;
}
