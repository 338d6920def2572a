// video_io.hpp -- codec-free clip I/O for the harness programs.
//
// The reference's drivers read .mp4 through cv::VideoCapture and write x264 through cv::VideoWriter
// (video_test.cpp:63-90, eval_jitter.cpp:29-47, grid_search_align.cpp:102-124).  Neither OpenCV nor a codec
// exists here, so the harness speaks the two uncompressed formats every encoder can produce:
//   * YUV4MPEG2 (.y4m): C420* / C422 / C444 / Cmono, 8 bit or pN (10/12/16 bit little-endian), progressive;
//     `ffmpeg -i in.mp4 out.y4m`.
//   * raw interleaved BGR (.bgr, 8 bit): size taken from a `_<W>x<H>` suffix of the file name or given by the caller;
//     `ffmpeg -i in.mp4 -f rawvideo -pix_fmt bgr24 clip_1920x1080.bgr`.
// Frames are handed to the library as interleaved BGR (u8, or u16 for deeper clips), the layout of the reference's
// CV_8UC3 cv::Mat.  YCbCr <-> BGR is BT.601 limited range in 8.8 fixed point (the integer form swscale and OpenCV
// use for untagged material); chroma is replicated on read and box-averaged on write.  Plain C++17, no device code.
#pragma once
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

namespace vsio {

inline int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// one pixel, `bits`-deep samples (8..16); limited range with the offsets scaled to the depth
inline void ycbcr_to_bgr(int y, int cb, int cr, int bits, int& b, int& g, int& r) {
    const int sh = bits - 8, maxv = (1 << bits) - 1;
    const int c = y - (16 << sh), d = cb - (128 << sh), e = cr - (128 << sh);
    r = clampi((298 * c + 409 * e + 128) >> 8, 0, maxv);
    g = clampi((298 * c - 100 * d - 208 * e + 128) >> 8, 0, maxv);
    b = clampi((298 * c + 516 * d + 128) >> 8, 0, maxv);
}
inline void bgr_to_ycbcr(int b, int g, int r, int bits, int& y, int& cb, int& cr) {
    const int sh = bits - 8, maxv = (1 << bits) - 1;
    y = clampi(((66 * r + 129 * g + 25 * b + 128) >> 8) + (16 << sh), 0, maxv);
    cb = clampi(((-38 * r - 74 * g + 112 * b + 128) >> 8) + (128 << sh), 0, maxv);
    cr = clampi(((112 * r - 94 * g - 18 * b + 128) >> 8) + (128 << sh), 0, maxv);
}

enum class Chroma { Mono, C420, C422, C444 };
constexpr int kMaxSide = 32768;       // (the engine's tile coordinates are 16-bit; also keeps w * h * 3 * 2 far inside size_t and the (w + 1) / 2 of the chroma planes inside int)

struct Format {
    int w = 0, h = 0;
    int fps_num = 30, fps_den = 1;   // video_test.cpp:69-73 falls back to 30 fps when the container has none
    int bits = 8;
    Chroma chroma = Chroma::C420;
    int cw() const { return chroma == Chroma::C444 ? w : (chroma == Chroma::Mono ? 0 : (w + 1) / 2); }
    int ch() const { return chroma == Chroma::C420 ? (h + 1) / 2 : (chroma == Chroma::Mono ? 0 : h); }
    size_t sample_bytes() const { return bits > 8 ? 2 : 1; }
    size_t frame_samples() const { return (size_t)w * h + 2 * (size_t)cw() * ch(); }
    size_t bgr_elems() const { return (size_t)w * h * 3; }
};

inline bool ends_with(const std::string& s, const std::string& suf) {
    return s.size() >= suf.size() && s.compare(s.size() - suf.size(), suf.size(), suf) == 0;
}

// "..._1920x1080.bgr" -> 1920, 1080
inline bool size_from_name(const std::string& path, int& w, int& h) {
    const size_t dot = path.rfind('.');
    const size_t us = path.rfind('_', dot);
    if (us == std::string::npos) return false;
    int a = 0, b = 0;
    if (std::sscanf(path.c_str() + us + 1, "%dx%d", &a, &b) != 2 || a <= 0 || b <= 0 || a > kMaxSide || b > kMaxSide) return false;
    w = a; h = b;
    return true;
}

// Sequential reader.  next() converts one frame to interleaved BGR: uint8_t when fmt.bits == 8, else uint16_t.
class Reader {
public:
    Format fmt;
    std::string error;
    ~Reader() { close(); }
    void close() { if (f_) { std::fclose(f_); f_ = nullptr; } }

    bool open(const std::string& path, int raw_w = 0, int raw_h = 0) {
        close();
        error.clear();
        f_ = std::fopen(path.c_str(), "rb");
        if (!f_) { error = "cannot open " + path; return false; }
        if (ends_with(path, ".y4m")) { raw_ = false; return parse_y4m_header(); }
        if (ends_with(path, ".bgr")) {
            raw_ = true;
            fmt = Format();
            fmt.chroma = Chroma::C444;
            if (raw_w > 0 && raw_h > 0) { fmt.w = raw_w; fmt.h = raw_h; }
            else if (!size_from_name(path, fmt.w, fmt.h)) { error = "raw .bgr needs a _<W>x<H> name suffix or an explicit size: " + path; return false; }
            return true;
        }
        error = "unknown clip type (want .y4m or .bgr): " + path;
        return false;
    }

    // false at end of file (error stays empty) or on a damaged stream (error set)
    bool next(void* bgr_out) {
        if (!f_) { error = "reader is not open"; return false; }
        if (raw_) {
            const size_t n = fmt.bgr_elems();
            const size_t got = std::fread(bgr_out, 1, n, f_);
            if (got == 0) return false;
            if (got != n) { error = "truncated raw frame"; return false; }
            return true;
        }
        char line[256];
        if (!read_line(line, sizeof line)) {            // end of file -- clean only if nothing at all was left
            if (line[0] != 0) error = "trailing bytes without a FRAME marker";
            return false;
        }
        if (std::strncmp(line, "FRAME", 5) != 0) { error = "missing FRAME marker"; return false; }
        const size_t bytes = fmt.frame_samples() * fmt.sample_bytes();
        buf_.resize(bytes);
        if (std::fread(buf_.data(), 1, bytes, f_) != bytes) { error = "truncated y4m frame"; return false; }
        if (fmt.bits == 8) unpack<uint8_t, uint8_t>(reinterpret_cast<const uint8_t*>(buf_.data()), static_cast<uint8_t*>(bgr_out));
        else unpack<uint16_t, uint16_t>(reinterpret_cast<const uint16_t*>(buf_.data()), static_cast<uint16_t*>(bgr_out));
        return true;
    }

private:
    bool read_line(char* line, size_t cap) {
        size_t n = 0;
        int c;
        bool any = false;
        while ((c = std::fgetc(f_)) != EOF) {
            if (c == '\n') { line[n] = 0; return true; }
            any = true;
            if (n + 1 < cap) line[n++] = (char)c;
        }
        line[n] = 0;
        if (any && n == 0) { line[0] = '?'; line[1] = 0; }       // (bytes were there, even if they were NULs: the caller tells a clean end from a damaged one by line[0])
        else if (any && line[0] == 0) line[0] = '?';
        return false;
    }
    bool parse_y4m_header() {
        char line[256];
        if (!read_line(line, sizeof line) || std::strncmp(line, "YUV4MPEG2", 9) != 0) { error = "not a YUV4MPEG2 stream"; return false; }
        fmt = Format();
        for (char* tok = std::strtok(line + 9, " "); tok; tok = std::strtok(nullptr, " ")) {
            switch (tok[0]) {
            case 'W': fmt.w = std::atoi(tok + 1); break;
            case 'H': fmt.h = std::atoi(tok + 1); break;
            case 'F': {
                int a = 0, b = 0;
                if (std::sscanf(tok + 1, "%d:%d", &a, &b) == 2 && a > 0 && b > 0) { fmt.fps_num = a; fmt.fps_den = b; }
                break;
            }
            case 'I': if (tok[1] != 'p' && tok[1] != '?') { error = "interlaced y4m is not supported"; return false; } break;
            case 'C': {
                std::string c(tok + 1);
                const size_t p = c.find('p', 3);      // "420p10", "444p12", not the 'p' of "420paldv"
                if (p != std::string::npos && p + 1 < c.size() && c[p + 1] >= '0' && c[p + 1] <= '9') {
                    fmt.bits = std::atoi(c.c_str() + p + 1);
                    c.resize(p);
                }
                if (c.compare(0, 3, "420") == 0) fmt.chroma = Chroma::C420;
                else if (c == "422") fmt.chroma = Chroma::C422;
                else if (c == "444") fmt.chroma = Chroma::C444;
                else if (c == "mono") fmt.chroma = Chroma::Mono;
                else { error = "unsupported y4m colourspace C" + std::string(tok + 1); return false; }
                break;
            }
            default: break;   // A (aspect), X (comments)
            }
        }
        if (fmt.w <= 0 || fmt.h <= 0) { error = "y4m header without W/H"; return false; }
        if (fmt.w > kMaxSide || fmt.h > kMaxSide) { error = "y4m frame size beyond " + std::to_string(kMaxSide) + " pixels a side"; return false; }
        if (fmt.bits < 8 || fmt.bits > 16) { error = "unsupported y4m sample depth"; return false; }
        return true;
    }
    template <typename S, typename D>
    void unpack(const S* src, D* dst) const {
        const int w = fmt.w, h = fmt.h, cw = fmt.cw(), ch = fmt.ch();
        const S* Y = src;
        const S* U = Y + (size_t)w * h;
        const S* V = U + (size_t)cw * ch;
        const int sx = fmt.chroma == Chroma::C444 ? 0 : 1, sy = fmt.chroma == Chroma::C420 ? 1 : 0;
        const int neutral = 128 << (fmt.bits - 8);
        for (int y = 0; y < h; y++) {
            for (int x = 0; x < w; x++) {
                int cb = neutral, cr = neutral;
                if (fmt.chroma != Chroma::Mono) {
                    const size_t ci = (size_t)(y >> sy) * cw + (x >> sx);
                    cb = U[ci]; cr = V[ci];
                }
                int b, g, r;
                ycbcr_to_bgr(Y[(size_t)y * w + x], cb, cr, fmt.bits, b, g, r);
                D* o = dst + ((size_t)y * w + x) * 3;
                o[0] = (D)b; o[1] = (D)g; o[2] = (D)r;
            }
        }
    }
    FILE* f_ = nullptr;
    bool raw_ = false;
    std::vector<char> buf_;
};

// Sequential writer (.y4m in the given chroma layout, or raw .bgr).
class Writer {
public:
    Format fmt;
    std::string error;
    ~Writer() { close(); }
    void close() { if (f_) { std::fclose(f_); f_ = nullptr; } }
    bool is_open() const { return f_ != nullptr; }

    bool open(const std::string& path, const Format& f) {
        close();
        fmt = f;
        raw_ = ends_with(path, ".bgr");
        if (!raw_ && !ends_with(path, ".y4m")) { error = "unknown clip type (want .y4m or .bgr): " + path; return false; }
        if (raw_ && fmt.bits != 8) { error = "raw .bgr is 8 bit only"; return false; }
        f_ = std::fopen(path.c_str(), "wb");
        if (!f_) { error = "cannot create " + path; return false; }
        if (!raw_) {
            const char* c = fmt.chroma == Chroma::C444 ? "444" : fmt.chroma == Chroma::C422 ? "422" : fmt.chroma == Chroma::Mono ? "mono" : "420jpeg";
            std::string cs = c;
            if (fmt.bits > 8) cs = (fmt.chroma == Chroma::C420 ? std::string("420") : cs) + "p" + std::to_string(fmt.bits);
            std::fprintf(f_, "YUV4MPEG2 W%d H%d F%d:%d Ip A1:1 C%s\n", fmt.w, fmt.h, fmt.fps_num, fmt.fps_den, cs.c_str());
        }
        return true;
    }

    bool write(const void* bgr) {
        if (!f_) { error = "writer is not open"; return false; }
        if (raw_) return std::fwrite(bgr, 1, fmt.bgr_elems(), f_) == fmt.bgr_elems();
        const size_t bytes = fmt.frame_samples() * fmt.sample_bytes();
        buf_.resize(bytes);
        if (fmt.bits == 8) pack<uint8_t>(static_cast<const uint8_t*>(bgr), reinterpret_cast<uint8_t*>(buf_.data()));
        else pack<uint16_t>(static_cast<const uint16_t*>(bgr), reinterpret_cast<uint16_t*>(buf_.data()));
        std::fputs("FRAME\n", f_);
        return std::fwrite(buf_.data(), 1, bytes, f_) == bytes;
    }

private:
    template <typename S>
    void pack(const S* src, S* dst) {
        const int w = fmt.w, h = fmt.h, cw = fmt.cw(), ch = fmt.ch();
        const int maxv = (1 << fmt.bits) - 1;
        S* Y = dst;
        S* U = Y + (size_t)w * h;
        S* V = U + (size_t)cw * ch;
        cb_.assign((size_t)w * h, 0);
        cr_.assign((size_t)w * h, 0);
        for (size_t i = 0; i < (size_t)w * h; i++) {
            int y, cb, cr;
            bgr_to_ycbcr(std::min<int>(src[3 * i], maxv), std::min<int>(src[3 * i + 1], maxv), std::min<int>(src[3 * i + 2], maxv), fmt.bits, y, cb, cr);
            Y[i] = (S)y; cb_[i] = cb; cr_[i] = cr;
        }
        if (fmt.chroma == Chroma::Mono) return;
        const int bx = fmt.chroma == Chroma::C444 ? 1 : 2, by = fmt.chroma == Chroma::C420 ? 2 : 1;
        for (int cy = 0; cy < ch; cy++) {
            for (int cx = 0; cx < cw; cx++) {
                int su = 0, sv = 0, cnt = 0;
                for (int dy = 0; dy < by; dy++) {
                    for (int dx = 0; dx < bx; dx++) {
                        const int x = std::min(cx * bx + dx, w - 1), y = std::min(cy * by + dy, h - 1);
                        su += cb_[(size_t)y * w + x]; sv += cr_[(size_t)y * w + x]; cnt++;
                    }
                }
                U[(size_t)cy * cw + cx] = (S)((su + cnt / 2) / cnt);
                V[(size_t)cy * cw + cx] = (S)((sv + cnt / 2) / cnt);
            }
        }
    }
    FILE* f_ = nullptr;
    bool raw_ = false;
    std::vector<char> buf_;
    std::vector<int> cb_, cr_;
};

// Whole clip in host memory, frames back to back as interleaved BGR (the grid searches load the clip once,
// grid_search_align.cpp:121-124).  data holds uint8_t (bits == 8) or uint16_t elements.
struct Clip {
    Format fmt;
    size_t frames = 0;
    std::vector<uint8_t> data;
    size_t elem_bytes() const { return fmt.bits > 8 ? 2 : 1; }
    size_t frame_bytes() const { return fmt.bgr_elems() * elem_bytes(); }
    const uint8_t* frame(size_t i) const { return data.data() + i * frame_bytes(); }
};

inline bool load_clip(const std::string& path, Clip& clip, std::string& error, size_t max_frames = 0) {
    Reader r;
    if (!r.open(path)) { error = r.error; return false; }
    clip.fmt = r.fmt;
    clip.frames = 0;
    clip.data.clear();
    const size_t fb = clip.frame_bytes();
    for (;;) {
        if (max_frames && clip.frames == max_frames) break;
        clip.data.resize((clip.frames + 1) * fb);
        if (!r.next(clip.data.data() + clip.frames * fb)) break;
        clip.frames++;
    }
    clip.data.resize(clip.frames * fb);
    if (!r.error.empty()) { error = r.error; return false; }
    return true;
}

}  // namespace vsio
