// facade_test.cpp -- the reference's align_test.cpp (TestSimilarityTransformsAll + a synthetic AlignImagePair /
// TestImageWarpCorrectness) rewritten with assertions against the drop-in facade headers.
//   usage: facade_test cpu   -> transform algebra only (no device)
//          facade_test gpu   -> + ImageWarp known answer, + align a synthetic pair, + stabilizer lag
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <string>
#include <vector>

#include "../../video_stabilizer_amd/facade/stabilizer.hpp"

static int fails = 0;
#define CHECK(cond) do { if (!(cond)) { std::printf("[FAIL] %s:%d %s\n", __FILE__, __LINE__, #cond); fails++; } } while (0)
static const float EPSILON = 1e-5f;                                    // align_test.cpp:249
static bool nearlyEqual(float a, float b) { return std::fabs(a - b) < EPSILON; }

static void TestSimilarityTransformInverse() {                         // align_test.cpp:261-303
    std::vector<SimilarityTransform> ts(4);
    ts[1].A = 0.1f; ts[1].TX = 10.f; ts[1].TY = 20.f;
    ts[2].B = 0.1f; ts[2].TX = 5.f; ts[2].TY = -5.f;
    ts[3].A = 0.05f; ts[3].B = 0.05f; ts[3].TX = 100.f; ts[3].TY = 50.f;
    std::vector<Point> pts = {{0.f, 0.f}, {100.f, 100.f}, {50.f, 200.f}, {-10.f, 30.f}, {1.3f, -2.7f}};
    for (auto& T : ts) {
        SimilarityTransform Tinv = T.inverse();
        for (auto& p : pts) {
            Point u = Tinv.warp(T.warp(p));
            CHECK(nearlyEqual((float)p.x, (float)u.x) && nearlyEqual((float)p.y, (float)u.y));
        }
    }
}
static void TestSimilarityTransformCompose() {                         // align_test.cpp:311-346
    SimilarityTransform T1, T2;
    T1.A = 0.1f; T1.TX = 10.f; T1.TY = 20.f;
    T2.B = 0.1f; T2.TX = 5.f; T2.TY = 5.f;
    SimilarityTransform T3 = T1.compose(T2);
    for (Point p : std::vector<Point>{{0.f, 0.f}, {10.f, 20.f}, {50.f, 50.f}, {-10.f, 30.f}}) {
        Point a = T3.warp(p), b = T2.warp(T1.warp(p));
        CHECK(nearlyEqual((float)a.x, (float)b.x) && nearlyEqual((float)a.y, (float)b.y));
    }
}
static void TestRandomized() {                                         // align_test.cpp:444-601
    std::mt19937 rng(12345);
    std::uniform_real_distribution<double> dA(-0.3, 0.3), dB(-0.2, 0.2), dT(-50, 50), dP(-100, 100);
    for (int i = 0; i < 50; i++) {
        SimilarityTransform T, U;
        T.A = dA(rng); T.B = dB(rng); T.TX = dT(rng); T.TY = dT(rng);
        U.A = dA(rng); U.B = dB(rng); U.TX = dT(rng); U.TY = dT(rng);
        for (int k = 0; k < 10; k++) {
            Point p{dP(rng), dP(rng)};
            Point u = T.inverse().warp(T.warp(p));
            CHECK(std::fabs(u.x - p.x) < EPSILON && std::fabs(u.y - p.y) < EPSILON);
            Point a = T.compose(U).warp(p), b = U.warp(T.warp(p));
            CHECK(std::fabs(a.x - b.x) < EPSILON && std::fabs(a.y - b.y) < EPSILON);
        }
        SimilarityTransform I = T.compose(T.inverse());
        CHECK(std::fabs(I.A) < EPSILON && std::fabs(I.B) < EPSILON && std::fabs(I.TX) < EPSILON && std::fabs(I.TY) < EPSILON);
    }
}
static void TestImageWarpCorrectness() {                               // align_test.cpp:358-400, exact instead of +-0.5 px
    vs::Buffer<uint8_t> in(64, 64);
    for (int y = 20; y < 30; y++) for (int x = 20; x < 30; x++) in(x, y) = 255;
    SimilarityTransform T; T.TX = 5; T.TY = 7;
    vs::Buffer<float> out(64, 64);
    CHECK(ImageWarp(in, T.inverse(), out));
    for (int y = 0; y < 64; y++) for (int x = 0; x < 64; x++)
        CHECK(out(x, y) == ((x >= 25 && x < 35 && y >= 27 && y < 37) ? 255.f : 0.f));
}
// A test double with the part of Halide::Runtime::Buffer<T>'s interface that imgproc.cpp uses (HalideBuffer.h: data(), width(),
// height(), channels(), dimensions(), dim(i).stride() / .extent(), (w, h[, c]) constructors, planar dense allocation): the
// operator templates of the facade must give the same bytes through it as through vs::Buffer<T>.
namespace halide_like {
struct Dim { int e, s, m = 0; int extent() const { return e; } int stride() const { return s; } int min() const { return m; } };
template <typename T, int Dims = -1, int InClassDimStorage = 4>
class Buffer {
public:
    Buffer() = default;
    Buffer(int w) : d_{{w, 1}}, own_((size_t)w) {}
    Buffer(int w, int h) : d_{{w, 1}, {h, w}}, own_((size_t)w * h) {}
    Buffer(int w, int h, int c) : d_{{w, 1}, {h, w}, {c, w * h}}, own_((size_t)w * h * c) {}
    int dimensions() const { return (int)d_.size(); }
    Dim dim(int i) const { return d_[i]; }
    int width() const { return d_.size() > 0 ? d_[0].e : 1; }
    int height() const { return d_.size() > 1 ? d_[1].e : 1; }
    int channels() const { return d_.size() > 2 ? d_[2].e : 1; }
    T* data() { return own_.data(); }
    const T* data() const { return own_.data(); }
    T& operator()(int x, int y = 0, int c = 0) { return own_[((size_t)c * height() + y) * width() + x]; }
    // what Halide::Runtime::Buffer offers beyond the dense case: crops (non-zero min), strided views, the host-dirty flag
    void set_min(int d, int m) { d_[d].m = m; }
    void set_stride(int d, int st) { d_[d].s = st; }
    void set_host_dirty(bool v = true) { dirty_ = v; }
    bool host_dirty() const { return dirty_; }
private:
    bool dirty_ = false;
    std::vector<Dim> d_;
    std::vector<T> own_;
};
}  // namespace halide_like

template <typename U8, typename U16, typename F32, typename F64>
static std::vector<double> run_chain(const std::vector<uint8_t>& g0, const std::vector<uint8_t>& g1, int w, int h) {
    // the per-level sequence of alignment.cpp:220-276 + one sparse_warpdiff / sparse_ica call, kernel by kernel
    U8 a(w, h), b(w, h), half(w / 2, h / 2);
    for (int y = 0; y < h; y++) for (int x = 0; x < w; x++) { a(x, y) = g0[(size_t)y * w + x]; b(x, y) = g1[(size_t)y * w + x]; }
    std::vector<double> sig;
    CHECK(PyrDown(a, half));
    for (int y = 0; y < h / 2; y += 7) for (int x = 0; x < w / 2; x += 5) sig.push_back(half(x, y));
    F32 gx(w, h), gy(w, h);
    CHECK(GradXY(a, gx, gy));
    int ts = 0;
    U16 lmx, lmy;
    CHECK(GradArgMax(gx, gy, ts, lmx, lmy));
    F32 jx, jy;
    CHECK(SparseJacobian(gx, gy, lmx, lmy, jx, jy));
    SimilarityTransform T; T.TX = 0.75; T.TY = -0.5; T.A = 0.001;
    U16 wdx, wdy;
    CHECK(SparseWarpDiff(b, a, lmx, T, wdx));
    CHECK(SparseWarpDiff(b, a, lmy, T, wdy));
    const int nt = lmx.width() * lmx.height();
    sig.push_back(ts); sig.push_back(nt);
    for (int i = 0; i < nt; i += 3) { sig.push_back(lmx.data()[i]); sig.push_back(lmy.data()[nt + i]); sig.push_back(jx.data()[i]); sig.push_back(wdx.data()[i]); sig.push_back(wdy.data()[i]); }
    // "selected" = every tile (n, 2) / (n, 4): the tables already have that layout
    U16 selx(nt, 2), sely(nt, 2);
    F32 sjx(nt, 4), sjy(nt, 4);
    for (int i = 0; i < 2 * nt; i++) { selx.data()[i] = lmx.data()[i]; sely.data()[i] = lmy.data()[i]; }
    for (int i = 0; i < 4 * nt; i++) { sjx.data()[i] = jx.data()[i]; sjy.data()[i] = jy.data()[i]; }
    F64 jtr;
    CHECK(SparseICA(b, a, selx, sely, sjx, sjy, T, jtr));
    for (int i = 0; i < 4; i++) sig.push_back(jtr.data()[i]);
    F32 warped(w, h);
    CHECK(ImageWarp(a, T, warped));
    for (int y = 3; y < h; y += 11) for (int x = 2; x < w; x += 9) sig.push_back(warped(x, y));
    if constexpr (vs::has_host_dirty<F32>::value)          // outputs written through the host pointer are marked for buffer classes that track it
        CHECK(half.host_dirty() && gx.host_dirty() && gy.host_dirty() && lmx.host_dirty() && jy.host_dirty() && wdx.host_dirty() &&
              jtr.host_dirty() && warped.host_dirty() && !a.host_dirty());
    return sig;
}

static std::vector<uint8_t> texture(int w, int h, double dx, double dy) {   // smooth synthetic BGR frame, shifted by (dx,dy)
    std::vector<uint8_t> f((size_t)w * h * 3);
    for (int y = 0; y < h; y++) for (int x = 0; x < w; x++) {
        double u = x + dx, v = y + dy;
        double g = 128 + 50 * std::sin(u * 0.11) * std::cos(v * 0.07) + 40 * std::sin((u + v) * 0.045) + 30 * std::cos(u * 0.031 - v * 0.052);
        int q = (int)std::floor(g + 0.5); q = q < 0 ? 0 : (q > 255 ? 255 : q);
        f[((size_t)y * w + x) * 3] = (uint8_t)q; f[((size_t)y * w + x) * 3 + 1] = (uint8_t)q; f[((size_t)y * w + x) * 3 + 2] = (uint8_t)q;
    }
    return f;
}
static void AlignImagePair() {                                         // align_test.cpp:625-691 on synthetic frames
    const int w = 640, h = 480;
    auto a = texture(w, h, 0, 0), b = texture(w, h, 2.5, -1.75);
    VideoAligner aligner;
    SimilarityTransform t1, t2;
    CHECK(!aligner.AlignNextFrame(a.data(), w, h, t1));                // first call: false
    CHECK(aligner.AlignNextFrame(b.data(), w, h, t2));
    std::printf("Alignment successful. Transform = %s\n", t2.toString().c_str());
    CHECK(std::fabs(std::fabs(t2.TX) - 2.5) < 0.3 && std::fabs(std::fabs(t2.TY) - 1.75) < 0.3);
    VideoStabilizerParams sp; sp.lag = 3; sp.smoother_memory = 1; sp.crop_pixels = 8;
    VideoStabilizer st(sp);
    int ow = 0, oh = 0, produced = 0;
    for (int i = 0; i < 6; i++) {
        auto f = texture(w, h, 0.4 * i, -0.3 * i);
        auto out = st.processFrame(f.data(), w, h, ow, oh);
        if (i < 3) CHECK(out.empty()); else { CHECK(!out.empty() && ow == w - 16 && oh == h - 16); produced++; }
    }
    CHECK(produced == 3);
}

// a cropped or strided Halide buffer must be refused, not written as if it were dense -- inputs and outputs alike, including an
// existing output whose shape already matches (no device is touched: the checks come first)
static void TestViewsAreRefused() {
    using U8 = halide_like::Buffer<uint8_t>; using F32 = halide_like::Buffer<float>; using U16 = halide_like::Buffer<uint16_t>;
    U8 in(64, 48), out(32, 24), crop(32, 24), strided(32, 24);
    crop.set_min(0, 4);
    strided.set_stride(1, 40);
    CHECK(!PyrDown(in, crop) && !PyrDown(in, strided) && !PyrDown(crop, out));
    F32 gx(64, 48), gy(64, 48), gcrop(64, 48);
    gcrop.set_min(1, 2);
    CHECK(!GradXY(in, gcrop, gy) && !GradXY(in, gx, gcrop));
    U16 lmx(32, 24, 2), lmy(32, 24, 2);                   // 64x48 -> tile size 2 -> 32 x 24 tiles: shapes match, no reallocation
    lmy.set_min(0, 1);
    int ts = 0;
    CHECK(!GradArgMax(gx, gy, ts, lmx, lmy) && ts == 2);
    U16 lm2(32, 24, 2), lmok(32, 24, 2);
    F32 jx(32, 24, 4), jy(32, 24, 4);
    jy.set_stride(2, 32 * 24 + 1);
    CHECK(!SparseJacobian(gx, gy, lm2, lmok, jx, jy));
    U16 wd(32, 24);
    wd.set_min(1, 1);
    SimilarityTransform T;
    CHECK(!SparseWarpDiff(in, in, lm2, T, wd));
    F32 warped(64, 48);
    warped.set_stride(1, 70);
    CHECK(!ImageWarp(in, T, warped));
    CHECK(!out.host_dirty() && !gx.host_dirty());         // nothing was written
}

static void TestOperatorsThroughBothBufferClasses() {
    const int w = 320, h = 240;
    auto c0 = texture(w, h, 0, 0), c1 = texture(w, h, 0.75, -0.5);
    std::vector<uint8_t> g0((size_t)w * h), g1((size_t)w * h);
    for (size_t i = 0; i < g0.size(); i++) { g0[i] = c0[3 * i]; g1[i] = c1[3 * i]; }
    const auto a = run_chain<vs::Buffer<uint8_t>, vs::Buffer<uint16_t>, vs::Buffer<float>, vs::Buffer<double>>(g0, g1, w, h);
    const auto b = run_chain<halide_like::Buffer<uint8_t>, halide_like::Buffer<uint16_t>, halide_like::Buffer<float>, halide_like::Buffer<double>>(g0, g1, w, h);
    CHECK(a.size() > 1000 && a.size() == b.size());
    bool same = a.size() == b.size();
    for (size_t i = 0; same && i < a.size(); i++) same = a[i] == b[i];
    CHECK(same);
}

int main(int argc, char** argv) {
    const std::string mode = argc > 1 ? argv[1] : "cpu";
    TestSimilarityTransformInverse();
    TestSimilarityTransformCompose();
    TestRandomized();
    TestViewsAreRefused();
    if (mode == "gpu") {
        TestImageWarpCorrectness();
        TestOperatorsThroughBothBufferClasses();
        AlignImagePair();
    }
    std::printf(fails ? "FAILED (%d)\n" : "ALL PASS\n", fails);
    return fails ? 1 : 0;
}
