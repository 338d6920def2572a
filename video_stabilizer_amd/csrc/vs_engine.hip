// vs_engine.hip -- engine level of the C ABI: VideoAligner (alignment.cpp:149-704) and
// VideoStabilizer (stabilizer.cpp:9-117) as a device-resident, batched pipeline.
//
// Design (DESIGN.md "Engine"):
//   * Gray pyramids of every frame of a batch live in HBM in one slab: slot s, level l at
//     pyr + s*pyr_frame + L[l].img_off.  Slot 0 is the carry-over frame of the previous call,
//     slots 1..n are this call's frames, so every stage is ONE launch over all frames.
//   * Odd frames (in sequence order) are keyframes (alignment.hpp:65-66): their keypoint tables
//     (arg-max coordinates + Jacobians, every level) are produced by the fused keyframe kernel.
//   * Every alignment depends on exactly two consecutive frames, so all frame pairs of a batch are
//     solved concurrently: per level one warpdiff launch, one selection step, one gather launch and
//     one persistent Gauss-Newton launch in which a whole workgroup owns a pair and iterates on the
//     device until it converges / diverges / runs out of iterations -- no host round trip per
//     iteration (the reference makes one Halide call per iteration, alignment.cpp:600-668).
//   * The transform algebra inside the loop is fp64 on the device, written exactly as imgproc.cpp.
#include "vs_internal.hpp"
#include "vs_kernels.hpp"
#include "vs_phase.hpp"
#include "vs_device.hpp"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <mutex>
#include <cmath>
#include <cstring>
#include <deque>
#include <thread>
#include <cstdlib>
#include <future>
#include <string>
#include <vector>

using vsi::set_error;
using namespace vsd;

#define VS_TRY(expr) do { int _r = (expr); if (_r < 0) return _r; } while (0)
#define VS_ARG(cond) do { if (!(cond)) return set_error(VS_ERR_ARG, "bad argument: %s (%s)", #cond, __func__); } while (0)

namespace {

constexpr int kMaxLevels = 16;
// on-device selection keeps (abs_delta,index) u32 + a u16 rank table per tile in LDS: 6 B x tiles <= 160 KB less
// the static part; index fits 16 bits; a thread's chunk fits a 32-bit mask
constexpr int kSelectCap = 26000;

struct LevelDims {
    int w, h, ts, tx, ty, nt, nsel;
    size_t img_off;   // bytes from the start of a frame's pyramid
    size_t lm_off;    // u16 elements from the start of a frame's arg-max table (x-set); y-set follows at +2*nt
    size_t jac_off;   // f32 elements from the start of a frame's Jacobian table (x-set); y-set at +4*nt
};

// per-pair solver state, device resident; copied back once per batch
struct PairState {
    double T[4];
    int32_t status;        // 1 running/aligned, 0 failed
    int32_t fail_reason;   // 2 max iters, 3 over displacement
    int32_t fail_level;
    int32_t pad;
    int32_t iterations[kMaxLevels];
    double condition[kMaxLevels];
    double level_T[kMaxLevels][4];   // the estimate each level ended on (vs_align_info::level_transform)
#ifdef VS_PROFILE_STAMPS
    unsigned long long stamps[kMaxLevels][6];   // diagnostic build only: shader-clock ticks per phase of a level
#endif
};

struct PairDesc {
    int32_t tmpl_slot;   // slot of the even (non-keyframe) frame: ScalePyramid[NonKeyframeIndex]
    int32_t key_slot;    // slot of the odd (keyframe) frame:      ScalePyramid[KeyframeIndex]
};

// latency mode: the descriptors of a small launch travel in the kernel arguments (no upload in front of the launch)
constexpr int kDirectMaxPairs = 16;
struct PairDescPack { PairDesc d[kDirectMaxPairs]; };

struct GnParams {
    double threshold, max_displacement;
    int max_iters;
    int pipeline;   // 1: the one-barrier iteration loop on the LDS-resident level (latency mode, few pairs in flight)
    int stall_helpers;   // test hook (VS_GN_STALL_HELPERS=1): helpers never report back, the leader's bounded wait must expire
    int sel_depth_cap;   // test hook (VS_GN_SELECT_DEPTH=k > 0): the on-device introselect gets k partition rounds instead of 2 lg n, so
                         // ordinary frames take the "libstdc++ would have heap-selected" exit (fail_reason 100 -> host redo)
    int sel_stable;      // VS_SELECT_STABLE: smallest by (abs_delta, tile index), survivors in tile order (stable_select)
};

// ---- phase-correlation start value (alignment.cpp:376-387) -----------------------------------------
// detected_shift of the level-2 images scaled by (1 << PhaseLevel) / float(1 << PyramidLevels) becomes the initial TX,TY
// (negated when the current frame is the keyframe), if the response clears the threshold.
__global__ void vs_k_phase_apply(PairState* __restrict__ states, const vsp::Result* __restrict__ res,
                                 const uint8_t* __restrict__ negate, int n_pairs, double threshold, float scale) {
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= n_pairs) return;
    const vsp::Result r = res[q];
    if (r.response > threshold) {
        double tx = r.dx * scale, ty = r.dy * scale;
        if (negate[q]) { tx = -tx; ty = -ty; }
        states[q].T[2] = tx;
        states[q].T[3] = ty;
    }
}

// ---- batched sparse_warpdiff: generators.cpp:646-700 for every (pair, set) ---------------------
__global__ __launch_bounds__(256) void vs_k_warpdiff_batch(const PairState* __restrict__ states,
                                                           const PairDesc* __restrict__ descs,
                                                           const uint8_t* __restrict__ pyr, size_t pyr_frame,
                                                           size_t img_off, int w, int h,
                                                           const uint16_t* __restrict__ lm_tab, size_t lm_frame,
                                                           size_t lm_off, int nt, uint16_t* __restrict__ wd,
                                                           float* __restrict__ wv, size_t wd_pair) {
    const int p = blockIdx.y, set = blockIdx.z;
    const PairState& st = states[p];
    if (st.status != 1) return;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= nt) return;
    const PairDesc d = descs[p];
    const uint8_t* tmpl = pyr + (size_t)d.tmpl_slot * pyr_frame + img_off;
    const uint8_t* key = pyr + (size_t)d.key_slot * pyr_frame + img_off;
    const uint16_t* lm = lm_tab + (size_t)d.key_slot * lm_frame + lm_off + (size_t)set * 2 * nt;
    double T[4] = {st.T[0], st.T[1], st.T[2], st.T[3]};
    float P[4];
    ul_params_sparse(T, w, h, P);
    const uint32_t xy = ((const uint32_t*)lm)[i];           // {x, y} pair of tile i
    int tile_x = min((int)(xy & 0xffffu), w - 1), tile_y = min((int)(xy >> 16), h - 1);
    float ox = (float)tile_x, oy = (float)tile_y;
    float Wx = (1.0f + P[0]) * ox - P[1] * oy + P[2];
    float Wy = P[1] * ox + (1.0f + P[0]) * oy + P[3];
    float v = lanczos_sample_u8(key, w, h, w, Wx, Wy);
    float diff = fabsf(v - (float)tmpl[(size_t)tile_y * w + tile_x]);
    diff = fminf(fmaxf(diff, 0.0f), 65535.0f);
    wd[(size_t)p * wd_pair + (size_t)set * nt + i] = (uint16_t)diff;
    wv[(size_t)p * wd_pair + (size_t)set * nt + i] = v;   // the first Gauss-Newton iteration samples the same point (PointRecs::r0)
}

// ---- gather of the selected keypoints + Jacobians: alignment.cpp:523-546 ------------------------
// The solver reads one compact record per selected point, built once per level (both sets back to
// back: x-set at [0,nsel), y-set at [nsel,2*nsel)):
//   xy  u32    x | y << 16                       (SelectedPixels, alignment.cpp:530-531)
//   tv  f32    float(template(min(x,w-1), min(y,h-1)))   -- constant over the iterations (generators.cpp:554-556)
//   j   float4 the four Jacobian components     (SelectedJacobian, alignment.cpp:532-534)
// a value that is the same in every lane, moved to the scalar registers (see gn_level)
template <typename T>
__device__ __forceinline__ T* uniform_ptr(T* p) {
    const unsigned long long v = (unsigned long long)p;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return (T*)(((unsigned long long)hi << 32) | lo);
}

//   r0  f32    tv - (the key frame sampled at the point warped by the transform the level STARTS from): the residual of the
//              first Gauss-Newton iteration.  sparse_warpdiff (generators.cpp:672-697) has just computed exactly that sample
//              for every tile -- same image, same fp32 kernel parameters, same point -- so the first sparse_ica pass
//              (generators.cpp:469-498) takes it from here instead of sampling again; the value is the same bit for bit.
struct PointRecs {
    uint32_t* xy;
    float* tv;
    float4* j;
    float* r0;
};
__device__ __forceinline__ PointRecs pair_recs(uint8_t* recs, size_t recs_pair, int p, int nt_cap) {
    uint8_t* base = recs + (size_t)p * recs_pair;
    PointRecs r;
    r.j = (float4*)base;                                      // 2*nt_cap float4
    r.xy = (uint32_t*)(base + (size_t)2 * nt_cap * 16);       // 2*nt_cap u32
    r.tv = (float*)(base + (size_t)2 * nt_cap * 20);          // 2*nt_cap f32
    r.r0 = (float*)(base + (size_t)2 * nt_cap * 24);          // 2*nt_cap f32
    return r;
}
__device__ __forceinline__ void write_rec(const PointRecs& rc, int slot, const uint16_t* __restrict__ lm,
                                          const float* __restrict__ jac, int nt, int t, const uint8_t* __restrict__ tmpl,
                                          int w, int h, const float* __restrict__ wv_set) {
    const uint32_t xy = ((const uint32_t*)lm)[t];
    const int px = (int)(xy & 0xffffu), py = (int)(xy >> 16);
    rc.xy[slot] = xy;
    const float tv = (float)tmpl[(size_t)min(py, h - 1) * w + min(px, w - 1)];
    rc.tv[slot] = tv;
    rc.r0[slot] = tv - wv_set[t];
    rc.j[slot] = ((const float4*)jac)[t];
}

__global__ __launch_bounds__(256) void vs_k_gather_selected(const PairState* __restrict__ states,
                                                            const PairDesc* __restrict__ descs,
                                                            const uint8_t* __restrict__ pyr, size_t pyr_frame,
                                                            size_t img_off, int w, int h,
                                                            const uint16_t* __restrict__ lm_tab, size_t lm_frame,
                                                            size_t lm_off, const float* __restrict__ jac_tab,
                                                            size_t jac_frame, size_t jac_off, int nt, int nsel,
                                                            const int32_t* __restrict__ idx, size_t idx_pair,
                                                            const float* __restrict__ wv,
                                                            uint8_t* __restrict__ recs, size_t recs_pair, int nt_cap) {
    const int p = blockIdx.y, set = blockIdx.z;
    if (states[p].status != 1) return;
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= nsel) return;
    const PairDesc d = descs[p];
    const uint16_t* lm = lm_tab + (size_t)d.key_slot * lm_frame + lm_off + (size_t)set * 2 * nt;
    const float* jac = jac_tab + (size_t)d.key_slot * jac_frame + jac_off + (size_t)set * 4 * nt;
    const uint8_t* tmpl = pyr + (size_t)d.tmpl_slot * pyr_frame + img_off;
    const int t = idx[(size_t)p * idx_pair + (size_t)set * nt + j];
    write_rec(pair_recs(recs, recs_pair, p, nt_cap), set * nsel + j, lm, jac, nt, t, tmpl, w, h, wv + (size_t)p * idx_pair + (size_t)set * nt);
}

// ---- helper workgroups (latency mode) -------------------------------------------------------------------
// When only a few pairs are in flight, a pair is launched as kCoopGroup workgroups: the leader runs the algorithm as
// always, the others take slices of the one embarrassingly parallel pass of the large levels -- sparse_warpdiff of every
// tile (generators.cpp:646-700), which on one CU is bound by that CU's L1 fill rate (every 4-byte window pulls a
// 128-byte line).  The pass is exact per point and every sum stays with the leader, so the results are bit-identical to
// the one-workgroup launch.
// Protocol (one CoopCtrl per pair in device memory, zeroed at allocation; every launch has a new `epoch`, flags hold the
// epoch they were raised in, so nothing is ever reset): helper g raises arrived[g]; at a shared level the leader hands
// slices to the helpers that have arrived BY THEN (assign[g], nslices, T, then t_ready) and never waits for one that has
// not; a helper writes the packed keys of its slice and raises wd_done[g]; the leader reads all keys into LDS and goes on
// alone.  `abort` releases helpers when the pair ends early.  All waits are bounded.
constexpr int kMaxDevices = 64;              // devices the per-device launch caches index; beyond that they are bypassed
constexpr int kCoopGroup = 16;         // workgroups per pair in latency mode (1080p: 8 and 16 alike; 4K: 16 is 3 % faster)
constexpr int kCoopMaxGroup = 16;
constexpr int kCoopMaxPairs = 128;     // helpers for launches of at most this many pairs (16 per pair up to 16 pairs, then as many as keep the launch within one workgroup per CU)
constexpr int kChipCUs = 256;
#ifndef VS_COOP_MIN_TILES
#define VS_COOP_MIN_TILES 4096
#endif
constexpr int kCoopMinTiles = VS_COOP_MIN_TILES;    // levels with at least this many tiles are shared
struct CoopLevel {
    int t_ready, nslices, pad[2];
    int assign[kCoopMaxGroup];         // slice number of helper g at this level, -1: not taking part
    int wd_done[kCoopMaxGroup];
    double T[4];                       // the transform the level starts from
};
struct CoopCtrl {
    int arrived[kCoopMaxGroup];
    int abort;
    int pad[15];
    CoopLevel lv[kMaxLevels];
};
constexpr size_t kCoopCtrlBytes = (sizeof(CoopCtrl) + 255) & ~(size_t)255;
// per pair: CoopCtrl | u32 keys[2 * nt_cap] (sparse_warpdiff, packed for the selection) | f32 values[2 * nt_cap] (the samples) |
// f32 template pixels[2 * nt_cap];
// these arrays are only ever touched with sc1 stores and sc1 loads (a line one XCD's L2 kept from a plain access would go stale)
__host__ __device__ inline size_t coop_pair_bytes(int nt_cap) { return kCoopCtrlBytes + (((size_t)nt_cap * 24 + 255) & ~(size_t)255); }

// ---- fused per-pair aligner: every level of alignment.cpp:390-688 in ONE launch ---------------------
struct FusedLevels {
    int levels;
    int w[kMaxLevels], h[kMaxLevels], nt[kMaxLevels], nsel[kMaxLevels], tx[kMaxLevels];
    unsigned long long img_off[kMaxLevels], lm_off[kMaxLevels], jac_off[kMaxLevels];
};

// The per-pair kernels are compiled for two workgroup sizes (vs_align_kernels.inc).  512 threads is the faster shape in
// every configuration measured (fewer waves behind every block barrier: 239 1080p pairs 0.70 -> 0.62 ms, one pair alone
// 0.445 -> 0.40 ms, 952 pairs 2.50 -> 2.09 ms, 119 4K pairs 1.34 -> 1.29 ms although the 20736-tile selection itself is
// slower with half the threads) and is the default for every level size; VS_SMALL_WG_TILES (build time) sends larger levels
// to the 1024-thread build.
namespace nt1024 {
constexpr int kGnThreads = 1024, kGnVirt = 1024;
#define VS_GN_FUSED_BOUNDS __launch_bounds__(kGnThreads)
#include "vs_align_kernels.inc"
#undef VS_GN_FUSED_BOUNDS
}  // namespace nt1024
#ifndef VS_NT_SMALL
#define VS_NT_SMALL 512
#endif
namespace nt512 {
constexpr int kGnThreads = VS_NT_SMALL, kGnVirt = VS_NT_SMALL;
#define VS_GN_FUSED_BOUNDS __launch_bounds__(kGnThreads)
#include "vs_align_kernels.inc"
#undef VS_GN_FUSED_BOUNDS
}  // namespace nt512
// Small-footprint build of the fused kernel for full batches (DESIGN.md "co-residency"): 256 hardware threads stand in for the
// 512 of nt512 (virtual threads: bit-identical sums), at most 128 VGPRs (4 waves per SIMD) and ~33 KB of LDS (selection arrays
// of one point set; no staged image) -- the footprint of ONE bgr_image_warp workgroup, so a pair's workgroup moves into a CU as
// soon as one warp workgroup leaves it and the alignment of clip k+1 runs under the warp launch of clip k instead of waiting
// for whole CUs to drain.  Levels whose selection arrays (6 B per tile) exceed 32 KB -- a 4K level 0: 20736 tiles -- select on a
// per-pair global scratch (introselect_block_g), up to 128 * 256 tiles.
#if VS_NT_SMALL == 512
namespace nt256v {
// (experiment, -DVS_SHARED_THREADS=512: the co-resident build with all 512 hardware threads at the same 128-VGPR cap -- two waves per SIMD of 128
// registers each instead of one; profiles/r06_shared_solver_512.txt)
#ifndef VS_SHARED_THREADS
#define VS_SHARED_THREADS 256
#endif
constexpr int kGnThreads = VS_SHARED_THREADS, kGnVirt = 512;
#ifndef VS_NT256_MINWAVES
#define VS_NT256_MINWAVES 4
#endif
#ifdef VS_NT256_NUM_VGPR
#define VS_GN_FUSED_BOUNDS __launch_bounds__(VS_SHARED_THREADS) __attribute__((amdgpu_num_vgpr(VS_NT256_NUM_VGPR)))
#else
#define VS_GN_FUSED_BOUNDS __launch_bounds__(VS_SHARED_THREADS, VS_NT256_MINWAVES)
#endif
#include "vs_align_kernels.inc"
#undef VS_GN_FUSED_BOUNDS
}  // namespace nt256v
#define VS_HAVE_NT256V 1
#endif
constexpr int kCoResidentMaxTiles = 128 * 256;          // introselect_block_g: 128 elements per thread (and <= kSelectCap)
constexpr int kSharedMinPairs = 32;                    // VS_BATCH_SHARED: launches of at least this many pairs take the small-footprint build
constexpr size_t kCoResidentDynMax = 32 * 1024;        // dynamic LDS of the small-footprint build; larger levels select on global scratch
#ifndef VS_SMALL_WG_TILES
#define VS_SMALL_WG_TILES 26000
#endif
constexpr int kSmallWgTiles = VS_SMALL_WG_TILES;
constexpr int kPipelineMaxPairs = 128;   // half the CUs   // largest level handled by the 512-thread kernels (<= kSelectCap <= 64 * 512)

}  // namespace

VS_BOUNDS_TU(vs_bounds_fetch_engine)

// =================================================================================================
// VideoAligner
// =================================================================================================
// host-resident video is uploaded in chunks of about this many bytes (192 MB = ~3.4 ms on PCIe 5 x16; measured on MI355X:
// 24 / 48 / 96 / 192 MB chunks reach 0.70 / 0.80 / 0.90 / 0.91 of the pinned-copy rate on a 1.5 GB batch -- every chunk
// costs a thread hand-over and a pipeline drain);
// VS_INGEST_CHUNK_BYTES overrides it (the tests use it to run many small chunks through the pipeline)
static size_t ingest_chunk_bytes() {
    const char* e = getenv("VS_INGEST_CHUNK_BYTES");
    const long long v = e ? atoll(e) : 0;
    return v > 0 ? (size_t)v : (size_t)192 << 20;
}
#define kIngestBytes ingest_chunk_bytes()

// Runs `fn(args...)` on a worker thread.  std::async may throw (std::system_error) when no thread can be started; no exception
// may cross the C ABI, so in that case the task runs on the calling thread (std::launch::deferred) -- no overlap, same result.
template <typename F, typename... A>
static std::future<hipError_t> run_async(F&& fn, A&&... args) {
    try {
        return std::async(std::launch::async, fn, args...);
    } catch (...) {
        return std::async(std::launch::deferred, fn, args...);
    }
}

struct vs_aligner {
    int device = 0;
    hipStream_t stream = nullptr;
    vs_aligner_params params;
    int select_mode = VS_SELECT_DEVICE;   // same survivors in the same order as the host path (tests/test_select_gpu.py)
    int batch_mode = VS_BATCH_EXCLUSIVE;  // VS_BATCH_SHARED: full batches through the small-footprint solver kernel
    int cu_count = kChipCUs;              // hipDeviceProp_t::multiProcessorCount of `device` (vs_aligner_create)
    bool coop_ok = true;                  // helper workgroups: only on the architecture their sc1 hand-offs were validated on (gfx950)

    // sequence state (alignment.hpp:61-70)
    int W = -1, H = -1, fmt = -1;
    int levels = 0;
    LevelDims L[kMaxLevels];
    long long seq = 0;          // frames consumed since (re)initialisation
    int clip_len = 0;           // > 0 during vs_aligner_align_clips: the batch is independent clips of this many frames
    size_t pyr_frame = 0, lm_frame = 0, jac_frame = 0;   // bytes / u16 elements / f32 elements per slot
    int nt_max = 0;

    // device storage
    int cap = 0;                // frames per internal chunk (slots = cap + 1)
    uint8_t* pyr = nullptr;
    uint16_t* lm = nullptr;
    float* jac = nullptr;
    PairState* states = nullptr;
    PairDesc* descs = nullptr;
    uint16_t* wd = nullptr;
    float* wv = nullptr;          // per pair: the 2*nt_max sparse_warpdiff samples of the current level (-> PointRecs::r0); fused kernel: + the template pixels
    int32_t* idx = nullptr;
    uint8_t* recs = nullptr;      // per pair: float4 j[2*nt_max] | u32 xy[2*nt_max] | f32 tv[2*nt_max]
    uint8_t* coop = nullptr;      // helper-workgroup control blocks + exchange buffers (kCoopMaxPairs pairs), see CoopCtrl
    uint8_t* selbuf = nullptr; size_t selbuf_bytes = 0;   // small-footprint solver: per pair u32 keys[nt_max] | u16 ranks[nt_max] (levels that do not fit its LDS)
    int coop_epoch = 0;
    void* stage = nullptr; size_t stage_bytes = 0;   // host-frame upload area (single chunk)
    // host-resident video: two upload areas filled alternately by an uploader thread on its own stream, so the
    // upload of chunk c+1 runs under the pipeline of chunk c (alignment.cpp:210-218: every frame arrives as a host cv::Mat)
    void* ingest[2] = {nullptr, nullptr}; size_t ingest_bytes = 0;
    hipStream_t copy_stream = nullptr;
    // phase-correlation mode (allocated on first use): level-2 half spectra per slot, per-pair scratch and results
    vsp::Context phase;
    int phase_cap = 0;
    float2* pspec = nullptr;
    float2* pG = nullptr;
    float* psurf = nullptr;
    vsp::Pair* ppairs = nullptr;
    uint8_t* pneg = nullptr;
    vsp::Result* pres = nullptr;
    std::vector<vsp::Result> h_pres;
    // pinned host mirrors
    uint16_t* h_wd = nullptr;
    int32_t* h_idx = nullptr;
    PairState* h_states = nullptr;
    PairDesc* h_descs = nullptr;   // pinned: the descriptor upload needs no host synchronisation before the launch

    std::vector<vs_align_info> info;   // last call
    int last_n = 0;

    // opt-in stage timing (alignment.cpp:10-147's PerformanceMetrics)
    bool timing = false;
    vs_stage_timings tm{};
    struct Span { int stage; hipEvent_t a, b; };
    std::vector<Span> spans;               // open spans of the current chunk
    std::vector<hipEvent_t> event_pool;
    hipEvent_t get_event() {
        if (!event_pool.empty()) { hipEvent_t e = event_pool.back(); event_pool.pop_back(); return e; }
        hipEvent_t e; (void)hipEventCreate(&e); return e;
    }
    void t_begin(int stage) {
        if (!timing) return;
        Span sp{stage, get_event(), get_event()};
        (void)hipEventRecord(sp.a, stream);
        spans.push_back(sp);
    }
    void t_end(int launches) {
        if (!timing) return;
        (void)hipEventRecord(spans.back().b, stream);
        tm.launches[spans.back().stage] += launches;
    }
    void t_collect() {   // call after a stream sync
        for (Span& sp : spans) {
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, sp.a, sp.b) == hipSuccess) tm.ms[sp.stage] += ms;
            event_pool.push_back(sp.a); event_pool.push_back(sp.b);
        }
        spans.clear();
    }

    ~vs_aligner() {
        release();
        for (hipEvent_t e : event_pool) (void)hipEventDestroy(e);
        if (copy_stream) (void)hipStreamDestroy(copy_stream);
        if (stream) { (void)vsi::retire_stream(stream); (void)hipStreamDestroy(stream); }   // (the stabilizer warps on this stream)
    }
    void release();
    int configure(int w, int h, int format, const vs_aligner_params& p);
    int ensure_capacity(int n);
    int ensure_phase();
    void release_phase();
    // One chunk = chunk_begin (everything up to and including the solver launch and the copy of its results towards the host: no
    // host synchronisation unless phase correlation is on) + chunk_end (wait, the rare host redo, conversion of the results).
    // run_chunk is the two back to back; the stabilizer puts a chunk's host work between the next chunk's begin and end.
    struct Chunk {
        bool open = false;
        int n = 0, n_pairs = 0, epoch = 0;
        long long seq0 = 0;
        vs_aligner_params p;
        vs_transform* out = nullptr; int32_t* status = nullptr; vs_align_info* infos = nullptr;
        std::vector<int> pair_frame;
        bool direct = false, use_host = false;
        GnParams gp{};
    } ck;
    int chunk_begin(const void* frames, size_t frame_stride, int n, int stride, int mem, const vs_aligner_params& p,
                    vs_transform* out, int32_t* status, vs_align_info* infos);
    int chunk_end();
    int run_chunk(const void* frames, size_t frame_stride, int n, int stride, int mem, const vs_aligner_params& p,
                  vs_transform* out, int32_t* status, vs_align_info* infos) {
        VS_TRY(chunk_begin(frames, frame_stride, n, stride, mem, p, out, status, infos));
        return chunk_end();
    }
    int select_host(int n_pairs, const LevelDims& l);
    // vs_aligner_align_batch / _clips in two halves (single-chunk device-resident batches run asynchronously in between; anything
    // else completes inside align_start)
    bool started = false, started_clips = false;
    int started_n = 0, started_result = 0;
    int32_t* started_status = nullptr;
};

void vs_aligner::release_phase() {
    void* d[] = {pspec, pG, psurf, ppairs, pneg, pres};
    for (void* p : d) if (p) (void)hipFree(p);
    pspec = nullptr; pG = nullptr; psurf = nullptr; ppairs = nullptr; pneg = nullptr; pres = nullptr;
    phase_cap = 0;
    phase.destroy();
}

// buffers of the phase-correlation mode for the current chunk capacity (PhaseLevel = 2, alignment.hpp:69)
int vs_aligner::ensure_phase() {
    const LevelDims& pl = L[2];
    if (phase.w != pl.w || phase.h != pl.h) {
        release_phase();
        const hipError_t pe = phase.configure(pl.w, pl.h, stream);
        if (pe == hipErrorInvalidValue)
            return set_error(VS_ERR_UNSUPPORTED, "phase_correlate: level 2 is %dx%d, padded extent over %d", pl.w, pl.h, vsp::kMaxLine);
        VS_HIP(pe);                                     // (a refused allocation or a failed upload is a HIP error, not a size problem)
    }
    if (phase_cap >= cap) return VS_OK;
    void* d[] = {pspec, pG, psurf, ppairs, pneg, pres};
    for (void* p : d) if (p) (void)hipFree(p);
    pspec = nullptr; pG = nullptr; psurf = nullptr; ppairs = nullptr; pneg = nullptr; pres = nullptr; phase_cap = 0;
    VS_HIP(vsi::dev_alloc((void**)&pspec, ((size_t)cap + 1) * phase.spec_frame() * sizeof(float2)));
    VS_HIP(vsi::dev_alloc((void**)&pG, (size_t)cap * phase.spec_frame() * sizeof(float2)));
    VS_HIP(vsi::dev_alloc((void**)&psurf, (size_t)cap * phase.surface_elems() * sizeof(float)));
    VS_HIP(vsi::dev_alloc((void**)&ppairs, (size_t)cap * sizeof(vsp::Pair)));
    VS_HIP(vsi::dev_alloc((void**)&pneg, (size_t)cap));
    VS_HIP(vsi::dev_alloc((void**)&pres, (size_t)cap * sizeof(vsp::Result)));
    phase_cap = cap;
    return VS_OK;
}

void vs_aligner::release() {
    release_phase();
    void* d[] = {pyr, lm, jac, states, descs, wd, wv, idx, recs, coop, stage, ingest[0], ingest[1], selbuf};
    for (void* p : d) if (p) (void)hipFree(p);
    selbuf = nullptr; selbuf_bytes = 0;
    ingest[0] = ingest[1] = nullptr; ingest_bytes = 0;
    void* hp[] = {h_wd, h_idx, h_states, h_descs};
    for (void* p : hp) if (p) (void)hipHostFree(p);
    pyr = nullptr; lm = nullptr; jac = nullptr; states = nullptr; descs = nullptr; wd = nullptr; wv = nullptr; idx = nullptr;
    recs = nullptr; coop = nullptr; stage = nullptr; stage_bytes = 0; h_wd = nullptr; h_idx = nullptr; h_states = nullptr; h_descs = nullptr;
    cap = 0;
}

// alignment.cpp:155-204: (re)initialisation on first use or size change
int vs_aligner::configure(int w, int h, int format, const vs_aligner_params& p) {
    int lv = 0, ww = w, hh = h;
    do { lv++; ww /= 2; hh /= 2; } while (ww >= p.pyramid_min_width && hh >= p.pyramid_min_height);
    // PhaseLevel = 2 is indexed unconditionally in the reference (alignment.cpp:227): < 3 levels is UB there
    if (lv < 3 || lv > kMaxLevels)
        return set_error(VS_ERR_UNSUPPORTED, "%dx%d with pyramid_min %dx%d gives %d pyramid levels; 3..%d supported", w, h,
                         p.pyramid_min_width, p.pyramid_min_height, lv, kMaxLevels);
    if (w > 65535 || h > 65535) return set_error(VS_ERR_UNSUPPORTED, "frame larger than 65535 (u16 keypoints)");
    // every level is validated BEFORE anything of the old configuration is released or overwritten: a refused size
    // leaves the handle exactly as it was (the reference marks a failed setup with LastWidth = -1, alignment.cpp:360)
    ww = w; hh = h;
    for (int i = 0; i < lv; i++) {
        if (i > 0) { ww /= 2; hh /= 2; }
        const int ts = vs_tile_size(ww, hh);
        if (ww < 4 || hh < 4 || (ww / ts) * (hh / ts) < 1)
            return set_error(VS_ERR_UNSUPPORTED, "pyramid level %d is %dx%d: levels below 4x4 are not supported", i, ww, hh);
    }
    release();
    W = w; H = h; fmt = format; levels = lv; seq = 0;
    size_t img = 0, lmo = 0, jo = 0;
    nt_max = 0;
    ww = w; hh = h;
    for (int i = 0; i < lv; i++) {
        if (i > 0) { ww /= 2; hh /= 2; }
        LevelDims& l = L[i];
        l.w = ww; l.h = hh;
        l.ts = vs_tile_size(ww, hh);
        l.tx = ww / l.ts; l.ty = hh / l.ts; l.nt = l.tx * l.ty;
        l.nsel = (int)static_cast<size_t>((size_t)l.nt * p.smallest_fraction);   // alignment.cpp:464-465
        l.img_off = img; img += ((size_t)ww * hh + 255) & ~(size_t)255;
        l.lm_off = lmo; lmo += (size_t)l.nt * 4;     // x-set (2*nt) + y-set (2*nt)
        l.jac_off = jo; jo += (size_t)l.nt * 8;      // x-set (4*nt) + y-set (4*nt)
        nt_max = std::max(nt_max, l.nt);
    }
    pyr_frame = img; lm_frame = (lmo + 63) & ~(size_t)63; jac_frame = (jo + 63) & ~(size_t)63;
    return VS_OK;
}

// Grows the per-chunk storage to n frames: allocate the whole new set -> copy the carry-over frame -> swap -> free the old set.
// Any failure on the way frees what was allocated here and leaves the handle exactly as it was (same buffers, same `cap`, the
// carry-over frame where it sat), so a failed call is simply a failed call: the next one -- after the caller has made room, or with
// a smaller batch -- finds a consistent handle (tests/test_alloc_failure_gpu.py walks a failure over every allocation).
int vs_aligner::ensure_capacity(int n) {
    if (n <= cap) return VS_OK;
    const int newcap = n;
    const size_t slots = (size_t)newcap + 1;
    const size_t coop_bytes = std::min(newcap, kCoopMaxPairs) * coop_pair_bytes(nt_max);
    // the new set; `owned` frees whatever is still in it when this function returns early
    struct NewSet {
        void* dev[10] = {};
        void* pinned[4] = {};
        ~NewSet() {
            for (void* q : dev) if (q) (void)hipFree(q);
            for (void* q : pinned) if (q) (void)hipHostFree(q);
        }
    } ns;
    enum { D_PYR, D_LM, D_JAC, D_STATES, D_DESCS, D_WD, D_IDX, D_WV, D_RECS, D_COOP };
    const size_t dev_bytes[10] = {
        slots * pyr_frame, slots * lm_frame * 2, slots * jac_frame * 4, sizeof(PairState) * newcap, sizeof(PairDesc) * newcap,
        (size_t)newcap * 2 * nt_max * sizeof(uint16_t), (size_t)newcap * 2 * nt_max * sizeof(int32_t),
        (size_t)newcap * 2 * nt_max * sizeof(float) * 2,      // samples of all pairs, then template pixels
        (size_t)newcap * 2 * nt_max * 28, coop_bytes};
    const size_t pinned_bytes[4] = {(size_t)newcap * 2 * nt_max * sizeof(uint16_t), (size_t)newcap * 2 * nt_max * sizeof(int32_t),
                                    sizeof(PairState) * newcap, sizeof(PairDesc) * newcap};
    for (int i = 0; i < 10; i++) VS_HIP(vsi::dev_alloc(&ns.dev[i], dev_bytes[i]));
    for (int i = 0; i < 4; i++) VS_HIP(vsi::pinned_alloc(&ns.pinned[i], pinned_bytes[i]));
    VS_HIP(hipMemsetAsync(ns.dev[D_COOP], 0, coop_bytes, stream));
    const bool carry = pyr && seq > 0;
    if (carry) {
        // the previous chunk's last frame (slot last_n of the old slabs) becomes slot 0 of the new ones
        VS_HIP(hipMemcpyAsync(ns.dev[D_PYR], pyr + (size_t)last_n * pyr_frame, pyr_frame, hipMemcpyDeviceToDevice, stream));
        VS_HIP(hipMemcpyAsync(ns.dev[D_LM], lm + (size_t)last_n * lm_frame, lm_frame * 2, hipMemcpyDeviceToDevice, stream));
        VS_HIP(hipMemcpyAsync(ns.dev[D_JAC], jac + (size_t)last_n * jac_frame, jac_frame * 4, hipMemcpyDeviceToDevice, stream));
    }
    VS_HIP(hipStreamSynchronize(stream));        // nothing reads the old set any more; nothing below can fail
    // ---- swap: the handle takes the new set, `ns` takes the old one and frees it on the way out ----
    void** dev_members[10] = {(void**)&pyr, (void**)&lm, (void**)&jac, (void**)&states, (void**)&descs, (void**)&wd, (void**)&idx,
                              (void**)&wv, (void**)&recs, (void**)&coop};
    void** pinned_members[4] = {(void**)&h_wd, (void**)&h_idx, (void**)&h_states, (void**)&h_descs};
    for (int i = 0; i < 10; i++) std::swap(*dev_members[i], ns.dev[i]);
    for (int i = 0; i < 4; i++) std::swap(*pinned_members[i], ns.pinned[i]);
    if (carry) last_n = 0;                       // the carry-over now lives in slot 0
    coop_epoch = 0;
    cap = newcap;
    return VS_OK;
}

// alignment.cpp:435-492: flatten row-major, std::nth_element on abs_delta, keep the first n.
// This IS the reference's selection (same STL call on the same element type and order), run on the
// host for every (pair, set) of the batch, spread over a few threads.
int vs_aligner::select_host(int n_pairs, const LevelDims& l) {
    struct DeltaPixel { uint16_t abs_delta, tile_x, tile_y; };   // alignment.hpp:84-87
    const int nt = l.nt, nsel = l.nsel, tx = l.tx;
    const size_t wd_pair = (size_t)2 * nt_max;
    auto work = [&](int begin, int end) {
        std::vector<DeltaPixel> v;
        for (int job = begin; job < end; job++) {
            const int p = job >> 1, set = job & 1;
            if (h_states[p].status != 1) continue;
            const uint16_t* src = h_wd + (size_t)p * wd_pair + (size_t)set * nt;
            if (select_mode == VS_SELECT_STABLE) {
                // (levels beyond the on-device capacity in VS_SELECT_STABLE mode: the same rule on the host --
                // smallest by (abs_delta, tile index), survivors in tile order)
                std::vector<uint64_t> key((size_t)nt);
                for (int i = 0; i < nt; i++) key[i] = ((uint64_t)src[i] << 32) | (uint32_t)i;
                std::vector<uint64_t> sorted(key);
                int32_t* dst = h_idx + (size_t)p * wd_pair + (size_t)set * nt;
                if (nsel > 0) {
                    std::nth_element(sorted.begin(), sorted.begin() + (nsel - 1), sorted.end());     // (distinct keys: the cut is unique)
                    const uint64_t cut = sorted[nsel - 1];
                    int m = 0;
                    for (int i = 0; i < nt; i++) if (key[i] <= cut) dst[m++] = i;
                }
                continue;
            }
            v.clear();
            v.reserve(nt);
            for (int j = 0; j < l.ty; j++)
                for (int k = 0; k < tx; k++)
                    v.push_back(DeltaPixel{src[(size_t)j * tx + k], (uint16_t)k, (uint16_t)j});
            std::nth_element(v.begin(), v.begin() + nsel, v.end(),
                             [](const DeltaPixel& a, const DeltaPixel& b) { return a.abs_delta < b.abs_delta; });
            int32_t* dst = h_idx + (size_t)p * wd_pair + (size_t)set * nt;
            for (int j = 0; j < nsel; j++) dst[j] = (int32_t)v[j].tile_y * tx + v[j].tile_x;
        }
    };
    const int jobs = n_pairs * 2;
    int nthreads = (int)std::min<unsigned>(std::max(1u, std::thread::hardware_concurrency()), 16u);
    nthreads = std::min(nthreads, std::max(1, jobs / 8));
    if (nthreads <= 1) {
        work(0, jobs);
    } else {
        std::vector<std::thread> th;
        const int per = (jobs + nthreads - 1) / nthreads;
        for (int t = 0; t < nthreads; t++) {
            int b = t * per, e = std::min(jobs, b + per);
            if (b < e) th.emplace_back(work, b, e);
        }
        for (auto& t : th) t.join();
    }
    return VS_OK;
}

// One chunk (n <= cap frames): the result of n successive AlignNextFrame calls.
int vs_aligner::chunk_begin(const void* frames, size_t frame_stride, int n, int stride, int mem,
                            const vs_aligner_params& p, vs_transform* out, int32_t* status, vs_align_info* infos) {
    hipStream_t s = stream;
    if (ck.open) return set_error(VS_ERR_ARG, "a chunk is already in flight on this handle");
    ck.n = n; ck.p = p; ck.out = out; ck.status = status; ck.infos = infos; ck.seq0 = seq;
    ck.n_pairs = 0; ck.direct = false; ck.use_host = false; ck.pair_frame.clear();
    const int ch = fmt == VS_FMT_GRAY8 ? 1 : 3;
    const int fbits = vs_format_bits(fmt);
    const size_t esz = fbits > 8 ? 2 : 1;

    // carry-over: the previous call's last frame (pyramid + keyframe tables) moves to slot 0
    if (seq > 0 && last_n != 0) {
        VS_HIP(hipMemcpyAsync(pyr, pyr + (size_t)last_n * pyr_frame, pyr_frame, hipMemcpyDeviceToDevice, s));
        VS_HIP(hipMemcpyAsync(lm, lm + (size_t)last_n * lm_frame, lm_frame * 2, hipMemcpyDeviceToDevice, s));
        VS_HIP(hipMemcpyAsync(jac, jac + (size_t)last_n * jac_frame, jac_frame * 4, hipMemcpyDeviceToDevice, s));
    }

    // ---- ComputePyramid (alignment.cpp:149-235) for all n frames ------------------------------
    const void* dframes = frames;
    t_begin(VS_STAGE_INGEST);
    if (mem == VS_MEM_HOST) {
        const size_t bytes = ((size_t)(n - 1) * frame_stride + (size_t)(H - 1) * stride + (size_t)W * ch) * esz;
        if (bytes > stage_bytes) {
            if (stage) (void)hipFree(stage);
            stage = nullptr; stage_bytes = 0;
            VS_HIP(vsi::dev_alloc(&stage, bytes));
            stage_bytes = bytes;
        }
        VS_HIP(hipMemcpyAsync(stage, frames, bytes, hipMemcpyHostToDevice, s));
        dframes = stage;
    }
    uint8_t* slot1 = pyr + pyr_frame;   // level 0 of slot 1
    if (fmt == VS_FMT_GRAY8) {
        for (int i = 0; i < n; i++)
            VS_HIP(hipMemcpy2DAsync(slot1 + (size_t)i * pyr_frame, W, (const uint8_t*)dframes + (size_t)i * frame_stride,
                                    stride, W, H, hipMemcpyDeviceToDevice, s));
    } else {
        // BGR -> gray level 0 and level 1 in one pass (levels >= 3 always, so level 1 exists)
        VS_HIP(vsk::ingest_pyr(dframes, W, H, stride, fbits > 8 ? 16 : 8, fbits - 8, slot1,
                               slot1 + L[1].img_off, n, frame_stride, pyr_frame, s));
#ifdef VS_EXP_REPEAT
        // analysis builds: a stage launched twice (idempotent) -- the step's growth is what the stage costs beside the warp
        if (VS_EXP_REPEAT & 1) VS_HIP(vsk::ingest_pyr(dframes, W, H, stride, fbits > 8 ? 16 : 8, fbits - 8, slot1, slot1 + L[1].img_off, n, frame_stride, pyr_frame, s));
#endif
    }
    t_end(1);
    t_begin(VS_STAGE_PYR_DOWN);
    for (int l = (fmt == VS_FMT_GRAY8 ? 1 : 2); l < levels; l++)
        VS_HIP(vsk::pyr_down(slot1 + L[l - 1].img_off, L[l - 1].w, L[l - 1].h, L[l - 1].w, slot1 + L[l].img_off, L[l].w,
                             L[l].h, L[l].w, n, pyr_frame, pyr_frame, s));
    t_end(levels - (fmt == VS_FMT_GRAY8 ? 1 : 2));

    // ---- PhaseImage (alignment.cpp:225-229): half spectra of level 2, for the carry-over slot too ------------
    if (p.phase_correlate) {
        VS_TRY(ensure_phase());
        t_begin(VS_STAGE_PHASE);
        const int first = (seq > 0 && clip_len == 0) ? 0 : 1;
        VS_HIP(phase.spectra(pyr + (size_t)first * pyr_frame + L[2].img_off, pyr_frame, L[2].w, n + 1 - first,
                             pspec + (size_t)first * phase.spec_frame(), s));
        t_end(2);
    }

    // ---- ComputeKeyFrame (alignment.cpp:237-276) for the odd frames of the sequence -----------
    // frame i of this chunk sits in slot i + 1; its index in its sequence is g(i) = seq + i, or i mod clip_len
    // when the batch is a set of independent clips (every clip starts its own sequence at 0)
    auto gidx = [&](int i) -> long long { return clip_len > 0 ? (long long)(i % clip_len) : seq + i; };
    {
        // runs of frames (first, count) whose odd members form a stride-2 progression
        std::vector<std::pair<int, int>> runs;
        if (clip_len > 0 && (clip_len & 1)) for (int c0 = 0; c0 < n; c0 += clip_len) runs.emplace_back(c0, std::min(clip_len, n - c0));
        else runs.emplace_back(0, n);
        bool any = false;
        int kf_launches = 0;
        for (auto& r : runs) {
            const int first_odd = r.first + ((gidx(r.first) & 1) ? 0 : 1);
            const int n_odd = first_odd < r.first + r.second ? (r.first + r.second - first_odd + 1) / 2 : 0;
            if (n_odd <= 0) continue;
            if (!any) { t_begin(VS_STAGE_KEYFRAME); any = true; }
            const size_t so = (size_t)(first_odd + 1);
            vsk::KeyframeLevels KL{};
            KL.n = levels;
            for (int l = 0; l < levels; l++)
                KL.lv[l] = vsk::KeyframeLevel{L[l].w, L[l].h, L[l].ts, L[l].tx, L[l].ty, 0, 0, L[l].img_off, L[l].lm_off, L[l].jac_off};
            if (vsk::keyframe_levels_supported(KL)) {
                // one launch for every level of every keyframe of the run
                VS_HIP(vsk::keyframe_levels(pyr + so * pyr_frame, lm + so * lm_frame, jac + so * jac_frame, KL, n_odd, 2 * pyr_frame,
                                            2 * lm_frame, 2 * jac_frame, s));
#ifdef VS_EXP_REPEAT
                if (VS_EXP_REPEAT & 4) VS_HIP(vsk::keyframe_levels(pyr + so * pyr_frame, lm + so * lm_frame, jac + so * jac_frame, KL, n_odd, 2 * pyr_frame, 2 * lm_frame, 2 * jac_frame, s));
#endif
                kf_launches += 1;
            } else {
                for (int l = 0; l < levels; l++) {
                    uint16_t* lmx = lm + so * lm_frame + L[l].lm_off;
                    float* jx = jac + so * jac_frame + L[l].jac_off;
                    VS_HIP(vsk::keyframe(pyr + so * pyr_frame + L[l].img_off, L[l].w, L[l].h, L[l].w, L[l].ts, lmx,
                                         lmx + 2 * (size_t)L[l].nt, jx, jx + 4 * (size_t)L[l].nt, n_odd, 2 * pyr_frame,
                                         2 * lm_frame, 2 * jac_frame, s, true));
                }
                kf_launches += levels;
            }
        }
        if (any) t_end(kf_launches);
    }

    // ---- frame pairs --------------------------------------------------------------------------
    // a pair exists for chunk frame i when g(i) >= 1: frames (g-1, g) = slots (i, i+1)
    std::vector<int>& pair_frame = ck.pair_frame;
    pair_frame.reserve(n);
    for (int i = 0; i < n; i++) {
        memset(&infos[i], 0, sizeof(vs_align_info));
        infos[i].levels = levels;
        out[i] = vs_transform{0, 0, 0, 0};
        status[i] = 0;
        if (gidx(i) >= 1) pair_frame.push_back(i);
        else infos[i].fail_reason = 1;            // alignment.cpp:231-234: first frame of a sequence
    }
    const int n_pairs = (int)pair_frame.size();
    ck.n_pairs = n_pairs;
    if (n_pairs > 0) {
        PairDesc* hd = h_descs;
        for (int q = 0; q < n_pairs; q++) {
            const int i = pair_frame[q];
            const int cur = i + 1, prev = i;
            if (gidx(i) & 1) { hd[q].key_slot = cur; hd[q].tmpl_slot = prev; }
            else { hd[q].key_slot = prev; hd[q].tmpl_slot = cur; }
        }
        for (int q = 0; q < n_pairs; q++) {
            memset(&h_states[q], 0, sizeof(PairState));
            h_states[q].status = 1;
        }
        // Latency mode (a few pairs, device-side selection, identity start): no copies around the launch at all -- the
        // descriptors travel in the kernel arguments, the kernel starts from the identity itself and writes its result straight
        // into the pinned host block the host reads after the synchronisation.
        const bool direct = n_pairs <= kDirectMaxPairs && !p.phase_correlate && select_mode != VS_SELECT_STL_HOST && nt_max <= kSelectCap;
        ck.direct = direct;
        if (!direct) {
            VS_HIP(hipMemcpyAsync(descs, hd, sizeof(PairDesc) * n_pairs, hipMemcpyHostToDevice, s));
            VS_HIP(hipMemcpyAsync(states, h_states, sizeof(PairState) * n_pairs, hipMemcpyHostToDevice, s));
        }
        // alignment.cpp:369-388: cv::phaseCorrelate(PhaseImage[Prev], PhaseImage[Curr]) starts TX,TY of every pair
        std::vector<vsp::Pair> hp;
        std::vector<uint8_t> hneg;
        const float phase_scale = (1 << 2) / float(1 << levels);
        auto apply_phase = [&]() -> int {
            hipLaunchKernelGGL(vs_k_phase_apply, dim3((n_pairs + 255) / 256), dim3(256), 0, s, states, pres, pneg, n_pairs,
                               p.phase_correlate_threshold, phase_scale);
            VS_HIP(hipGetLastError());
            return VS_OK;
        };
        if (p.phase_correlate) {
            hp.resize(n_pairs); hneg.resize(n_pairs);
            for (int q = 0; q < n_pairs; q++) {
                const int i = pair_frame[q];
                hp[q] = vsp::Pair{i, i + 1};
                hneg[q] = (gidx(i) & 1) ? 1 : 0;          // CurrFrameIndex == KeyframeIndex
            }
            VS_HIP(hipMemcpyAsync(ppairs, hp.data(), sizeof(vsp::Pair) * n_pairs, hipMemcpyHostToDevice, s));
            VS_HIP(hipMemcpyAsync(pneg, hneg.data(), (size_t)n_pairs, hipMemcpyHostToDevice, s));
            t_begin(VS_STAGE_PHASE);
            VS_HIP(phase.correlate(pspec, ppairs, n_pairs, pG, psurf, pres, s));
            VS_TRY(apply_phase());
            t_end(4);
            h_pres.resize(n_pairs);
            VS_HIP(hipMemcpyAsync(h_pres.data(), pres, sizeof(vsp::Result) * n_pairs, hipMemcpyDeviceToHost, s));
        }
        // (the descriptors and the start states come from pinned memory that is not touched again before the final
        // synchronisation of this call: no host synchronisation in front of the launch -- it would expose the launch latency)
        if (p.phase_correlate) VS_HIP(hipStreamSynchronize(s));   // hp, hneg go out of use

        const size_t recs_pair = (size_t)2 * nt_max * 28;
        // The pipelined iteration loop shortens one pair's critical path at the price of a speculative sampling pass per level:
        // worth it while the launch does not fill the chip (the results are bit-identical either way).
        static const int pipe_env = []() { const char* e = getenv("VS_GN_PIPELINE"); return e ? atoi(e) : -1; }();
        const int pipeline = pipe_env >= 0 ? pipe_env : (n_pairs <= kPipelineMaxPairs ? 1 : 0);
        static const int stall_env = []() { const char* e = getenv("VS_GN_STALL_HELPERS"); return e ? atoi(e) : 0; }();
        static const int depth_env = []() { const char* e = getenv("VS_GN_SELECT_DEPTH"); return e ? atoi(e) : 0; }();
        GnParams gp{p.threshold, p.max_displacement, p.max_iters, pipeline, stall_env, depth_env, select_mode == VS_SELECT_STABLE ? 1 : 0};
        ck.gp = gp;
        const bool use_host = select_mode == VS_SELECT_STL_HOST || nt_max > kSelectCap;
        ck.use_host = use_host;
        if (!use_host) {
            // VS_SELECT_DEVICE: every level of every pair in one launch (selection = on-device introselect)
            FusedLevels fl;
            fl.levels = levels;
            for (int l = 0; l < levels; l++) {
                fl.w[l] = L[l].w; fl.h[l] = L[l].h; fl.nt[l] = L[l].nt; fl.nsel[l] = L[l].nsel; fl.tx[l] = L[l].tx;
                fl.img_off[l] = L[l].img_off; fl.lm_off[l] = L[l].lm_off; fl.jac_off[l] = L[l].jac_off;
            }
            // selection arrays in LDS: 6 B per tile for one point set; 12 B when both sets fit, so that they are selected
            // side by side (introselect_dual) -- always for 1080p, for every level but the finest at 4K
            size_t dyn = (((size_t)nt_max * ((size_t)nt_max * 12 <= 150 * 1024 ? 12 : 6) + 15) & ~(size_t)15);
            // ... and room for the coarsest level's image if it fits next to the static LDS (the kernel stages it for the
            // Gauss-Newton iterations): 480x270 at 1080p / 4K with pyramid_min_width 256
            {
                const size_t img_bytes = (size_t)L[levels - 1].w * L[levels - 1].h + 8;     // the coarsest level
                if (img_bytes <= 150 * 1024) dyn = std::max(dyn, (img_bytes + 15) & ~(size_t)15);
                // ... and its selection arrays behind the image, so that the level's sparse_warpdiff samples from LDS too
                const size_t both = ((img_bytes + 15) & ~(size_t)15) + (size_t)L[levels - 1].nt * 12;
                if (both <= 150 * 1024) dyn = std::max(dyn, (both + 15) & ~(size_t)15);
            }
            dyn = std::max<size_t>(dyn, 512);       // (tiny levels: the wave-level selection rounds use 128 bytes behind the keys as scratch)
            const bool small_wg = nt_max <= kSmallWgTiles;
            // latency mode: helper workgroups for the large levels of a pair (see CoopCtrl)
            static const int coop_env = []() { const char* e = getenv("VS_GN_HELPERS"); return e ? atoi(e) : -1; }();
            int group = 1;
            if (coop_ok && n_pairs <= kCoopMaxPairs && L[0].nt >= kCoopMinTiles)
                group = coop_env >= 0 ? std::max(1, std::min(coop_env, kCoopMaxGroup)) : std::max(1, std::min(kCoopGroup, cu_count / n_pairs));
            // full batches of a handle in VS_BATCH_SHARED mode: the small-footprint build (shares CUs with whatever else is
            // running, e.g. the previous clip's warp launch); VS_GN_CORESIDENT=0 / 1 overrides the rule (1: every launch, helpers off)
            bool cores = false;
#ifdef VS_HAVE_NT256V
            static const int cores_env = []() { const char* e = getenv("VS_GN_CORESIDENT"); return e ? atoi(e) : -1; }();
            cores = small_wg && nt_max <= kCoResidentMaxTiles &&
                    (cores_env >= 0 ? cores_env != 0 : (batch_mode == VS_BATCH_SHARED && n_pairs >= kSharedMinPairs));
            const size_t selbuf_pair = (((size_t)nt_max * 6 + 255) & ~(size_t)255);
            if (cores) {
                group = 1;
                dyn = 16;
                bool need_global = false;
                for (int l = 0; l < levels; l++) {
                    const size_t nt = (size_t)L[l].nt;
                    const size_t need = ((nt <= 32 * (nt256v::kGnThreads / 2) && nt * 12 <= kCoResidentDynMax ? nt * 12 : nt * 6) + 15) & ~(size_t)15;
                    if (need <= kCoResidentDynMax) dyn = std::max(dyn, need); else need_global = true;
                }
                dyn = std::max<size_t>(dyn, 512);
                // (the kernel takes a non-null scratch pointer as "LDS block sized per level": always passed in this mode)
                const size_t want = need_global ? selbuf_pair * (size_t)std::max(cap, n_pairs) : 256;
                if (selbuf_bytes < want) {
                    if (selbuf) (void)hipFree(selbuf);
                    selbuf = nullptr; selbuf_bytes = 0;
                    VS_HIP(vsi::dev_alloc((void**)&selbuf, want));
                    selbuf_bytes = want;
                }
            }
            const auto kernel = cores ? nt256v::vs_k_align_pairs : (small_wg ? nt512::vs_k_align_pairs : nt1024::vs_k_align_pairs);
#else
            const auto kernel = small_wg ? nt512::vs_k_align_pairs : nt1024::vs_k_align_pairs;
#endif
#ifdef VS_HAVE_NT256V
            const int kthreads = cores ? nt256v::kGnThreads : (small_wg ? nt512::kGnThreads : nt1024::kGnThreads);
#else
            const int kthreads = small_wg ? nt512::kGnThreads : nt1024::kGnThreads;
#endif
            {   // The limit belongs to (kernel, device) and is shared by every handle of the process: it is only ever raised,
                // and only when a launch needs more than was granted before (the call costs microseconds of host time).
                static std::mutex dyn_mu;
                static size_t dyn_set[3][kMaxDevices];
                std::lock_guard<std::mutex> g(dyn_mu);
                size_t& granted = dyn_set[cores ? 2 : (small_wg ? 0 : 1)][std::min(std::max(device, 0), kMaxDevices - 1)];
                if (granted < dyn || device >= kMaxDevices) {
                    VS_HIP(hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)std::max(dyn, granted)));
                    granted = std::max(dyn, granted);
                }
            }
            t_begin(VS_STAGE_GN);
            const int epoch = ++coop_epoch;
            ck.epoch = epoch;
            PairDescPack dpack{};
            if (direct) for (int q = 0; q < n_pairs; q++) dpack.d[q] = hd[q];
            hipLaunchKernelGGL(kernel, dim3(n_pairs * group), dim3(kthreads), dyn, s,
                               direct ? h_states : states, descs, pyr, pyr_frame, lm, lm_frame, jac, jac_frame, recs, recs_pair, nt_max, (int)dyn,
                               fl, gp, group, coop, epoch, wv, dpack, direct ? 1 : 0, cores ? selbuf : nullptr, selbuf_pair);
            VS_HIP(hipGetLastError());
            t_end(1);
            if (!direct) VS_HIP(hipMemcpyAsync(h_states, states, sizeof(PairState) * n_pairs, hipMemcpyDeviceToHost, s));
        }
    }
    ck.open = true;
    return VS_OK;
}

int vs_aligner::chunk_end() {
    if (!ck.open) return set_error(VS_ERR_ARG, "no chunk in flight on this handle");
    ck.open = false;
    hipStream_t s = stream;
    const int n = ck.n, n_pairs = ck.n_pairs, epoch = ck.epoch;
    const vs_aligner_params& p = ck.p;
    vs_transform* out = ck.out; int32_t* status = ck.status; vs_align_info* infos = ck.infos;
    const std::vector<int>& pair_frame = ck.pair_frame;
    const bool direct = ck.direct;
    bool use_host = ck.use_host;
    const GnParams gp = ck.gp;
    const long long seq0 = ck.seq0;
    auto gidx = [&](int i) -> long long { return clip_len > 0 ? (long long)(i % clip_len) : seq0 + i; };
    if (n_pairs > 0) {
        PairDesc* hd = h_descs;
        static const bool poll_done = []() { const char* e = getenv("VS_GN_POLL"); return e ? atoi(e) != 0 : true; }();
        const float phase_scale = (1 << 2) / float(1 << levels);
        auto apply_phase = [&]() -> int {
            hipLaunchKernelGGL(vs_k_phase_apply, dim3((n_pairs + 255) / 256), dim3(256), 0, s, states, pres, pneg, n_pairs,
                               p.phase_correlate_threshold, phase_scale);
            VS_HIP(hipGetLastError());
            return VS_OK;
        };
        const size_t wd_pair = (size_t)2 * nt_max, recs_pair = (size_t)2 * nt_max * 28;
        if (!use_host) {
            if (direct && poll_done && !timing) {
                // the kernel writes a pair's `pad` word last, behind a system-scope fence: the results are in host memory when
                // every pair's word has arrived -- the host spins on them instead of sleeping in the runtime (the stream itself
                // is ordered by the next enqueue); after ~2 ms it falls back to the synchronisation, which also reports faults
                const auto t0 = std::chrono::steady_clock::now();
                bool all = false;
                while (!all) {
                    all = true;
                    for (int q = 0; q < n_pairs; q++)
                        if (reinterpret_cast<volatile int32_t*>(&h_states[q].pad)[0] != (int32_t)epoch) { all = false; break; }
                    if (!all && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(2)) break;
                }
                std::atomic_thread_fence(std::memory_order_acquire);
                if (!all) VS_HIP(hipStreamSynchronize(s));
                else {
                    // the results are in; helper workgroups may still be winding down, and a fault in any of them must be this
                    // call's error, not the next one's: one non-blocking look at the stream
                    const hipError_t qe = hipStreamQuery(s);
                    if (qe != hipSuccess && qe != hipErrorNotReady) VS_HIP(qe);
                }
            } else {
                VS_HIP(hipStreamSynchronize(s));
            }
            for (int q = 0; q < n_pairs; q++)
                if (h_states[q].fail_reason >= 100) use_host = true;   // 100: libstdc++ would have heap-selected; 101: a helper workgroup timed out -- redo on the host
            if (use_host) {
                for (int q = 0; q < n_pairs; q++) { memset(&h_states[q], 0, sizeof(PairState)); h_states[q].status = 1; }
                if (direct) VS_HIP(hipMemcpyAsync(descs, hd, sizeof(PairDesc) * n_pairs, hipMemcpyHostToDevice, s));
                VS_HIP(hipMemcpyAsync(states, h_states, sizeof(PairState) * n_pairs, hipMemcpyHostToDevice, s));
                if (p.phase_correlate) VS_TRY(apply_phase());
                VS_HIP(hipStreamSynchronize(s));
            }
        }
        for (int l = levels - 1; use_host && l >= 0; l--) {
            const LevelDims& ld = L[l];
            t_begin(VS_STAGE_WARPDIFF);
            hipLaunchKernelGGL(vs_k_warpdiff_batch, dim3((ld.nt + 255) / 256, n_pairs, 2), dim3(256), 0, s, states, descs, pyr,
                               pyr_frame, ld.img_off, ld.w, ld.h, lm, lm_frame, ld.lm_off, ld.nt, wd, wv, wd_pair);
            VS_HIP(hipGetLastError());
            t_end(1);
            {
                const auto t0 = std::chrono::steady_clock::now();
                VS_HIP(hipMemcpyAsync(h_wd, wd, (size_t)n_pairs * wd_pair * 2, hipMemcpyDeviceToHost, s));
                VS_HIP(hipMemcpyAsync(h_states, states, sizeof(PairState) * n_pairs, hipMemcpyDeviceToHost, s));
                VS_HIP(hipStreamSynchronize(s));
                VS_TRY(select_host(n_pairs, ld));
                VS_HIP(hipMemcpyAsync(idx, h_idx, (size_t)n_pairs * wd_pair * 4, hipMemcpyHostToDevice, s));
                if (timing) tm.ms[VS_STAGE_SELECT] += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
            }
            if (ld.nsel > 0) {
                t_begin(VS_STAGE_GATHER);
                hipLaunchKernelGGL(vs_k_gather_selected, dim3((ld.nsel + 255) / 256, n_pairs, 2), dim3(256), 0, s, states,
                                   descs, pyr, pyr_frame, ld.img_off, ld.w, ld.h, lm, lm_frame, ld.lm_off, jac, jac_frame,
                                   ld.jac_off, ld.nt, ld.nsel, idx, wd_pair, wv, recs, recs_pair, nt_max);
                VS_HIP(hipGetLastError());
                t_end(1);
            }
            t_begin(VS_STAGE_GN);
            if (nt_max <= kSmallWgTiles)
                hipLaunchKernelGGL(nt512::vs_k_gn_level, dim3(n_pairs), dim3(nt512::kGnThreads), 0, s, states, descs, pyr, pyr_frame,
                                   ld.img_off, ld.w, ld.h, ld.nsel, recs, recs_pair, nt_max, l, gp);
            else
                hipLaunchKernelGGL(nt1024::vs_k_gn_level, dim3(n_pairs), dim3(nt1024::kGnThreads), 0, s, states, descs, pyr, pyr_frame,
                                   ld.img_off, ld.w, ld.h, ld.nsel, recs, recs_pair, nt_max, l, gp);
            VS_HIP(hipGetLastError());
            t_end(1);
        }
        if (use_host) {
            VS_HIP(hipMemcpyAsync(h_states, states, sizeof(PairState) * n_pairs, hipMemcpyDeviceToHost, s));
            VS_HIP(hipStreamSynchronize(s));
        }
#ifdef VS_PROFILE_STAMPS
        for (int l = levels - 1; l >= 0; l--) {
            const unsigned long long* t = h_states[0].stamps[l];
            fprintf(stderr, "[stamps] level %d: warpdiff %llu select %llu gather %llu gn(hessian+%d iters) %llu ticks\n", l, t[1] - t[0],
                    t[2] - t[1], t[3] - t[2], h_states[0].iterations[l], t[4] - t[3]);
        }
#endif
        for (int q = 0; q < n_pairs; q++) {
            const int i = pair_frame[q];
            const PairState& st = h_states[q];
            vs_align_info& inf = infos[i];
            inf.status = st.status; inf.fail_reason = st.fail_reason; inf.fail_level = st.fail_level;
            for (int l = 0; l < levels; l++) {
                inf.iterations[l] = st.iterations[l]; inf.condition[l] = st.condition[l]; tm.gn_iterations += st.iterations[l];
                if (st.iterations[l] > 0) {      // the level was reached (a level always runs at least one iteration)
                    inf.selected_x[l] = L[l].nsel; inf.selected_y[l] = L[l].nsel;      // alignment.cpp:464-465: both sets keep size * fraction
                    inf.level_transform[l] = vs_transform{st.level_T[l][0], st.level_T[l][1], st.level_T[l][2], st.level_T[l][3]};
                }
            }
            if (p.phase_correlate) { inf.phase_dx = h_pres[q].dx; inf.phase_dy = h_pres[q].dy; inf.phase_response = h_pres[q].response; }
            vs_transform t{st.T[0], st.T[1], st.T[2], st.T[3]};
            if (st.status == 1) {
                if ((gidx(i) & 1) == 0) t = vs_transform_inverse(&t);   // alignment.cpp:690-693
                status[i] = 1;
            }
            // on failure the reference returns false with `transform` left at the estimate it had reached
            // (alignment.cpp:661-667,674-677: no level rescale, no inversion) -- and VideoStabilizer feeds that value to
            // the smoother all the same (stabilizer.cpp:18-44), so it is part of the observable behaviour
            out[i] = t;
        }
    } else {
        VS_HIP(hipStreamSynchronize(s));
    }
    if (timing) { t_collect(); tm.frames += n; }
    seq = seq0 + n;
    last_n = n;
    return VS_OK;
}

extern "C" {

// Kernel-level form of the keep-best-fraction step (alignment.cpp:435-492) for n_arrays independent
// warpdiff tables: out_idx[a*tx*ty + 0..count) = tile_y*tx+tile_x of the survivors in std::nth_element's order.
static int select_smallest_impl(const uint16_t* warpdiff, int n_arrays, int tx, int ty, float fraction, int32_t* out_idx,
                                int32_t* status, int mem, void* stream, int rule);
int vs_select_smallest(const uint16_t* warpdiff, int n_arrays, int tx, int ty, float fraction, int32_t* out_idx,
                       int32_t* status, int mem, void* stream) try {
    VS_ARG(status);
    return select_smallest_impl(warpdiff, n_arrays, tx, ty, fraction, out_idx, status, mem, stream, 0);
} VS_CATCH_ALL
// ... under VS_SELECT_STABLE's rule: smallest by (abs_delta, tile index), the survivors in ascending tile order
int vs_select_smallest_stable(const uint16_t* warpdiff, int n_arrays, int tx, int ty, float fraction, int32_t* out_idx, int mem,
                              void* stream) try {
    return select_smallest_impl(warpdiff, n_arrays, tx, ty, fraction, out_idx, nullptr, mem, stream, 1);
} VS_CATCH_ALL
static int select_smallest_impl(const uint16_t* warpdiff, int n_arrays, int tx, int ty, float fraction, int32_t* out_idx,
                                int32_t* status, int mem, void* stream, int rule) {
    VS_ARG(warpdiff && out_idx && (status || rule) && n_arrays >= 1 && tx >= 1 && ty >= 1 && fraction > 0.0f && fraction <= 1.0f);
    const int nt = tx * ty;
    if (nt > kSelectCap) return set_error(VS_ERR_UNSUPPORTED, "%d tiles exceed the on-device selection capacity %d", nt, kSelectCap);
    if (!vsi::device_ready()) return VS_ERR_HIP;
    hipStream_t s = (hipStream_t)stream;
    const int nsel = (int)static_cast<size_t>((size_t)nt * fraction);
    vsi::Staged a, o, st;
    VS_TRY(a.in(warpdiff, (size_t)n_arrays * nt * 2, mem, s));
    VS_TRY(o.out(out_idx, (size_t)n_arrays * nt * 4, mem));
    if (status) VS_TRY(st.out(status, (size_t)n_arrays * 4, mem));
    // (a | posR; the wave-level rounds use the first 128 bytes of posR as scratch whatever the array length)
    const size_t dyn = std::max<size_t>((((size_t)nt * 6 + 15) & ~(size_t)15), (((size_t)nt * 4 + 128 + 15) & ~(size_t)15));
    const bool small_wg = nt <= kSmallWgTiles;
    const auto kernel = small_wg ? nt512::vs_k_select : nt1024::vs_k_select;
    VS_HIP(hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn));
    hipLaunchKernelGGL(kernel, dim3(n_arrays), dim3(small_wg ? nt512::kGnThreads : nt1024::kGnThreads), dyn, s, a.as<uint16_t>(), nt,
                       nsel, o.as<int32_t>(), status ? st.as<int32_t>() : (int32_t*)nullptr, rule);
    VS_HIP(hipGetLastError());
    if (status) VS_TRY(vsi::finish_outputs(mem, s, {&o, &st}));
    else VS_TRY(vsi::finish_outputs(mem, s, {&o}));
    return nsel;
}

vs_aligner* vs_aligner_create(const vs_aligner_params* params, int device) try {
    if (!vsi::device_ready()) return nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || device < 0 || device >= n) {
        set_error(VS_ERR_ARG, "device %d out of range (%d devices)", device, n);
        return nullptr;
    }
    if (hipSetDevice(device) != hipSuccess) { set_error(VS_ERR_HIP, "hipSetDevice(%d) failed", device); return nullptr; }
    vs_aligner* a = new vs_aligner();
    a->device = device;
    {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, device) == hipSuccess) {
            if (prop.multiProcessorCount > 0) a->cu_count = prop.multiProcessorCount;
            a->coop_ok = strncmp(prop.gcnArchName, "gfx950", 6) == 0;
        }
    }
    if (params) a->params = *params; else vs_aligner_params_default(&a->params);
    {   // VS_SELECT_MODE=0|1|2 (read once): the selection mode new handles start in -- for callers that cannot reach
        // vs_aligner_set_select_mode (the facade classes, the harness programs); an explicit set_select_mode still wins
        static const int env_mode = []() {
            const char* e = getenv("VS_SELECT_MODE");
            const int m = e ? atoi(e) : -1;
            if (m == VS_SELECT_STL_HOST || m == VS_SELECT_DEVICE || m == VS_SELECT_STABLE) {
                fprintf(stderr, "libvs_amd: VS_SELECT_MODE=%d -- new aligner / stabilizer handles start in selection mode %d (%s)\n", m, m,
                        m == VS_SELECT_STABLE ? "VS_SELECT_STABLE" : (m == VS_SELECT_DEVICE ? "VS_SELECT_DEVICE" : "VS_SELECT_STL_HOST"));
                return m;
            }
            return -1;
        }();
        if (env_mode >= 0) a->select_mode = env_mode;
    }
    // VS_ALIGNER_STREAM_PRIORITY=low|high (read once; experiments): the handle's stream at the device's least / greatest priority
    static const int prio_env = []() { const char* e = getenv("VS_ALIGNER_STREAM_PRIORITY"); return !e ? 0 : (e[0] == 'l' ? -1 : (e[0] == 'h' ? 1 : 0)); }();
    hipError_t se;
    if (prio_env != 0) {
        int least = 0, greatest = 0;
        (void)hipDeviceGetStreamPriorityRange(&least, &greatest);
        se = hipStreamCreateWithPriority(&a->stream, hipStreamNonBlocking, prio_env < 0 ? least : greatest);
    } else se = hipStreamCreateWithFlags(&a->stream, hipStreamNonBlocking);
    if (se != hipSuccess) {
        set_error(VS_ERR_HIP, "hipStreamCreate failed");
        delete a;
        return nullptr;
    }
    return a;
} VS_CATCH_ALL_NULL

void vs_aligner_destroy(vs_aligner* a) {
    if (!a) return;
    (void)hipSetDevice(a->device);
    delete a;
}

int vs_aligner_set_select_mode(vs_aligner* a, int mode) try {
    VS_ARG(a && (mode == VS_SELECT_STL_HOST || mode == VS_SELECT_DEVICE || mode == VS_SELECT_STABLE));
    a->select_mode = mode;
    return VS_OK;
} VS_CATCH_ALL

int vs_aligner_get_select_mode(const vs_aligner* a) try {
    VS_ARG(a);
    return a->select_mode;
} VS_CATCH_ALL

int vs_aligner_set_batch_mode(vs_aligner* a, int mode) try {
    VS_ARG(a && (mode == VS_BATCH_EXCLUSIVE || mode == VS_BATCH_SHARED));
    a->batch_mode = mode;
    return VS_OK;
} VS_CATCH_ALL

int vs_aligner_reset(vs_aligner* a) try {
    VS_ARG(a);
    a->seq = 0;
    return VS_OK;
} VS_CATCH_ALL

void* vs_aligner_stream(const vs_aligner* a) { return a ? (void*)a->stream : nullptr; }

// Everything enqueued on `producer_stream` so far happens before anything the handle enqueues from now on.
int vs_aligner_wait_stream(vs_aligner* a, void* producer_stream) try {
    VS_ARG(a);
    VS_HIP(hipSetDevice(a->device));
    hipEvent_t ev = nullptr;
    VS_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    hipError_t e = hipEventRecord(ev, (hipStream_t)producer_stream);
    if (e == hipSuccess) e = hipStreamWaitEvent(a->stream, ev, 0);
    (void)hipEventDestroy(ev);        // the wait already holds what it needs; destruction is deferred by the runtime
    VS_HIP(e);
    return VS_OK;
} VS_CATCH_ALL

}  // extern "C"

// First half of vs_aligner_align_batch (clip_frames == 0) / vs_aligner_align_clips (clip_frames > 0).  With `async`, a
// device-resident batch that fits one chunk is only enqueued (vs_aligner::chunk_begin) and align_finish completes it: the caller
// may do unrelated host work in between, but nothing on this handle.  Every other case completes here and align_finish just
// hands out the result.
static int align_start_impl(vs_aligner* a, const void* frames, size_t frame_stride, int n, int w, int h, int stride, int format, int mem,
                            const vs_aligner_params* params, vs_transform* out, int32_t* status, bool async);
static void align_close_clips(vs_aligner* a) {
    if (a->started_clips) { a->started_clips = false; a->clip_len = 0; a->seq = 0; }
}
// An align call that returns an error (a refused allocation, a HIP failure) ends the running sequence: the next frame is the
// first frame of a new one, exactly as a fresh handle would treat it -- the reference's protocol for a failed kernel call
// (alignment.cpp:357-367: LastWidth = -1, so the next AlignNextFrame re-initialises).  Device buffers stay as they are.
static void align_failed(vs_aligner* a) {
    a->seq = 0;
    (void)hipStreamSynchronize(a->stream);                 // nothing of the failed call is still running when its events go back to the pool
    a->ck.open = false;
    for (vs_aligner::Span& sp : a->spans) { a->event_pool.push_back(sp.a); a->event_pool.push_back(sp.b); }
    a->spans.clear();
}
static int align_start(vs_aligner* a, const void* frames, size_t frame_stride, int n, int clip_frames, int w, int h, int stride, int format,
                       int mem, const vs_aligner_params* params, vs_transform* out, int32_t* status, bool async) {
    VS_ARG(a && clip_frames >= 0);
    if (a->started || a->ck.open) return set_error(VS_ERR_ARG, "an alignment is already in flight on this handle");
    a->started_result = 0;
    if (clip_frames > 0) { a->seq = 0; a->clip_len = clip_frames; a->started_clips = true; }
    // (guarded: an exception -- a host allocation that fails -- ends the call like any other stage failure, under the protocol below)
    const int r = vsi::guarded([&] { return align_start_impl(a, frames, frame_stride, n, w, h, stride, format, mem, params, out, status, async); });
    // a call REJECTED for its arguments (VS_ERR_ARG: every such test precedes the first touch of the handle's state) leaves the running
    // sequence alone -- the reference resets only when a kernel stage fails (alignment.cpp:357-367); any later error ends it
    if (r < 0) { a->started = false; align_close_clips(a); if (r != VS_ERR_ARG) align_failed(a); }
    return r;
}
static int align_finish(vs_aligner* a) {
    int r = a->started_result;
    if (a->started) {
        a->started = false;
        r = vsi::guarded([&] { return a->chunk_end(); });
        if (r == VS_OK) for (int i = 0; i < a->started_n; i++) r += a->started_status[i];
    }
    align_close_clips(a);
    if (r < 0) align_failed(a);
    return r;
}
// an alignment that was started and will not be finished (an error elsewhere): drain the stream, forget the chunk
static void align_abandon(vs_aligner* a) {
    if (a->ck.open) { (void)hipStreamSynchronize(a->stream); a->ck.open = false; }
    a->started = false;
    align_close_clips(a);
}

extern "C" {

int vs_aligner_align_batch(vs_aligner* a, const void* frames, size_t frame_stride, int n, int w, int h, int stride,
                           int format, int mem, const vs_aligner_params* params, vs_transform* out, int32_t* status) try {
    const int r = align_start(a, frames, frame_stride, n, 0, w, h, stride, format, mem, params, out, status, false);
    return r < 0 ? r : align_finish(a);
} VS_CATCH_ALL

}  // extern "C"

static int align_start_impl(vs_aligner* a, const void* frames, size_t frame_stride, int n, int w, int h, int stride, int format, int mem,
                            const vs_aligner_params* params, vs_transform* out, int32_t* status, bool async) {
    VS_ARG(a && frames && out && status && n >= 1 && w >= 8 && h >= 8);
    VS_ARG(w <= 65535 && h <= 65535);                      // (tile coordinates are 16-bit; keeps w * ch and the frame spans below inside their types)
    VS_ARG(vs_format_bits(format) != 0);
    const int ch = format == VS_FMT_GRAY8 ? 1 : 3;
    VS_ARG(stride >= w * ch);
    VS_ARG(n == 1 || frame_stride >= (size_t)(h - 1) * stride + (size_t)w * ch);
    const vs_aligner_params& p = params ? *params : a->params;
    VS_ARG(p.max_iters >= 1 && p.smallest_fraction > 0.0f && p.smallest_fraction <= 1.0f);
    VS_HIP(hipSetDevice(a->device));
    if (a->W != w || a->H != h || a->fmt != format) VS_TRY(a->configure(w, h, format, p));   // alignment.cpp:155
    // nsel follows the *current* params like the reference (alignment.cpp:464-465)
    for (int l = 0; l < a->levels; l++)
        a->L[l].nsel = (int)static_cast<size_t>((size_t)a->L[l].nt * p.smallest_fraction);
    a->info.assign(n, vs_align_info{});
    const size_t esz = vs_format_bits(format) > 8 ? 2 : 1;
    // chunking bounds device memory: at most ~6 GiB of pyramids per handle
    int max_chunk = (int)std::max<size_t>(2, std::min<size_t>(1024, ((size_t)6 << 30) / std::max<size_t>(1, a->pyr_frame)));
    if (a->clip_len > 0) {   // whole clips per chunk
        if (a->clip_len > max_chunk) return set_error(VS_ERR_UNSUPPORTED, "clips of %d frames exceed the %d-frame chunk", a->clip_len, max_chunk);
        max_chunk -= max_chunk % a->clip_len;
    }
    int aligned = 0;
    // Host-resident frames, more than one upload's worth: pipelined ingest.  The batch is cut into chunks of ~kIngestBytes;
    // an uploader thread copies chunk c+1 into the other upload area (its own stream) while chunk c runs through the
    // pipeline from device memory.  Same chunks, same arithmetic as the device-resident path: results are identical.
    const size_t frame_bytes = ((size_t)(h - 1) * stride + (size_t)w * ch) * esz;
    const size_t step_bytes = n > 1 ? frame_stride * esz : frame_bytes;
    int host_chunk = (int)std::min<size_t>((size_t)max_chunk, std::max<size_t>(4, kIngestBytes / std::max<size_t>(1, step_bytes)));
    if (a->clip_len > 0) host_chunk = std::max(a->clip_len, host_chunk - host_chunk % a->clip_len);
    if (mem == VS_MEM_HOST && n > host_chunk) {
        const size_t area = (size_t)(host_chunk - 1) * step_bytes + frame_bytes;
        if (a->ingest_bytes < area) {
            for (void*& q : a->ingest) { if (q) (void)hipFree(q); q = nullptr; }
            a->ingest_bytes = 0;
            VS_HIP(vsi::dev_alloc(&a->ingest[0], area));
            VS_HIP(vsi::dev_alloc(&a->ingest[1], area));
            a->ingest_bytes = area;
        }
        if (!a->copy_stream) VS_HIP(hipStreamCreateWithFlags(&a->copy_stream, hipStreamNonBlocking));
        auto upload = [a, frames, step_bytes, frame_bytes, n, host_chunk](int c) -> hipError_t {
            const int off = c * host_chunk, m = std::min(host_chunk, n - off);
            hipError_t e = hipSetDevice(a->device);
            if (e != hipSuccess) return e;
            e = hipMemcpyAsync(a->ingest[c & 1], (const uint8_t*)frames + (size_t)off * step_bytes, (size_t)(m - 1) * step_bytes + frame_bytes,
                               hipMemcpyHostToDevice, a->copy_stream);
            return e != hipSuccess ? e : hipStreamSynchronize(a->copy_stream);
        };
        const int n_chunks = (n + host_chunk - 1) / host_chunk;
        std::future<hipError_t> next = run_async(upload, 0);
        for (int c = 0; c < n_chunks; c++) {
            const int off = c * host_chunk, m = std::min(host_chunk, n - off);
            const hipError_t ue = next.get();                 // chunk c is in ingest[c & 1]
            if (c + 1 < n_chunks) next = run_async(upload, c + 1);   // chunk c-1 (same area) has been consumed
            int r = ue == hipSuccess ? VS_OK : set_error(VS_ERR_HIP, "frame upload failed: %s", hipGetErrorString(ue));
            if (r == VS_OK) r = a->ensure_capacity(m);
            if (r == VS_OK) r = a->run_chunk(a->ingest[c & 1], frame_stride, m, stride, VS_MEM_DEVICE, p, out + off, status + off, a->info.data() + off);
            if (r != VS_OK) { if (next.valid()) (void)next.get(); return r; }
        }
    } else if (async && mem == VS_MEM_DEVICE && n <= max_chunk) {
        VS_TRY(a->ensure_capacity(n));
        VS_TRY(a->chunk_begin(frames, frame_stride, n, stride, mem, p, out, status, a->info.data()));
        a->started = true; a->started_n = n; a->started_status = status;
        return VS_OK;
    } else {
        for (int off = 0; off < n; off += max_chunk) {
            const int m = std::min(max_chunk, n - off);
            VS_TRY(a->ensure_capacity(m));
            const uint8_t* base = (const uint8_t*)frames + (size_t)off * frame_stride * esz;
            VS_TRY(a->run_chunk(base, frame_stride, m, stride, mem, p, out + off, status + off, a->info.data() + off));
        }
    }
    for (int i = 0; i < n; i++) aligned += status[i];
    a->started_result = aligned;
    return VS_OK;
}

extern "C" {

// n_clips independent clips of frames_per_clip frames each, back to back in memory: the results of aligning every
// clip with its own fresh VideoAligner, computed together (every stage is one launch over all clips, so short
// clips still fill the GPU).  The handle's running sequence is reset before and after.
int vs_aligner_align_clips(vs_aligner* a, const void* frames, size_t frame_stride, int n_clips, int frames_per_clip, int w,
                           int h, int stride, int format, int mem, const vs_aligner_params* params, vs_transform* out,
                           int32_t* status) try {
    VS_ARG(a && n_clips >= 1 && frames_per_clip >= 1 && (long long)n_clips * frames_per_clip <= 0x7fffffff);
    const int r = align_start(a, frames, frame_stride, n_clips * frames_per_clip, frames_per_clip, w, h, stride, format, mem, params, out,
                              status, false);
    return r < 0 ? r : align_finish(a);
} VS_CATCH_ALL

int vs_aligner_align_next(vs_aligner* a, const void* frame, int w, int h, int stride, int format, int mem,
                          const vs_aligner_params* params, vs_transform* out) try {
    int32_t st = 0;
    int r = vs_aligner_align_batch(a, frame, 0, 1, w, h, stride, format, mem, params, out, &st);
    if (r < 0) return r;
    return st;
} VS_CATCH_ALL

int vs_aligner_enable_timing(vs_aligner* a, int enable) try {
    VS_ARG(a);
    a->timing = enable != 0;
    memset(&a->tm, 0, sizeof(a->tm));
    return VS_OK;
} VS_CATCH_ALL

int vs_aligner_get_timings(vs_aligner* a, vs_stage_timings* out) try {
    VS_ARG(a && out);
    *out = a->tm;
    return VS_OK;
} VS_CATCH_ALL

int vs_aligner_get_info(const vs_aligner* a, int i, vs_align_info* info) try {
    VS_ARG(a && info && i >= 0 && i < (int)a->info.size());
    *info = a->info[i];
    return VS_OK;
} VS_CATCH_ALL

int vs_aligner_level_dims(const vs_aligner* a, int level, int* w, int* h, int* tx, int* ty, int* ts) try {
    VS_ARG(a && level >= 0 && level < a->levels);
    const LevelDims& l = a->L[level];
    if (w) *w = l.w; if (h) *h = l.h; if (tx) *tx = l.tx; if (ty) *ty = l.ty; if (ts) *ts = l.ts;
    return VS_OK;
} VS_CATCH_ALL

// frame i of the most recent chunk lives in slot i+1 (only valid for calls that fit one chunk)
int vs_aligner_read_level_image(const vs_aligner* a, int i, int level, uint8_t* out) try {
    VS_ARG(a && out && level >= 0 && level < a->levels && i >= 0 && i < a->last_n);
    const LevelDims& l = a->L[level];
    VS_HIP(hipSetDevice(a->device));
    VS_HIP(hipMemcpy(out, a->pyr + (size_t)(i + 1) * a->pyr_frame + l.img_off, (size_t)l.w * l.h, hipMemcpyDeviceToHost));
    return VS_OK;
} VS_CATCH_ALL
int vs_aligner_read_level_argmax(const vs_aligner* a, int i, int level, int set, uint16_t* out) try {
    VS_ARG(a && out && level >= 0 && level < a->levels && i >= 0 && i < a->last_n && (set == 0 || set == 1));
    const LevelDims& l = a->L[level];
    VS_HIP(hipSetDevice(a->device));
    // the table holds {x, y} pairs per tile; the API hands out the reference's planar form (x[nt] then y[nt])
    std::vector<uint16_t> pairs((size_t)l.nt * 2);
    VS_HIP(hipMemcpy(pairs.data(), a->lm + (size_t)(i + 1) * a->lm_frame + l.lm_off + (size_t)set * 2 * l.nt, (size_t)l.nt * 4,
                     hipMemcpyDeviceToHost));
    for (int t = 0; t < l.nt; t++) { out[t] = pairs[2 * (size_t)t]; out[(size_t)l.nt + t] = pairs[2 * (size_t)t + 1]; }
    return VS_OK;
} VS_CATCH_ALL
int vs_aligner_read_level_jacobian(const vs_aligner* a, int i, int level, int set, float* out) try {
    VS_ARG(a && out && level >= 0 && level < a->levels && i >= 0 && i < a->last_n && (set == 0 || set == 1));
    const LevelDims& l = a->L[level];
    VS_HIP(hipSetDevice(a->device));
    // the table holds one float4 per tile; the API hands out the reference's planar form (4 planes of nt)
    std::vector<float> quads((size_t)l.nt * 4);
    VS_HIP(hipMemcpy(quads.data(), a->jac + (size_t)(i + 1) * a->jac_frame + l.jac_off + (size_t)set * 4 * l.nt, (size_t)l.nt * 16,
                     hipMemcpyDeviceToHost));
    for (int t = 0; t < l.nt; t++)
        for (int c = 0; c < 4; c++) out[(size_t)c * l.nt + t] = quads[4 * (size_t)t + c];
    return VS_OK;
} VS_CATCH_ALL

}  // extern "C"

// =================================================================================================
// VideoStabilizer (stabilizer.cpp:3-117): scalar bookkeeping on the host, frames stay in HBM
// =================================================================================================
struct vs_stabilizer {
    vs_stabilizer_params params;
    vs_aligner* aligner = nullptr;
    vs_smoother* smoother = nullptr;
    int frame_index = 0;
    std::deque<vs_transform> measurements;
    struct Held { void* ptr; bool owned; };   // owned: a buffer of ours; else a frame of the batch being processed
    std::deque<Held> frames;       // the buffered input frames (stabilizer.cpp:15), dense, in device memory
    std::vector<void*> pool;       // recycled frame buffers
    size_t frame_bytes = 0;
    void* batch_in = nullptr; size_t batch_in_bytes = 0;     // dense device copy of the current batch
    // host callers: cropped outputs of the current chunk on their way down.  Two areas, used alternately by the chunks of a
    // pipelined batch; a downloader thread drains area k on down_stream while the next chunk is computed into area k^1.
    void* batch_out[2] = {nullptr, nullptr}; size_t batch_out_bytes[2] = {0, 0};
    std::future<hipError_t> down[2];
    hipEvent_t down_ev[2] = {nullptr, nullptr};
    hipStream_t down_stream = nullptr, up_stream = nullptr;
    void* pipe_in[2] = {nullptr, nullptr}; size_t pipe_in_bytes = 0;     // upload areas of a pipelined host batch
    // PITCHED host frames travel as ONE linear copy of their whole span (gaps included) into a device area and are made dense by device-to-device
    // 2-D copies: the HIP runtime never gets a 2-D copy out of pageable caller memory (profiles/r06_flake.md); dense frames: one linear copy as ever
    void* span_in[2] = {nullptr, nullptr}; size_t span_in_bytes[2] = {0, 0};
    // device-resident clip batches: the warps of clip group g run on warp_stream under the alignment of group g + 1 (stab_run)
    hipStream_t warp_stream = nullptr;
    hipEvent_t warp_ev = nullptr;
    bool overlap_warps = false;    // set by stab_run around the group / chunk calls
    bool defer_own = false;        // set for all but the last time chunk of one long device-resident clip: frames still queued stay
                                   // pointers into the caller's batch (it outlives the call), only the last chunk copies them out
    std::vector<void*> held_release;   // buffers whose last reader is a warp on warp_stream: back into the pool after its synchronisation
    // alignment results of the chunk being processed [tb] and of the chunk whose alignment is already running [tb ^ 1]
    std::vector<vs_transform> t_buf[2];
    std::vector<int32_t> st_buf[2];
    int tb = 0;
    bool prefetched = false;       // the alignment of the next stab_run_impl call's frames has been started by the previous call
    const void* next_frames = nullptr; int next_n = 0;   // set by stab_run: the chunk after the one being processed (0: none)
    vs_transform accum{0, 0, 0, 0}, last_meas{0, 0, 0, 0};
    int last_success = 0;
    int w = 0, h = 0, fmt = -1;
};

static void stab_drop_frames(vs_stabilizer* s) {
    for (auto& f : s->frames) if (f.owned) (void)hipFree(f.ptr);
    for (void* p : s->pool) (void)hipFree(p);
    s->frames.clear();
    s->pool.clear();
}

extern "C" {

vs_stabilizer* vs_stabilizer_create(const vs_stabilizer_params* params, int device) try {
    vs_stabilizer_params p;
    if (params) p = *params; else vs_stabilizer_params_default(&p);
    vs_aligner* a = vs_aligner_create(&p.aligner, device);
    if (!a) return nullptr;
    // the aligner owns streams and device slabs: whatever fails from here on releases it (a host allocation that throws included)
    struct Guard { vs_aligner* a; ~Guard() { if (a) vs_aligner_destroy(a); } } guard{a};
    vs_stabilizer* s = new vs_stabilizer();
    s->params = p;
    s->aligner = a;
    guard.a = nullptr;                                     // (from here vs_stabilizer_destroy releases it)
    s->smoother = vs_smoother_create(p.lag, p.smoother_memory, p.lambda);   // stabilizer.cpp:4
    if (!s->smoother) { vs_stabilizer_destroy(s); return nullptr; }         // (last error: the smoother's)
    return s;
} VS_CATCH_ALL_NULL

void vs_stabilizer_destroy(vs_stabilizer* s) {
    if (!s) return;
    (void)hipSetDevice(s->aligner->device);
    stab_drop_frames(s);
    for (auto& f : s->down) if (f.valid()) (void)f.get();
    if (s->batch_in) (void)hipFree(s->batch_in);
    for (void* q : s->batch_out) if (q) (void)hipFree(q);
    for (void* q : s->pipe_in) if (q) (void)hipFree(q);
    for (void* q : s->span_in) if (q) (void)hipFree(q);
    for (hipEvent_t e : s->down_ev) if (e) (void)hipEventDestroy(e);
    if (s->warp_stream) { (void)vsi::retire_stream(s->warp_stream); (void)hipStreamDestroy(s->warp_stream); }
    if (s->warp_ev) (void)hipEventDestroy(s->warp_ev);
    if (s->down_stream) (void)hipStreamDestroy(s->down_stream);
    if (s->up_stream) (void)hipStreamDestroy(s->up_stream);
    vs_smoother_destroy(s->smoother);
    vs_aligner_destroy(s->aligner);
    delete s;
}

// n successive VideoStabilizer::processFrame calls (stabilizer.cpp:9-117) as one batch: one batched alignment,
// the scalar bookkeeping on the host exactly as the reference orders it, then batched warps of every frame that
// became due.  has_output[i] = 1 when input frame i produced an output, written to out + i*out_frame_stride.
int vs_stabilizer_reset(vs_stabilizer* s);

// n successive processFrame calls; clip_len > 0: the n frames are n / clip_len independent clips, each run through a
// fresh stabilizer (reset before every clip and after the last), all of them aligned and warped together.
static int stab_run_impl(vs_stabilizer* s, const void* frames, size_t frame_stride, int n, int clip_len, int w, int h, int stride,
                         int format, int in_mem, int out_mem, int slot, void* out, size_t out_frame_stride, int32_t* has_output,
                         int* out_w, int* out_h);
static int stab_run_host_pipelined(vs_stabilizer* s, const void* frames, size_t frame_stride, int n, int clip_len, int chunk, int w,
                                   int h, int stride, int format, void* out, size_t out_frame_stride, int32_t* has_output, int* out_w,
                                   int* out_h);

// While a batch is in flight the frame queue holds non-owned pointers into the caller's buffer (or into batch_in); they
// become copies of our own only at the end of a successful run.  Whatever stops a run early -- a HIP error, a refused
// warp -- must not leave such an entry behind for the next call to warp from: the stabilizer is reset to a clean
// "new clip" state (and the stream drained, so nothing still reads the caller's frames), and the error is passed on.
static int stab_run(vs_stabilizer* s, const void* frames, size_t frame_stride, int n, int clip_len, int w, int h, int stride,
                    int format, int mem, void* out, size_t out_frame_stride, int32_t* has_output, int* out_w, int* out_h) {
    // host-resident batches longer than one upload chunk run as a three-stage pipeline: upload / compute / download
    int chunk = 0;
    if (s && mem == VS_MEM_HOST && n > 1 && w > 0 && h > 0) {
        const size_t fb = (size_t)w * h * 3 * (vs_format_bits(format) > 8 ? 2 : 1);
        chunk = (int)std::max<size_t>(4, kIngestBytes / std::max<size_t>(1, fb));
        if (clip_len > 0) chunk = std::max(clip_len, chunk - chunk % clip_len);
    }
    int r;
    // Device-resident clip batches (vs_stabilizer_process_clips, VS_MEM_DEVICE, dense frames): the clips are cut into groups and the
    // warps of group g go to a stream of their own, so that they run under the alignment of group g + 1 -- which then takes the
    // small-footprint solver build (VS_BATCH_SHARED: it shares CUs with the warp grid).  Every group goes through stab_run_impl
    // exactly as a process_clips call of its own would (clips are independent: stabilizer.cpp keeps no state across a reset), so
    // the grouping cannot change results.  VS_STAB_OVERLAP=0 turns it off.
    static const bool overlap_env = []() { const char* e = getenv("VS_STAB_OVERLAP"); return e ? atoi(e) != 0 : true; }();
    static const bool prefetch_env = []() { const char* e = getenv("VS_STAB_PREFETCH"); return e ? atoi(e) != 0 : true; }();
    // (the solver build under the overlapped warps: the small-footprint one beside a Lanczos2 warp, which fills the CUs for longer than the
    // alignment pass takes; beside the fixed-point bilinear warp -- a quarter of the alignment pass -- the exclusive 512-thread build, whose
    // shorter solver chain is worth more than the shared CUs: 1080p x 480 frames 101 k -> 120 k frames/s, 4K x 240 21.1 k -> 34.3 k
    // (profiles/r05_stab_cv_solver.txt; VS_STAB_CV_SOLVER=1 selects the small build for an A/B))
    static const int cv_solver_env = []() { const char* e = getenv("VS_STAB_CV_SOLVER"); return e && atoi(e) == VS_BATCH_SHARED ? VS_BATCH_SHARED : VS_BATCH_EXCLUSIVE; }();
    const int overlap_mode = s && s->params.warp_mode == VS_WARP_BILINEAR_CV ? cv_solver_env : VS_BATCH_SHARED;
    const int n_clips_all = clip_len > 0 ? n / clip_len : 0;
    const bool dense_dev = s && mem == VS_MEM_DEVICE && w > 0 && stride == 3 * w && frame_stride == (size_t)h * stride;
    int group_clips = 0;
    if (overlap_env && dense_dev && clip_len >= 2 && n_clips_all >= 2 && n == n_clips_all * clip_len) {
        // groups of at least kSharedMinPairs pairs (the small build's threshold), at most 4 groups (VS_STAB_GROUPS): every group boundary is a host
        // synchronisation and a latency-bound solver launch -- c5 (8 clips x 60 x 4K 10-bit) 17.3-18.1 k frames/s with 8 groups, 18.9-19.5 k with 4
        group_clips = std::max(1, (kSharedMinPairs + clip_len - 2) / (clip_len - 1));
        // (beside the fixed-point bilinear warp, with the exclusive solver build: 2 groups -- c5 27.8 k frames/s with 4 groups, 28.9 k with 2, 24.8 k with 8)
        static const int groups_env = []() { const char* e = getenv("VS_STAB_GROUPS"); const int v = e ? atoi(e) : 0; return v >= 1 ? v : 0; }();
        const int max_groups = groups_env ? groups_env : (overlap_mode == VS_BATCH_EXCLUSIVE ? 2 : 4);
        group_clips = std::max(group_clips, (n_clips_all + max_groups - 1) / max_groups);
        if (group_clips >= n_clips_all) group_clips = 0;
    }
    // one long clip: time chunks of >= 48 frames (the small solver build's threshold with room to spare), at most 4 of them (VS_STAB_TIME_CHUNKS:
    // 1080p x480 64 k frames/s with 8 chunks, 70 k with 6, 72 k with 4 or 3, 69-71 k with 2; profiles/r04_stab_long_clip.txt)
    int time_chunk = 0;
    static const int max_time_chunks = []() { const char* e = getenv("VS_STAB_TIME_CHUNKS"); const int v = e ? atoi(e) : 0; return v >= 1 ? v : 4; }();
    // (with the exclusive solver build a chunk is a latency-bound chain of its own: chunks of >= 120 frames -- 4K x240 34.2 k frames/s in 4 chunks, 35.8 k in 2)
    if (overlap_env && dense_dev && clip_len == 0 && n >= 96)
        time_chunk = std::max(overlap_mode == VS_BATCH_EXCLUSIVE ? 120 : 48, (n + max_time_chunks - 1) / max_time_chunks);
    if (time_chunk >= n) time_chunk = 0;
    if (chunk > 0 && n > chunk)
        r = vsi::guarded([&] { return stab_run_host_pipelined(s, frames, frame_stride, n, clip_len, chunk, w, h, stride, format, out, out_frame_stride,
                                                              has_output, out_w, out_h); });
    else if (group_clips > 0) {
        vs_aligner* a = s->aligner;
        r = 0;
        hipError_t he = hipSetDevice(a->device);
        if (he == hipSuccess && !s->warp_stream) he = hipStreamCreateWithFlags(&s->warp_stream, hipStreamNonBlocking);
        if (he == hipSuccess && !s->warp_ev) he = hipEventCreateWithFlags(&s->warp_ev, hipEventDisableTiming);
        // whatever the handle's stream was told to wait for (vs_stabilizer_wait_stream) holds for the warps too
        if (he == hipSuccess) he = hipEventRecord(s->warp_ev, a->stream);
        if (he == hipSuccess) he = hipStreamWaitEvent(s->warp_stream, s->warp_ev, 0);
        if (he != hipSuccess) r = set_error(VS_ERR_HIP, "stabilizer warp stream: %s", hipGetErrorString(he));
        const int saved_mode = a->batch_mode;
        a->batch_mode = overlap_mode;
        s->overlap_warps = true;
        const size_t esz = vs_format_bits(format) > 8 ? 2 : 1;
        for (int c0 = 0; r >= 0 && c0 < n_clips_all; c0 += group_clips) {
            const int nc = std::min(group_clips, n_clips_all - c0), f0 = c0 * clip_len;
            const int nc_next = std::min(group_clips, n_clips_all - c0 - nc);
            s->next_n = prefetch_env ? std::max(0, nc_next) * clip_len : 0;
            s->next_frames = (const uint8_t*)frames + (size_t)(f0 + nc * clip_len) * frame_stride * esz;
            const int rg = stab_run_impl(s, (const uint8_t*)frames + (size_t)f0 * frame_stride * esz, frame_stride, nc * clip_len, clip_len, w, h,
                                         stride, format, mem, mem, -1, (uint8_t*)out + (size_t)f0 * out_frame_stride * esz, out_frame_stride,
                                         has_output + f0, out_w, out_h);
            r = rg < 0 ? rg : r + rg;
        }
        s->overlap_warps = false;
        s->next_n = 0;
        a->batch_mode = saved_mode;
        const hipError_t we = s->warp_stream ? hipStreamSynchronize(s->warp_stream) : hipSuccess;   // every warp has landed before the call returns
        if (we != hipSuccess && r >= 0) r = set_error(VS_ERR_HIP, "stabilizer warps: %s", hipGetErrorString(we));
        for (void* b : s->held_release) s->pool.push_back(b);
        s->held_release.clear();
    } else if (time_chunk > 0) {
        // ONE long device-resident clip: cut in time.  The batched form is n successive process calls, so the chunks are the same
        // calls in the same order; the warps of chunk c (on warp_stream) run under the alignment of chunk c + 1.  Frames still
        // queued at a chunk boundary stay pointers into the caller's batch until the last chunk copies them out; buffers of
        // earlier calls whose last reader is a warp on warp_stream return to the pool only after that stream's synchronisation.
        vs_aligner* a = s->aligner;
        r = 0;
        hipError_t he = hipSetDevice(a->device);
        if (he == hipSuccess && !s->warp_stream) he = hipStreamCreateWithFlags(&s->warp_stream, hipStreamNonBlocking);
        if (he == hipSuccess && !s->warp_ev) he = hipEventCreateWithFlags(&s->warp_ev, hipEventDisableTiming);
        if (he == hipSuccess) he = hipEventRecord(s->warp_ev, a->stream);
        if (he == hipSuccess) he = hipStreamWaitEvent(s->warp_stream, s->warp_ev, 0);
        if (he != hipSuccess) r = set_error(VS_ERR_HIP, "stabilizer warp stream: %s", hipGetErrorString(he));
        const int saved_mode = a->batch_mode;
        a->batch_mode = overlap_mode;
        s->overlap_warps = true;
        const size_t esz = vs_format_bits(format) > 8 ? 2 : 1;
        for (int f0 = 0; r >= 0 && f0 < n; f0 += time_chunk) {
            const int m = std::min(time_chunk, n - f0);
            s->defer_own = f0 + m < n;
            s->next_n = prefetch_env ? std::max(0, std::min(time_chunk, n - f0 - m)) : 0;
            s->next_frames = (const uint8_t*)frames + (size_t)(f0 + m) * frame_stride * esz;
            const int rg = stab_run_impl(s, (const uint8_t*)frames + (size_t)f0 * frame_stride * esz, frame_stride, m, 0, w, h, stride, format, mem,
                                         mem, -1, (uint8_t*)out + (size_t)f0 * out_frame_stride * esz, out_frame_stride, has_output + f0, out_w,
                                         out_h);
            r = rg < 0 ? rg : r + rg;
        }
        s->defer_own = false;
        s->overlap_warps = false;
        s->next_n = 0;
        a->batch_mode = saved_mode;
        const hipError_t we = s->warp_stream ? hipStreamSynchronize(s->warp_stream) : hipSuccess;
        if (we != hipSuccess && r >= 0) r = set_error(VS_ERR_HIP, "stabilizer warps: %s", hipGetErrorString(we));
        for (void* b : s->held_release) s->pool.push_back(b);
        s->held_release.clear();
    } else
        r = stab_run_impl(s, frames, frame_stride, n, clip_len, w, h, stride, format, mem, mem, -1, out, out_frame_stride, has_output,
                          out_w, out_h);
    if (s) for (auto& f : s->down) if (f.valid()) {          // every download has landed before the call returns
        const hipError_t de = f.get();
        if (de != hipSuccess && r >= 0) r = set_error(VS_ERR_HIP, "output download failed: %s", hipGetErrorString(de));
    }
    if (r < 0 && s && s->aligner) {
        const std::string why = vs_last_error();             // the reset below must not hide the cause
        align_abandon(s->aligner);                           // (a next chunk's alignment may have been started)
        s->prefetched = false; s->next_n = 0;
        (void)hipStreamSynchronize(s->aligner->stream);
        for (auto it = s->frames.begin(); it != s->frames.end();) it = it->owned ? it + 1 : s->frames.erase(it);
        (void)vs_stabilizer_reset(s);
        set_error(r, "%s", why.c_str());
    }
    return r;
}

// The batch split into chunks of `chunk` frames (whole clips in clip mode): an uploader thread fills the other upload area
// with chunk c+1 while chunk c is aligned and warped, and a downloader thread drains chunk c's outputs while chunk c+1 is
// computed -- upload, compute and download overlap, and the link carries input and output at the same time (full duplex).
// Every chunk goes through stab_run_impl exactly as a separate vs_stabilizer_process_batch call would, which is the
// definition of the batched form ("n successive process calls"), so the results do not depend on the chunking.
static int stab_run_host_pipelined(vs_stabilizer* s, const void* frames, size_t frame_stride, int n, int clip_len, int chunk, int w,
                                   int h, int stride, int format, void* out, size_t out_frame_stride, int32_t* has_output, int* out_w,
                                   int* out_h) {
    VS_ARG(s && frames && out && has_output && out_w && out_h);
    VS_ARG(w > 0 && h > 0 && w <= 65535 && h <= 65535);
    VS_ARG(format != VS_FMT_GRAY8 && vs_format_bits(format) != 0 && stride >= 3 * w);
    VS_ARG(frame_stride >= (size_t)(h - 1) * stride + (size_t)3 * w);
    vs_aligner* a = s->aligner;
    VS_HIP(hipSetDevice(a->device));
    const size_t esz = vs_format_bits(format) > 8 ? 2 : 1;
    const size_t fbytes = (size_t)w * h * 3 * esz;
    if (s->pipe_in_bytes < fbytes * chunk) {
        for (void*& q : s->pipe_in) { if (q) (void)hipFree(q); q = nullptr; }
        s->pipe_in_bytes = 0;
        VS_HIP(vsi::dev_alloc(&s->pipe_in[0], fbytes * chunk));
        VS_HIP(vsi::dev_alloc(&s->pipe_in[1], fbytes * chunk));
        s->pipe_in_bytes = fbytes * chunk;
    }
    if (!s->up_stream) VS_HIP(hipStreamCreateWithFlags(&s->up_stream, hipStreamNonBlocking));
    const bool dense = stride == 3 * w && frame_stride == (size_t)h * stride;
    auto upload = [=](int c) -> hipError_t {                 // chunk c -> pipe_in[c & 1], dense
        const int off = c * chunk, m = std::min(chunk, n - off);
        hipError_t e = hipSetDevice(a->device);
        const uint8_t* src = (const uint8_t*)frames + (size_t)off * frame_stride * esz;
        if (e == hipSuccess && dense) e = hipMemcpyAsync(s->pipe_in[c & 1], src, fbytes * m, hipMemcpyHostToDevice, s->up_stream);
        if (e == hipSuccess && !dense) {                     // pitched: the chunk's span in one linear copy, then dense by device-to-device 2-D copies
            const size_t span = ((size_t)(m - 1) * frame_stride + (size_t)(h - 1) * stride + (size_t)3 * w) * esz;
            void*& area = s->span_in[c & 1];
            if (s->span_in_bytes[c & 1] < span) {
                if (area) (void)hipFree(area);
                area = nullptr; s->span_in_bytes[c & 1] = 0;
                e = vsi::dev_alloc(&area, span);
                if (e == hipSuccess) s->span_in_bytes[c & 1] = span;
            }
            if (e == hipSuccess) e = hipMemcpyAsync(area, src, span, hipMemcpyHostToDevice, s->up_stream);
            for (int i = 0; e == hipSuccess && i < m; i++)
                e = hipMemcpy2DAsync((uint8_t*)s->pipe_in[c & 1] + (size_t)i * fbytes, (size_t)w * 3 * esz, (const uint8_t*)area + (size_t)i * frame_stride * esz,
                                     (size_t)stride * esz, (size_t)w * 3 * esz, h, hipMemcpyDeviceToDevice, s->up_stream);
        }
        return e != hipSuccess ? e : hipStreamSynchronize(s->up_stream);
    };
    const int n_chunks = (n + chunk - 1) / chunk;
    std::future<hipError_t> next = run_async(upload, 0);
    int produced = 0;
    for (int c = 0; c < n_chunks; c++) {
        const int off = c * chunk, m = std::min(chunk, n - off);
        const hipError_t ue = next.get();
        if (c + 1 < n_chunks) next = run_async(upload, c + 1);
        int r = ue == hipSuccess ? VS_OK : set_error(VS_ERR_HIP, "frame upload failed: %s", hipGetErrorString(ue));
        if (r == VS_OK)
            r = stab_run_impl(s, s->pipe_in[c & 1], (size_t)w * h * 3, m, clip_len, w, h, 3 * w, format, VS_MEM_DEVICE, VS_MEM_HOST, c & 1,
                              (uint8_t*)out + (size_t)off * out_frame_stride * esz, out_frame_stride, has_output + off, out_w, out_h);
        if (r < 0) { if (next.valid()) (void)next.get(); return r; }
        produced += r;
    }
    return produced;
}

static int stab_run_impl_unguarded(vs_stabilizer* s, const void* frames, size_t frame_stride, int n, int clip_len, int w, int h, int stride,
                         int format, int mem, int out_mem, int slot_arg, void* out, size_t out_frame_stride, int32_t* has_output,
                         int* out_w, int* out_h) {
    // slot_arg >= 0: a chunk of the pipelined host batch -- its outputs leave through output area `slot_arg` and a downloader
    // thread of their own, under the next chunk's compute.  slot_arg < 0: a call on its own (process / process_batch that fits
    // one chunk): the copies go onto the handle's stream, nothing to overlap with, no thread.
    const bool threaded_download = slot_arg >= 0;
    const int slot = threaded_download ? slot_arg : 0;
    VS_ARG(s && frames && out && has_output && out_w && out_h && n >= 1);
    VS_ARG(format != VS_FMT_GRAY8 && vs_format_bits(format) != 0);
    VS_ARG(w > 0 && h > 0 && w <= 65535 && h <= 65535);
    VS_ARG(stride >= 3 * w);
    const int crop = s->params.crop_pixels > 0 ? s->params.crop_pixels : 0;
    VS_ARG(w > 2 * crop && h > 2 * crop);
    const int ow = w - 2 * crop, oh = h - 2 * crop;
    VS_ARG(n == 1 || (frame_stride >= (size_t)(h - 1) * stride + (size_t)3 * w && out_frame_stride >= (size_t)ow * oh * 3));
    vs_aligner* a = s->aligner;
    VS_HIP(hipSetDevice(a->device));
    hipStream_t st = a->stream;
    const int fbits = vs_format_bits(format);
    const size_t esz = fbits > 8 ? 2 : 1;
    const size_t fbytes = (size_t)w * h * 3 * esz;
    if (s->w != w || s->h != h || s->fmt != format) {
        // a size change restarts the aligner (alignment.cpp:155).  The reference would go on warping queued frames of the
        // old size with measurements of the new one; here the change starts a new clip, cleanly: queued frames of the
        // old size are dropped and the smoother, the accumulated correction and the frame counter start over.
        stab_drop_frames(s);
        VS_TRY(vs_stabilizer_reset(s));
        s->w = w; s->h = h; s->fmt = format; s->frame_bytes = fbytes;
    }
    *out_w = ow; *out_h = oh;

    // stabilizer.cpp:15: a private dense copy of every input frame, in device memory
    const uint8_t* dense = nullptr;
    const bool already_dense = mem == VS_MEM_DEVICE && stride == 3 * w && (n == 1 || frame_stride == (size_t)h * stride);
    if (already_dense) {
        dense = (const uint8_t*)frames;      // read in place during this call; the tail is copied out below
    } else {
        if (s->batch_in_bytes < fbytes * n) {
            if (s->batch_in) (void)hipFree(s->batch_in);
            s->batch_in = nullptr; s->batch_in_bytes = 0;
            VS_HIP(vsi::dev_alloc(&s->batch_in, fbytes * n));
            s->batch_in_bytes = fbytes * n;
        }
        const hipMemcpyKind kind = mem == VS_MEM_HOST ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice;
        if (stride == 3 * w && (n == 1 || frame_stride == (size_t)h * stride)) {
            VS_HIP(hipMemcpyAsync(s->batch_in, frames, fbytes * n, kind, st));   // dense input: one linear copy at the full link rate
        } else {
            const uint8_t* from = (const uint8_t*)frames;
            if (mem == VS_MEM_HOST) {                       // pitched host frames: the span in one linear copy (see span_in), 2-D copies on the device only
                const size_t span = ((size_t)(n - 1) * frame_stride + (size_t)(h - 1) * stride + (size_t)3 * w) * esz;
                if (s->span_in_bytes[0] < span) {
                    if (s->span_in[0]) (void)hipFree(s->span_in[0]);
                    s->span_in[0] = nullptr; s->span_in_bytes[0] = 0;
                    VS_HIP(vsi::dev_alloc(&s->span_in[0], span));
                    s->span_in_bytes[0] = span;
                }
                VS_HIP(hipMemcpyAsync(s->span_in[0], frames, span, hipMemcpyHostToDevice, st));
                from = (const uint8_t*)s->span_in[0];
            }
            for (int i = 0; i < n; i++)
                VS_HIP(hipMemcpy2DAsync((uint8_t*)s->batch_in + (size_t)i * fbytes, (size_t)w * 3 * esz,
                                        from + (size_t)i * frame_stride * esz, (size_t)stride * esz,
                                        (size_t)w * 3 * esz, h, hipMemcpyDeviceToDevice, st));
        }
        dense = (const uint8_t*)s->batch_in;
    }

    // stabilizer.cpp:18-19 for all n frames.  In a chunked device-resident batch (stab_run) the alignment of the NEXT chunk is
    // started as soon as this one's results are in, so that it runs under this chunk's host work (smoother, correction chain, warp
    // launches) as well as under its warps.
    const int cur = s->tb;
    if (s->prefetched) {
        s->prefetched = false;              // started by the previous call, into [cur]
    } else {
        s->t_buf[cur].resize(n);
        s->st_buf[cur].resize(n);
        VS_TRY(align_start(a, dense, (size_t)w * h * 3, n, clip_len, w, h, w * 3, format, VS_MEM_DEVICE, &s->params.aligner,
                           s->t_buf[cur].data(), s->st_buf[cur].data(), false));
    }
    {
        const int r = align_finish(a);
        if (r < 0) return r;
    }
    if (s->next_n > 0 && already_dense) {
        s->t_buf[cur ^ 1].resize(s->next_n);
        s->st_buf[cur ^ 1].resize(s->next_n);
        VS_TRY(align_start(a, s->next_frames, (size_t)w * h * 3, s->next_n, clip_len, w, h, w * 3, format, VS_MEM_DEVICE, &s->params.aligner,
                           s->t_buf[cur ^ 1].data(), s->st_buf[cur ^ 1].data(), true));
        s->prefetched = true;
        s->tb = cur ^ 1;
    }
    const std::vector<vs_transform>& t_buf = s->t_buf[cur];
    const std::vector<int32_t>& st_buf = s->st_buf[cur];

    struct Job { const void* src; vs_transform sampling; int i; void* release; };
    std::vector<Job> jobs;
    for (int i = 0; i < n; i++) {
        if (clip_len > 0 && i % clip_len == 0) VS_TRY(vs_stabilizer_reset(s));   // a new clip: frames still queued are dropped
        ++s->frame_index;
        s->frames.push_back(vs_stabilizer::Held{(void*)(dense + (size_t)i * fbytes), false});
        const vs_transform meas = t_buf[i];
        const bool success = st_buf[i] == 1;
        s->last_meas = meas; s->last_success = success ? 1 : 0;
        has_output[i] = 0;

        vs_transform earliest_smoothed{0, 0, 0, 0};
        if (s->params.enable_smoother) (void)vs_smoother_update(s->smoother, &meas, &earliest_smoothed);   // :35
        if (!success) s->accum = vs_transform{0, 0, 0, 0};                                                  // :39-41
        s->measurements.push_back(meas);                                                                   // :44
        if (s->measurements.size() > (size_t)s->params.lag) {                                              // :48
            vs_transform earliest = s->measurements.front();
            s->measurements.pop_front();
            vs_transform jitter;
            if (s->params.enable_smoother) {
                vs_transform inv = vs_transform_inverse(&earliest_smoothed);
                jitter = vs_transform_compose(&earliest, &inv);                                            // :60
            } else {
                jitter = earliest;
            }
            vs_transform na = vs_transform_compose(&s->accum, &jitter);                                    // :66
            const double disp = vs_transform_max_corner_displacement(&na, w, h);                           // :69-70
            double decay;
            if (disp > s->params.max_disp) {
                decay = s->params.max_decay;
            } else if (disp > s->params.min_disp) {
                double f = (disp - s->params.min_disp) / (s->params.max_disp - s->params.min_disp);
                f = std::max(0.0, std::min(1.0, f));
                decay = s->params.min_decay * (1.0 - f) + s->params.max_decay * f;
            } else {
                decay = s->params.min_decay;
            }
            na.TX *= decay; na.TY *= decay; na.A *= decay; na.B *= decay;                                  // :88-91
            s->accum = na;
            if (!s->frames.empty()) {
                vs_stabilizer::Held src = s->frames.front();
                s->frames.pop_front();
                // :97-99: warpBySimilarityTransform(frame, accum^-1); cv::warpAffine without WARP_INVERSE_MAP
                // inverts the matrix it is given (imgproc.cpp:472), so the sampling map is (accum^-1)^-1.
                // (VS_WARP_BILINEAR_CV is cv::warpAffine itself, inversion included: it takes the correction as the reference hands it over)
                vs_transform correction = vs_transform_inverse(&na);
                jobs.push_back(Job{src.ptr, s->params.warp_mode == VS_WARP_BILINEAR_CV ? correction : vs_transform_inverse(&correction), i,
                                   src.owned ? src.ptr : nullptr});
                has_output[i] = 1;
            }
        }
    }

    // warp every due frame, runs of consecutive batch frames as one launch.  The crop of stabilizer.cpp:102-109 is the
    // output window of the warp: the margin is never computed and no full-size intermediate frame exists.  Device callers
    // get the window written straight into `out`; host callers through a dense staging buffer and one copy per frame.
    if (!jobs.empty()) {
        const size_t obytes = (size_t)ow * oh * 3 * esz;
        const bool to_host = out_mem == VS_MEM_HOST;
        if (to_host) {
            if (s->down[slot].valid()) {                     // the previous user of this area has been drained
                const hipError_t de = s->down[slot].get();
                if (de != hipSuccess) return set_error(VS_ERR_HIP, "output download failed: %s", hipGetErrorString(de));
            }
            if (s->batch_out_bytes[slot] < obytes * jobs.size()) {
                if (s->batch_out[slot]) (void)hipFree(s->batch_out[slot]);
                s->batch_out[slot] = nullptr; s->batch_out_bytes[slot] = 0;
                VS_HIP(vsi::dev_alloc(&s->batch_out[slot], obytes * jobs.size()));
                s->batch_out_bytes[slot] = obytes * jobs.size();
            }
        }
        std::vector<vs_transform> ts;
        for (size_t j = 0; j < jobs.size();) {
            size_t e = j + 1;
            while (e < jobs.size() && (const uint8_t*)jobs[e].src == (const uint8_t*)jobs[e - 1].src + fbytes &&
                   jobs[e].i == jobs[e - 1].i + 1) e++;
            ts.clear();
            for (size_t k = j; k < e; k++) ts.push_back(jobs[k].sampling);
            void* dst = to_host ? (void*)((uint8_t*)s->batch_out[slot] + j * obytes)
                                : (void*)((uint8_t*)out + (size_t)jobs[j].i * out_frame_stride * esz);
            const size_t dst_fs = to_host ? (size_t)ow * oh * 3 : out_frame_stride;
            // (device output of an overlapped clip batch: the warps go to warp_stream and run under the next group's alignment)
            hipStream_t ws = (s->overlap_warps && !to_host) ? s->warp_stream : st;
            // (beside the next group's alignment the Lanczos2 warp keeps its standard window: see vsi::warp_keeps_solver_slot)
            struct SlotHint { bool& f; bool old; SlotHint(bool on) : f(vsi::warp_keeps_solver_slot()), old(f) { f = on; } ~SlotHint() { f = old; } } hint(s->overlap_warps);
            int wr = vs_bgr_image_warp_roi_batch(jobs[j].src, (size_t)w * h * 3, (int)(e - j), w, h, w * 3, 3, (int)esz * 8, ts.data(),
                                                 s->params.warp_mode, s->params.warp_border, vs_format_max_value(format), crop, crop, ow, oh,
                                                 dst, dst_fs, ow * 3, VS_MEM_DEVICE, ws);
            if (wr < 0) return wr;
            j = e;
        }
        for (size_t j = 0; j < jobs.size(); j++)
            if (jobs[j].release) {
                if (s->overlap_warps && !to_host) s->held_release.push_back(jobs[j].release);   // read on warp_stream, refilled on st
                else s->pool.push_back(jobs[j].release);                                        // reused only by later work on this stream
            }
        if (to_host) {
            const bool dense_out = out_frame_stride * esz == obytes;
            if (!threaded_download) {
                // runs of outputs that are contiguous on both sides as one copy, behind the warps on the same stream; the
                // synchronisation at the end of this call covers them
                for (size_t j = 0; j < jobs.size();) {
                    size_t k = j + 1;
                    while (dense_out && k < jobs.size() && jobs[k].i == jobs[k - 1].i + 1) k++;
                    VS_HIP(hipMemcpyAsync((uint8_t*)out + (size_t)jobs[j].i * out_frame_stride * esz,
                                          (const uint8_t*)s->batch_out[slot] + j * obytes, obytes * (k - j), hipMemcpyDeviceToHost, st));
                    j = k;
                }
            } else {
            // hand the area to the downloader: it waits (on its own stream) for the warps above, then copies every output
            // to the caller's memory -- runs of outputs that are contiguous on both sides as one copy
            if (!s->down_stream) VS_HIP(hipStreamCreateWithFlags(&s->down_stream, hipStreamNonBlocking));
            if (!s->down_ev[slot]) VS_HIP(hipEventCreateWithFlags(&s->down_ev[slot], hipEventDisableTiming));
            VS_HIP(hipEventRecord(s->down_ev[slot], st));
            std::vector<int> idx(jobs.size());
            for (size_t j = 0; j < jobs.size(); j++) idx[j] = jobs[j].i;
            const int device = a->device;
            const uint8_t* area = (const uint8_t*)s->batch_out[slot];
            hipStream_t ds = s->down_stream;
            hipEvent_t ev = s->down_ev[slot];
            const bool out_dense = out_frame_stride * esz == obytes;
            s->down[slot] = run_async([=]() -> hipError_t {
                hipError_t e = hipSetDevice(device);
                if (e == hipSuccess) e = hipStreamWaitEvent(ds, ev, 0);
                for (size_t j = 0; e == hipSuccess && j < idx.size();) {
                    size_t k = j + 1;
                    while (out_dense && k < idx.size() && idx[k] == idx[k - 1] + 1) k++;
                    e = hipMemcpyAsync((uint8_t*)out + (size_t)idx[j] * out_frame_stride * esz, area + j * obytes, obytes * (k - j),
                                       hipMemcpyDeviceToHost, ds);
                    j = k;
                }
                return e != hipSuccess ? e : hipStreamSynchronize(ds);
            });
            }
        }
    }
    if (clip_len > 0) VS_TRY(vs_stabilizer_reset(s));   // nothing carries over from the last clip
    // frames of this batch that are still queued move into buffers of our own
    for (auto& f : s->frames) {
        if (f.owned || s->defer_own) continue;
        void* copy = nullptr;
        if (!s->pool.empty()) { copy = s->pool.back(); s->pool.pop_back(); }
        else VS_HIP(vsi::dev_alloc(&copy, fbytes));
        VS_HIP(hipMemcpyAsync(copy, f.ptr, fbytes, hipMemcpyDeviceToDevice, st));
        f.ptr = copy; f.owned = true;
    }
    if (!s->prefetched) VS_HIP(hipStreamSynchronize(st));   // (with the next chunk's alignment in flight its completion is the next call's wait)
    int produced = 0;
    for (int i = 0; i < n; i++) produced += has_output[i];
    return produced;
}
static int stab_run_impl(vs_stabilizer* s, const void* frames, size_t frame_stride, int n, int clip_len, int w, int h, int stride,
                         int format, int mem, int out_mem, int slot_arg, void* out, size_t out_frame_stride, int32_t* has_output,
                         int* out_w, int* out_h) {
    // (guarded: an exception inside a chunk -- a host allocation that fails -- comes back as an error code, so that the callers' loops restore
    // the handle's modes and stab_run's failure protocol runs)
    return vsi::guarded([&] { return stab_run_impl_unguarded(s, frames, frame_stride, n, clip_len, w, h, stride, format, mem, out_mem, slot_arg, out, out_frame_stride, has_output, out_w, out_h); });
}

int vs_stabilizer_process_batch(vs_stabilizer* s, const void* frames, size_t frame_stride, int n, int w, int h, int stride,
                                int format, int mem, void* out, size_t out_frame_stride, int32_t* has_output, int* out_w,
                                int* out_h) try {
    return stab_run(s, frames, frame_stride, n, 0, w, h, stride, format, mem, out, out_frame_stride, has_output, out_w, out_h);
} VS_CATCH_ALL

int vs_stabilizer_process_clips(vs_stabilizer* s, const void* frames, size_t frame_stride, int n_clips, int frames_per_clip,
                                int w, int h, int stride, int format, int mem, void* out, size_t out_frame_stride,
                                int32_t* has_output, int* out_w, int* out_h) try {
    VS_ARG(n_clips >= 1 && frames_per_clip >= 1 && (long long)n_clips * frames_per_clip <= 0x7fffffff);
    return stab_run(s, frames, frame_stride, n_clips * frames_per_clip, frames_per_clip, w, h, stride, format, mem, out,
                    out_frame_stride, has_output, out_w, out_h);
} VS_CATCH_ALL

// forget the clip: the next frame starts a new sequence (device buffers are kept)
void* vs_stabilizer_stream(const vs_stabilizer* s) { return s && s->aligner ? (void*)s->aligner->stream : nullptr; }
int vs_stabilizer_set_select_mode(vs_stabilizer* s, int mode) try {
    VS_ARG(s && s->aligner);
    return vs_aligner_set_select_mode(s->aligner, mode);
} VS_CATCH_ALL
int vs_stabilizer_get_select_mode(const vs_stabilizer* s) try {
    VS_ARG(s && s->aligner);
    return s->aligner->select_mode;
} VS_CATCH_ALL
int vs_stabilizer_wait_stream(vs_stabilizer* s, void* producer_stream) try {
    VS_ARG(s && s->aligner);
    return vs_aligner_wait_stream(s->aligner, producer_stream);
} VS_CATCH_ALL

int vs_stabilizer_reset(vs_stabilizer* s) try {
    VS_ARG(s);
    VS_HIP(hipSetDevice(s->aligner->device));
    for (auto& f : s->frames) if (f.owned) s->pool.push_back(f.ptr);
    s->frames.clear();
    s->measurements.clear();
    // (make the new smoother first: if that fails the handle keeps a valid, if stale, one -- never a null pointer for the next call to walk into)
    vs_smoother* fresh = vs_smoother_create(s->params.lag, s->params.smoother_memory, s->params.lambda);
    if (!fresh) return VS_ERR_NOMEM;
    vs_smoother_destroy(s->smoother);
    s->smoother = fresh;
    s->accum = vs_transform{0, 0, 0, 0};
    s->last_meas = vs_transform{0, 0, 0, 0};
    s->last_success = 0;
    s->frame_index = 0;
    return vs_aligner_reset(s->aligner);
} VS_CATCH_ALL

int vs_stabilizer_process(vs_stabilizer* s, const void* frame, int w, int h, int stride, int format, int mem, void* out,
                          int* out_w, int* out_h) try {
    int32_t has = 0;
    int r = vs_stabilizer_process_batch(s, frame, 0, 1, w, h, stride, format, mem, out, 0, &has, out_w, out_h);
    return r < 0 ? r : has;
} VS_CATCH_ALL

void vs_stabilizer_state(const vs_stabilizer* s, vs_transform* last_meas, vs_transform* accum, int* last_success) {
    if (last_meas) *last_meas = s->last_meas;
    if (accum) *accum = s->accum;
    if (last_success) *last_success = s->last_success;
}

}  // extern "C"
