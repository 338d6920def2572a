// vs_capi.hip -- kernel-level entry points of the C ABI (include/vs_amd.h).
// Each mirrors one Halide AOT function the reference's imgproc.cpp calls; see the header for the
// file:line of the interface it replaces.  VS_MEM_HOST stages through device memory and syncs;
// VS_MEM_DEVICE only enqueues on the caller's stream.
#include "vs_internal.hpp"
#include "vs_kernels.hpp"
#include "vs_phase.hpp"

#include <algorithm>
#include <atomic>
#include <new>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <mutex>
#include <vector>

namespace vsi {

bool device_ready() {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        set_error(VS_ERR_HIP, "no usable HIP device (%s): libvs_amd has no CPU fallback",
                  e == hipSuccess ? "device count 0" : hipGetErrorString(e));
        return false;
    }
    return true;
}

// ---- allocation + fault injection (tests/test_alloc_failure_gpu.py) -----------------------------------------------
// g_alloc_count counts the library's allocations since the last vs_test_fail_alloc call; the one that brings it to
// g_alloc_fail_at fails with hipErrorOutOfMemory (once: the count moves on).  VS_TEST_FAIL_ALLOC=k arms it from the
// environment for programs that cannot call the hook.
// The ENVIRONMENT-driven hooks (VS_TEST_FAIL_ALLOC, VS_TEST_POISON_*) are honoured only when VS_TEST_HOOKS=1 is set as well: a stray
// variable in a production environment must not make a real allocation fail or cost every allocation a memset.
static const char* test_env(const char* name) {
    static const bool on = []() { const char* h = getenv("VS_TEST_HOOKS"); return h && atoi(h) == 1; }();
    return on ? getenv(name) : nullptr;
}
std::atomic<long long> g_alloc_count{0};
std::atomic<long long> g_alloc_fail_at{[]() { const char* e = test_env("VS_TEST_FAIL_ALLOC"); return e ? atoll(e) : 0LL; }()};
static bool alloc_injected_failure() {
    const long long k = ++g_alloc_count, at = g_alloc_fail_at.load(std::memory_order_relaxed);
    if (at < 0 && k == -at) throw std::bad_alloc();        // vs_test_fail_alloc(-k): a HOST allocation failing at this point of the call
    return k == at;
}
hipError_t dev_alloc(void** p, size_t bytes) {
    *p = nullptr;
    if (alloc_injected_failure()) return hipErrorOutOfMemory;
    hipError_t e = hipMalloc(p, bytes);
    if (e != hipSuccess) { *p = nullptr; (void)hipGetLastError(); }      // (the sticky error belongs to this call, which reports it)
    // test hook (VS_TEST_POISON_ALLOC=<byte>): every fresh device allocation starts filled with that byte, so that a result which depends on
    // memory the library never wrote changes with the byte (tests/test_uninitialised_memory_gpu.py)
    static const int poison = []() { const char* v = test_env("VS_TEST_POISON_ALLOC"); return v ? (int)strtol(v, nullptr, 0) & 255 : -1; }();
    static const int only = []() { const char* v = test_env("VS_TEST_POISON_ONLY"); return v ? atoi(v) : 0; }();      // (0: every allocation; k: the k-th only)
    static std::atomic<int> nth{0};
    const int k = ++nth;
    if (e == hipSuccess && poison >= 0 && bytes) {
        static const long long r_off = []() { const char* v = test_env("VS_TEST_POISON_OFF"); return v ? atoll(v) : 0LL; }();
        static const long long r_len = []() { const char* v = test_env("VS_TEST_POISON_LEN"); return v ? atoll(v) : -1LL; }();
        if (only == k && r_len >= 0) {                     // (debugging aid: only bytes [off, off + len) of the k-th allocation get the byte)
            e = hipMemset(*p, 0, bytes);
            if (e == hipSuccess && r_off < (long long)bytes) e = hipMemset((char*)*p + r_off, poison, (size_t)std::min<long long>(r_len, (long long)bytes - r_off));
        } else
        e = hipMemset(*p, (only == 0 || only == k) ? poison : 0, bytes);
        if (e == hipSuccess) e = hipDeviceSynchronize();
        if (only == k) std::fprintf(stderr, "[poison] allocation %d: %zu bytes\n", k, bytes);
        if (e != hipSuccess) { (void)hipFree(*p); *p = nullptr; (void)hipGetLastError(); }     // the contract: *p is nullptr on failure
    }
    return e;
}
hipError_t pinned_alloc(void** p, size_t bytes) {
    *p = nullptr;
    if (alloc_injected_failure()) return hipErrorOutOfMemory;
    const hipError_t e = hipHostMalloc(p, bytes);
    if (e != hipSuccess) { *p = nullptr; (void)hipGetLastError(); }
    return e;
}

// ---- the staging pool ------------------------------------------------------------------------------------------------------------
// Free blocks are kept per process (any thread, any device: a block remembers its device); a lease takes the smallest free block that
// is large enough, or allocates one (capacities are rounded up so that a sweep of slightly different sizes reuses blocks).  At most
// kStageKeepBlocks / kStageKeepBytes stay cached; what does not fit is freed on release.  Nothing is freed at process exit (the HIP
// runtime may already be gone).
namespace {
constexpr size_t kStageKeepBlocks = 24, kStageKeepBytes = (size_t)3 << 30;
struct StagePool {
    std::mutex mu;
    std::vector<StageBlock*> free_;
    size_t kept_bytes = 0;
};
StagePool& stage_pool() { static StagePool* p = new StagePool(); return *p; }
size_t stage_capacity(size_t n) {
    if (n <= 4096) return 4096;
    if (n <= ((size_t)64 << 20)) { size_t c = 4096; while (c < n) c <<= 1; return c; }
    const size_t g = (size_t)64 << 20;
    return (n + g - 1) / g * g;
}
void stage_destroy(StageBlock* b) {
    if (b->dev) (void)hipFree(b->dev);
    if (b->pin) (void)hipHostFree(b->pin);
    delete b;
}
thread_local unsigned long long t_sync_epoch = 1;
}  // namespace
unsigned long long host_sync_epoch() { return t_sync_epoch; }
void host_synced() { ++t_sync_epoch; }

StageBlock* stage_acquire(size_t bytes) {
    int device = 0;
    if (hipGetDevice(&device) != hipSuccess) { (void)hipGetLastError(); set_error(VS_ERR_HIP, "hipGetDevice failed (staging)"); return nullptr; }
    const size_t cap = stage_capacity(bytes ? bytes : 1);
    StageBlock* b = nullptr;
    {
        StagePool& p = stage_pool();
        std::lock_guard<std::mutex> g(p.mu);
        size_t best = (size_t)-1;
        for (size_t i = 0; i < p.free_.size(); i++)
            if (p.free_[i]->device == device && p.free_[i]->cap >= cap && (best == (size_t)-1 || p.free_[i]->cap < p.free_[best]->cap)) best = i;
        if (best != (size_t)-1) {
            b = p.free_[best];
            p.free_.erase(p.free_.begin() + (long)best);
            p.kept_bytes -= b->cap;
        }
    }
    static const int poison = []() { const char* v = test_env("VS_TEST_POISON_ALLOC"); return v ? (int)strtol(v, nullptr, 0) & 255 : -1; }();
    if (b) {
        // test hook: a reused block starts from the poison byte like a fresh allocation (tests/test_uninitialised_memory_gpu.py)
        if (poison >= 0 && (hipMemset(b->dev, poison, b->cap) != hipSuccess || hipDeviceSynchronize() != hipSuccess)) {
            (void)hipGetLastError(); stage_destroy(b); set_error(VS_ERR_HIP, "poisoning a staging block failed"); return nullptr;
        }
        if (poison >= 0) memset(b->pin, poison, b->cap);
        return b;
    }
    b = new StageBlock();
    b->cap = cap; b->device = device;
    hipError_t e = dev_alloc(&b->dev, cap);
    if (e == hipSuccess) e = pinned_alloc(&b->pin, cap);
    if (e != hipSuccess) {
        stage_destroy(b);
        set_error(VS_ERR_HIP, "staging block of %zu bytes: %s", cap, hipGetErrorString(e));
        return nullptr;
    }
    if (poison >= 0) memset(b->pin, poison, cap);
    return b;
}
void stage_release(StageBlock* b) {
    if (!b) return;
    {
        StagePool& p = stage_pool();
        std::lock_guard<std::mutex> g(p.mu);
        if (p.free_.size() < kStageKeepBlocks && p.kept_bytes + b->cap <= kStageKeepBytes) {
            p.free_.push_back(b);
            p.kept_bytes += b->cap;
            return;
        }
        // the cache is full: drop the smallest cached block if this one is larger (large blocks are the expensive ones to make)
        size_t small = (size_t)-1;
        for (size_t i = 0; i < p.free_.size(); i++)
            if (small == (size_t)-1 || p.free_[i]->cap < p.free_[small]->cap) small = i;
        if (small != (size_t)-1 && p.free_[small]->cap < b->cap && p.kept_bytes - p.free_[small]->cap + b->cap <= kStageKeepBytes) {
            std::swap(b, p.free_[small]);
            p.kept_bytes += p.free_[small]->cap - b->cap;
        }
    }
    stage_destroy(b);
}

Staged::~Staged() {
    if (!blk) return;
    // an error return between the first enqueue and the call's synchronisation: work that reads or writes the block may still be in flight
    if (epoch == host_sync_epoch()) { (void)hipDeviceSynchronize(); (void)hipGetLastError(); }
    stage_release(blk);
}
int Staged::lease(size_t n) {
    if (blk) { stage_release(blk); blk = nullptr; }
    blk = stage_acquire(n);
    if (!blk) return VS_ERR_HIP;
    dev = blk->dev;
    epoch = host_sync_epoch();
    return VS_OK;
}
int Staged::in(const void* ptr, size_t n, int mem, hipStream_t s) {
    bytes = n; is_out = false;
    if (mem == VS_MEM_DEVICE) { dev = const_cast<void*>(ptr); staged = false; return VS_OK; }
    staged = true; host = const_cast<void*>(ptr);
    if (int r = lease(n)) return r;
    if (n) {
        memcpy(blk->pin, ptr, n);
        VS_HIP(hipMemcpyAsync(dev, blk->pin, n, hipMemcpyHostToDevice, s));
    }
    return VS_OK;
}
int Staged::out(void* ptr, size_t n, int mem) {
    bytes = n; is_out = true;
    if (mem == VS_MEM_DEVICE) { dev = ptr; staged = false; return VS_OK; }
    staged = true; host = ptr;
    return lease(n);
}
int Staged::out_image(void* ptr, size_t row_bytes_, size_t rows_, size_t pitch_, size_t frames_, size_t frame_pitch_, int mem) {
    const size_t n = frames_ == 0 || rows_ == 0 ? 0 : (frames_ - 1) * frame_pitch_ + (rows_ - 1) * pitch_ + row_bytes_;
    const int r = out(ptr, n, mem);
    const bool dense = pitch_ == row_bytes_ && (frames_ <= 1 || frame_pitch_ == rows_ * pitch_);
    if (r == VS_OK && staged && !dense) { row_bytes = row_bytes_; rows = rows_; pitch = pitch_; frames = frames_; frame_pitch = frame_pitch_; }
    return r;
}
int Staged::finish(hipStream_t s) {
    if (!(staged && is_out && bytes)) return VS_OK;
    VS_HIP(hipMemcpyAsync(blk->pin, dev, bytes, hipMemcpyDeviceToHost, s));     // dense: the gaps between rows travel too and are dropped by complete()
    copied_back = true;
    return VS_OK;
}
void Staged::complete() {
    if (!copied_back) return;
    copied_back = false;
    const char* m = (const char*)blk->pin;
    if (rows == 0) { memcpy(host, m, bytes); return; }
    for (size_t f = 0; f < frames; f++)               // pitched: the bytes between the rows stay as the caller had them
        for (size_t r = 0; r < rows; r++)
            memcpy((char*)host + f * frame_pitch + r * pitch, m + f * frame_pitch + r * pitch, row_bytes);
}
int finish_outputs(int mem, hipStream_t s, std::initializer_list<Staged*> outs) {
    for (Staged* o : outs) if (int r = o->finish(s)) return r;
    if (mem != VS_MEM_HOST) return VS_OK;
    VS_HIP(hipStreamSynchronize(s));
    host_synced();
    for (Staged* o : outs) o->complete();
    return VS_OK;
}

}  // namespace vsi

using vsi::Staged;
using vsi::set_error;

#define VS_TRY(expr) do { int _r = (expr); if (_r != VS_OK) return _r; } while (0)
#define VS_ARG(cond) do { if (!(cond)) return set_error(VS_ERR_ARG, "bad argument: %s (%s)", #cond, __func__); } while (0)
// every image extent an entry point takes: positive and at most 65535 a side (tile coordinates are 16-bit, row offsets are 24-bit multiplies, and
// the products w * channels / roi.x + roi.w of the checks that follow stay inside int)
#define VS_DIMS(w, h) VS_ARG((w) > 0 && (h) > 0 && (w) <= 65535 && (h) <= 65535)
#define VS_TRY_RING(expr) do { int _r = (expr); if (_r != VS_OK) return _r; } while (0)


// Pinned host ring + device mirror for small parameter blocks.  upload() copies `n` float4 into the next free span of
// the pinned ring, enqueues H2D into the same span of the device mirror on `s`, and returns the device pointer.  A span
// is tracked as busy from the moment its copy is enqueued (an event is recorded right behind the copy) and fence() moves
// that event behind the consuming kernel; it is reused only after its event has completed, so the source of an in-flight
// copy is never overwritten -- also when the call fails between upload() and fence().
// One ring per (running thread, device): calls from different threads or for different devices never wait on each other and
// need no lock on the call path (the header promises that distinct handles are independent).  Events are pooled.
namespace {
// span bookkeeping of a ring of `slots` units: reserve() hands out the next free span (waiting for every in-flight span it overlaps),
// commit() marks it busy behind an event on the stream, fence() moves that event behind the consuming kernel
struct SpanRing {
    size_t slots = 0;
    size_t head = 0;
    struct Busy { size_t begin, end; hipEvent_t ev; hipStream_t stream; };
    std::deque<Busy> busy;
    std::mutex mu;                               // the owning thread (reserve / commit / fence) against vsi::retire_stream from any thread
    std::vector<hipEvent_t> pool;
    int take_event(hipEvent_t* ev) {
        if (!pool.empty()) { *ev = pool.back(); pool.pop_back(); return VS_OK; }
        VS_HIP(hipEventCreateWithFlags(ev, hipEventDisableTiming));
        return VS_OK;
    }
    int retire(std::deque<Busy>::iterator it) {
        VS_HIP(hipEventSynchronize(it->ev));
        pool.push_back(it->ev);
        busy.erase(it);
        return VS_OK;
    }
    // (mu held) the next span of n units: [*b, *b + n), every in-flight span that overlaps it waited for and retired
    int reserve(size_t n, size_t* b) {
        if (head + n > slots) head = 0;
        const size_t lo = head, hi = head + n;
        for (bool again = true; again;) {
            again = false;
            for (auto it = busy.begin(); it != busy.end(); ++it)
                if (it->begin < hi && lo < it->end) { VS_TRY_RING(retire(it)); again = true; break; }
        }
        while (busy.size() > 64) VS_TRY_RING(retire(busy.begin()));   // keep the list short: retire the oldest
        *b = lo;
        return VS_OK;
    }
    // (mu held) the span is busy from now on: an event recorded on `s` right behind what the caller has just enqueued
    int commit(size_t b, size_t n, hipStream_t s, hipError_t enqueue_result) {
        Busy bz{b, b + n, nullptr, s};
        VS_TRY_RING(take_event(&bz.ev));
        hipError_t err = enqueue_result;
        if (err == hipSuccess) err = hipEventRecord(bz.ev, s);
        if (err != hipSuccess) { pool.push_back(bz.ev); VS_HIP(err); }
        busy.push_back(bz);
        head = b + n;
        return VS_OK;
    }
    // called after the consuming kernel has been enqueued on `s`: the span that begins at b stays busy until that kernel is done
    int fence_span(size_t b, hipStream_t s) {
        std::lock_guard<std::mutex> g(mu);
        for (auto& z : busy)
            if (z.begin == b) { VS_HIP(hipEventRecord(z.ev, s)); z.stream = s; return VS_OK; }
        return VS_OK;
    }
    // every span whose event was recorded on `s` is waited for and retired NOW, while the stream still exists: this runtime's
    // hipEventSynchronize looks at the stream an event was last recorded on, so an event must never outlive that stream in here
    hipError_t forget_stream(hipStream_t s) {
        std::lock_guard<std::mutex> g(mu);
        hipError_t first = hipSuccess;
        for (auto it = busy.begin(); it != busy.end();) {
            if (it->stream == s) {
                const hipError_t e = hipEventSynchronize(it->ev);
                if (e != hipSuccess && first == hipSuccess) first = e;
                pool.push_back(it->ev); it = busy.erase(it);
            }
            else ++it;
        }
        return first;
    }
};
struct ParamRing : SpanRing {
    static constexpr size_t kSlots = 1 << 15;    // 32768 float4 = 512 KiB
    float4* host = nullptr;
    float4* dev = nullptr;
    ParamRing() { slots = kSlots; }
    int upload(const float* src, size_t n, hipStream_t s, float4** out) {
        std::lock_guard<std::mutex> g(mu);
        if (n == 0 || n > kSlots / 2) return vsi::set_error(VS_ERR_ARG, "parameter block of %zu frames is too large", n);
        if (!host) VS_HIP(vsi::pinned_alloc((void**)&host, kSlots * sizeof(float4)));
        if (!dev) VS_HIP(vsi::dev_alloc((void**)&dev, kSlots * sizeof(float4)));
        size_t b = 0;
        VS_TRY_RING(reserve(n, &b));
        memcpy(host + b, src, n * sizeof(float4));
        VS_TRY_RING(commit(b, n, s, hipMemcpyAsync(dev + b, host + b, n * sizeof(float4), hipMemcpyHostToDevice, s)));
        *out = dev + b;
        return VS_OK;
    }
    // a span of the DEVICE half only (the caller fills it with a kernel: see param_block_to_device)
    int take(size_t n, hipStream_t s, float4** out) {
        std::lock_guard<std::mutex> g(mu);
        if (n == 0 || n > kSlots / 2) return vsi::set_error(VS_ERR_ARG, "parameter block of %zu frames is too large", n);
        if (!dev) VS_HIP(vsi::dev_alloc((void**)&dev, kSlots * sizeof(float4)));
        size_t b = 0;
        VS_TRY_RING(reserve(n, &b));
        VS_TRY_RING(commit(b, n, s, hipSuccess));
        *out = dev + b;
        return VS_OK;
    }
    int fence(float4* p, hipStream_t s) { return fence_span((size_t)(p - dev), s); }
};
// Small parameter blocks (up to kParamBlockSlots float4: the per-frame parameters + extents of up to 64 frames) reach the device as KERNEL ARGUMENTS of a
// one-workgroup kernel that writes them into the ring's span: no host-to-device copy stands between the host's numbers and the warp kernel for the
// per-frame drop-in call and the small batches of the parity tests (round 6, profiles/r06_flake.md: the matrix upload was the second of the two
// runtime-ordered transfers the round-5 failure could have come from).
constexpr int kParamBlockSlots = 128;
struct ParamBlock { float4 v[kParamBlockSlots]; };
__global__ __launch_bounds__(kParamBlockSlots) void vs_k_param_block(ParamBlock b, float4* __restrict__ dst, int n) {
    const int i = (int)threadIdx.x;
    if (i < n) dst[i] = b.v[i];
}
// Device-only scratch for VS_WARP_BILINEAR_CV's per-frame coordinate tables (vs_k_cv_tables writes them, the warp kernel of the same
// call reads them): 48 KiB per 4K frame.  take() hands out n ints; the caller enqueues the table kernel and the warp kernel, then fence()s.
struct TableRing : SpanRing {
    static constexpr size_t kInts = (size_t)8 << 20;     // 32 MiB: ~680 4K frames; a call takes at most half per launch group
    int* dev = nullptr;
    TableRing() { slots = kInts; }
    int take(size_t n, hipStream_t s, int** out) {
        std::lock_guard<std::mutex> g(mu);
        if (n == 0 || n > kInts / 2) return vsi::set_error(VS_ERR_ARG, "coordinate tables of %zu ints are too large", n);
        if (!dev) VS_HIP(vsi::dev_alloc((void**)&dev, kInts * sizeof(int)));
        size_t b = 0;
        VS_TRY_RING(reserve(n, &b));
        VS_TRY_RING(commit(b, n, s, hipSuccess));
        *out = dev + b;
        return VS_OK;
    }
    int fence(int* p, hipStream_t s) { return fence_span((size_t)(p - dev), s); }
};
// A thread takes a ring pair per device on first use and hands it back to a process-wide pool when it exits; the next thread that
// needs one for that device reuses it (its in-flight spans are retired by event as always).  So the pinned / device memory
// held is bounded by the largest number of threads that were inside bgr_image_warp entry points at the same time, not by the
// number of threads that ever called one (a thread-per-frame caller used to leave 1 MiB behind per thread).  Nothing is freed
// at thread or process exit: the HIP runtime may already be gone by then.
struct Rings { ParamRing params; TableRing tables; };
struct RingPool {
    std::mutex mu;
    std::vector<Rings*> free_[16];
    std::vector<Rings*> all;                     // every ring pair ever made (rings are never destroyed)
};
RingPool& ring_pool() { static RingPool* p = new RingPool(); return *p; }      // never destroyed: outlives every thread's holder
struct RingHolder {
    Rings* r[16] = {};
    ~RingHolder() {
        RingPool& p = ring_pool();
        std::lock_guard<std::mutex> g(p.mu);
        for (int d = 0; d < 16; d++) if (r[d]) p.free_[d].push_back(r[d]);
    }
};
Rings* thread_rings() {
    thread_local RingHolder holder;
    int device = 0;
    if (hipGetDevice(&device) != hipSuccess || device < 0 || device >= 16) return nullptr;
    if (!holder.r[device]) {
        RingPool& p = ring_pool();
        std::lock_guard<std::mutex> g(p.mu);
        if (!p.free_[device].empty()) { holder.r[device] = p.free_[device].back(); p.free_[device].pop_back(); }
        else { holder.r[device] = new Rings(); p.all.push_back(holder.r[device]); }
    }
    return holder.r[device];
}
ParamRing* param_ring() { Rings* r = thread_rings(); return r ? &r->params : nullptr; }
TableRing* table_ring() { Rings* r = thread_rings(); return r ? &r->tables : nullptr; }
}  // namespace

// A stream is about to be destroyed (a handle's own stream: ~vs_aligner; a caller's stream: vs_stream_retire): nothing in the
// library may refer to it afterwards.
bool& vsi::warp_keeps_solver_slot() { thread_local bool v = false; return v; }

hipError_t vsi::retire_stream(hipStream_t s) {
    std::vector<Rings*> rings;
    {
        RingPool& p = ring_pool();
        std::lock_guard<std::mutex> g(p.mu);
        rings = p.all;
    }
    hipError_t first = hipSuccess;
    for (Rings* r : rings) {
        const hipError_t e1 = r->params.forget_stream(s), e2 = r->tables.forget_stream(s);
        if (e1 != hipSuccess && first == hipSuccess) first = e1;
        if (e2 != hipSuccess && first == hipSuccess) first = e2;
    }
    return first;
}

static inline size_t img_span(int w, int h, int stride, int channels) {
    return (size_t)(h - 1) * stride + (size_t)w * channels;
}

extern "C" {

int vs_stream_retire(void* stream) try {
    VS_HIP(vsi::retire_stream((hipStream_t)stream));      // a fault of the work that was in flight on the stream surfaces here
    return VS_OK;
} VS_CATCH_ALL

// ---- the debug build's bounds record (vs_device.hpp, -DVS_DEBUG_BOUNDS) ---------------------------------------------------
#ifdef VS_DEBUG_BOUNDS
}  // extern "C"
#include "vs_device.hpp"
extern "C" int vs_bounds_fetch_engine(unsigned out[8], int reset);
extern "C" int vs_bounds_fetch_warp(unsigned out[8], int reset);
extern "C" int vs_bounds_fetch_phase(unsigned out[8], int reset);
namespace {
// the checker checked: element 11 of an 8-element LDS array through a Span -- reported under site 900, executed on element 0
__global__ void vs_k_bounds_selftest(int* out) {
    __shared__ int arr[8];
    if (threadIdx.x < 8) arr[threadIdx.x] = 100 + (int)threadIdx.x;
    __syncthreads();
    VS_SPAN(int*, a, arr, 8, 900);
    if (threadIdx.x == 3) out[0] = a[11] + a[2];          // 100 (element 0 stands in) + 102
}
}  // namespace
VS_BOUNDS_TU(vs_bounds_fetch_capi)
extern "C" {
#endif

int vs_debug_bounds_check(void) try {
#ifdef VS_DEBUG_BOUNDS
    if (!vsi::device_ready()) return VS_ERR_HIP;
    VS_HIP(hipDeviceSynchronize());
    int (*const fetch[])(unsigned*, int) = {vs_bounds_fetch_engine, vs_bounds_fetch_warp, vs_bounds_fetch_phase, vs_bounds_fetch_capi};
    const char* const names[] = {"vs_engine.hip", "vs_warp.hip", "vs_phase.hip", "vs_capi.hip"};
    unsigned total = 0;
    char msg[512] = "";
    for (int k = 0; k < 4; k++) {
        unsigned r[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        if (fetch[k](r, 1) != 0) return set_error(VS_ERR_HIP, "bounds record of %s is not readable", names[k]);
        if (r[0] && !total)
            snprintf(msg, sizeof msg, "%s: site %u, index %d, limit %u, workgroup %u, thread %u", names[k], r[1], (int)r[2], r[3], r[4], r[5]);
        total += r[0];
    }
    if (total) set_error(VS_ERR_STATE, "%u out-of-bounds index(es) since the last check; first: %s", total, msg);
    return (int)std::min<unsigned>(total, 0x7fffffffu);
#else
    return set_error(VS_ERR_UNSUPPORTED, "vs_debug_bounds_check: this is not a -DVS_DEBUG_BOUNDS build of libvs_amd (tools/build_variant.sh bounds)");
#endif
} VS_CATCH_ALL

int vs_debug_bounds_selftest(void) try {
#ifdef VS_DEBUG_BOUNDS
    if (!vsi::device_ready()) return VS_ERR_HIP;
    vsi::DevBuf out;
    VS_HIP(out.alloc(sizeof(int)));
    hipLaunchKernelGGL(vs_k_bounds_selftest, dim3(1), dim3(64), 0, nullptr, out.as<int>());
    VS_HIP(hipGetLastError());
    int v = 0;
    VS_HIP(hipMemcpy(&v, out.p, sizeof(int), hipMemcpyDeviceToHost));
    return v;                                  // 202 when the violating access was redirected to element 0
#else
    return set_error(VS_ERR_UNSUPPORTED, "vs_debug_bounds_selftest: this is not a -DVS_DEBUG_BOUNDS build of libvs_amd");
#endif
} VS_CATCH_ALL

int vs_test_fail_alloc(int k) try {
    const long long seen = vsi::g_alloc_count.exchange(0);
    vsi::g_alloc_fail_at.store(k);                         // k > 0: the k-th allocation reports out-of-memory; k < 0: it throws std::bad_alloc; 0: disarmed
    return (int)std::min<long long>(seen, 0x7fffffff);
} VS_CATCH_ALL

int vs_device_count(void) try {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
} VS_CATCH_ALL

int vs_calib_copy12(const void* src_dev, void* dst_dev, size_t bytes, void* stream) try {
    VS_ARG(src_dev && dst_dev && bytes >= 12);
    if (!vsi::device_ready()) return VS_ERR_HIP;
    VS_HIP(vsk::calib_copy12(src_dev, dst_dev, bytes, (hipStream_t)stream));
    return VS_OK;
} VS_CATCH_ALL

// the shader clock from inside a kernel: s_memtime counts shader cycles, s_memrealtime a constant 100 MHz
__global__ __launch_bounds__(256) void vs_k_clock_probe(unsigned long long* out, float seed, int iters) {
    float a0 = seed + (float)threadIdx.x, a1 = a0 + 1.0f, a2 = a0 + 2.0f, a3 = a0 + 3.0f;
    const float m = 0.999999f, c = 1.0e-7f;
    __builtin_amdgcn_sched_barrier(0);
    const unsigned long long t0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
    __builtin_amdgcn_sched_barrier(0);
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int k = 0; k < 8; k++) {
            a0 = __builtin_fmaf(a0, m, c); a1 = __builtin_fmaf(a1, m, c); a2 = __builtin_fmaf(a2, m, c); a3 = __builtin_fmaf(a3, m, c);
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    const unsigned long long t1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime();
    if (blockIdx.x == 0 && threadIdx.x == 0) { out[0] = t1 - t0; out[1] = r1 - r0; }
    if (a0 + a1 + a2 + a3 == 12345.678f) out[2] = 1;                  // keeps the chains alive
}

int vs_shader_clock_probe(void* stream, double* shader_mhz) try {
    VS_ARG(shader_mhz);
    if (!vsi::device_ready()) return VS_ERR_HIP;
    hipStream_t s = (hipStream_t)stream;
    unsigned long long* d = nullptr;
    VS_HIP(vsi::dev_alloc((void**)&d, 4 * sizeof(unsigned long long)));
    unsigned long long h[2] = {0, 0};
    hipError_t e = hipMemsetAsync(d, 0, 4 * sizeof(unsigned long long), s);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(vs_k_clock_probe, dim3(2048), dim3(256), 0, s, d, 1.0f, 3000);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(h, d, sizeof(h), hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    (void)hipFree(d);
    VS_HIP(e);
    *shader_mhz = h[1] ? 100.0 * (double)h[0] / (double)h[1] : 0.0;
    return VS_OK;
} VS_CATCH_ALL

int vs_pyr_down(const uint8_t* in, int w, int h, int in_stride, uint8_t* out, int ow, int oh, int out_stride, int mem,
                void* stream) try {
    VS_DIMS(w, h); VS_DIMS(ow, oh);
    VS_ARG(in && out && w > 0 && h > 0 && ow > 0 && oh > 0 && in_stride >= w && out_stride >= ow);
    VS_ARG(2 * ow <= w + 1 && 2 * oh <= h + 1);
    if (!vsi::device_ready()) return VS_ERR_HIP;
    hipStream_t s = (hipStream_t)stream;
    Staged a, b;
    VS_TRY(a.in(in, img_span(w, h, in_stride, 1), mem, s));
    VS_TRY(b.out_image(out, (size_t)ow, (size_t)oh, (size_t)out_stride, 1, 0, mem));
    VS_HIP(vsk::pyr_down(a.as<uint8_t>(), w, h, in_stride, b.as<uint8_t>(), ow, oh, out_stride, 1, 0, 0, s));
    return vsi::finish_outputs(mem, s, {&b});
} VS_CATCH_ALL

int vs_optimal_dft_size(int n) { return vsp::optimal_dft_size(n); }

int vs_phase_correlate(const uint8_t* a, const uint8_t* b, int w, int h, int stride, int mem, void* stream, float* surface,
                       double* result) try {
    VS_DIMS(w, h);
    VS_ARG(a && b && result && w > 0 && h > 0 && stride >= w);
    if (!vsi::device_ready()) return VS_ERR_HIP;
    hipStream_t s = (hipStream_t)stream;
    vsp::Context ctx;
    struct Guard { vsp::Context& c; ~Guard() { c.destroy(); } } guard{ctx};
    if (ctx.configure(w, h, s) != hipSuccess)
        return set_error(VS_ERR_UNSUPPORTED, "phase correlation: padded extent over %d (%dx%d image)", vsp::kMaxLine, w, h);
    Staged ia, ib, surf;
    VS_TRY(ia.in(a, img_span(w, h, stride, 1), mem, s));
    VS_TRY(ib.in(b, img_span(w, h, stride, 1), mem, s));
    vsi::DevBuf spec, G, surf_own, pairs, res;
    VS_HIP(spec.alloc(2 * ctx.spec_frame() * sizeof(float2)));
    VS_HIP(G.alloc(ctx.spec_frame() * sizeof(float2)));
    VS_HIP(pairs.alloc(sizeof(vsp::Pair)));
    VS_HIP(res.alloc(sizeof(vsp::Result)));
    float* d_surf;
    if (surface) { VS_TRY(surf.out(surface, ctx.surface_elems() * sizeof(float), mem)); d_surf = surf.as<float>(); }
    else { VS_HIP(surf_own.alloc(ctx.surface_elems() * sizeof(float))); d_surf = surf_own.as<float>(); }
    const vsp::Pair pr{0, 1};
    VS_HIP(hipMemcpyAsync(pairs.p, &pr, sizeof pr, hipMemcpyHostToDevice, s));
    VS_HIP(ctx.spectra(ia.as<uint8_t>(), 0, stride, 1, spec.as<float2>(), s));
    VS_HIP(ctx.spectra(ib.as<uint8_t>(), 0, stride, 1, spec.as<float2>() + ctx.spec_frame(), s));
    VS_HIP(ctx.correlate(spec.as<float2>(), pairs.as<vsp::Pair>(), 1, G.as<float2>(), d_surf, res.as<vsp::Result>(), s));
    if (surface) VS_TRY(surf.finish(s));
    vsp::Result r;
    VS_HIP(hipMemcpyAsync(&r, res.p, sizeof r, hipMemcpyDeviceToHost, s));
    VS_HIP(hipStreamSynchronize(s));        // the result is a host value: this call always synchronises
    vsi::host_synced();
    surf.complete();
    result[0] = r.dx; result[1] = r.dy; result[2] = r.response;
    return VS_OK;
} VS_CATCH_ALL

int vs_grad_xy(const uint8_t* in, int w, int h, int stride, float* gx, float* gy, int mem, void* stream) try {
    VS_DIMS(w, h);
    VS_ARG(in && gx && gy && w > 0 && h > 0 && stride >= w);
    if (!vsi::device_ready()) return VS_ERR_HIP;
    hipStream_t s = (hipStream_t)stream;
    Staged a, x, y;
    VS_TRY(a.in(in, img_span(w, h, stride, 1), mem, s));
    VS_TRY(x.out(gx, (size_t)w * h * 4, mem));
    VS_TRY(y.out(gy, (size_t)w * h * 4, mem));
    VS_HIP(vsk::grad_xy(a.as<uint8_t>(), w, h, stride, x.as<float>(), y.as<float>(), s));
    return vsi::finish_outputs(mem, s, {&x, &y});
} VS_CATCH_ALL

int vs_grad_argmax(const float* gx, const float* gy, int w, int h, int ts, uint16_t* lmx, uint16_t* lmy, int mem,
                   void* stream) try {
    VS_DIMS(w, h);
    VS_ARG(gx && gy && lmx && lmy && w > 0 && h > 0 && ts >= 1 && ts <= 64);
    if (!vsi::device_ready()) return VS_ERR_HIP;
    hipStream_t s = (hipStream_t)stream;
    const size_t nt = (size_t)(w / ts) * (h / ts);
    Staged a, b, x, y;
    VS_TRY(a.in(gx, (size_t)w * h * 4, mem, s));
    VS_TRY(b.in(gy, (size_t)w * h * 4, mem, s));
    VS_TRY(x.out(lmx, nt * 2 * 2, mem));
    VS_TRY(y.out(lmy, nt * 2 * 2, mem));
    VS_HIP(vsk::grad_argmax(a.as<float>(), b.as<float>(), w, h, ts, x.as<uint16_t>(), y.as<uint16_t>(), s));
    return vsi::finish_outputs(mem, s, {&x, &y});
} VS_CATCH_ALL

int vs_sparse_jac(const float* gx, const float* gy, int w, int h, const uint16_t* lmx, const uint16_t* lmy, int tx, int ty,
                  float* out_x, float* out_y, int mem, void* stream) try {
    VS_DIMS(w, h);
    VS_ARG(gx && gy && lmx && lmy && out_x && out_y && w > 0 && h > 0 && tx > 0 && ty > 0);
    if (!vsi::device_ready()) return VS_ERR_HIP;
    hipStream_t s = (hipStream_t)stream;
    const size_t nt = (size_t)tx * ty;
    Staged a, b, c, d, x, y;
    VS_TRY(a.in(gx, (size_t)w * h * 4, mem, s));
    VS_TRY(b.in(gy, (size_t)w * h * 4, mem, s));
    VS_TRY(c.in(lmx, nt * 4, mem, s));
    VS_TRY(d.in(lmy, nt * 4, mem, s));
    VS_TRY(x.out(out_x, nt * 16, mem));
    VS_TRY(y.out(out_y, nt * 16, mem));
    VS_HIP(vsk::sparse_jac(a.as<float>(), b.as<float>(), w, h, c.as<uint16_t>(), d.as<uint16_t>(), (int)nt, x.as<float>(),
                           y.as<float>(), s));
    return vsi::finish_outputs(mem, s, {&x, &y});
} VS_CATCH_ALL

int vs_keyframe_fused(const uint8_t* in, int w, int h, int stride, int ts, uint16_t* lmx, uint16_t* lmy, float* jx,
                      float* jy, int mem, void* stream) try {
    VS_ARG(in && lmx && lmy && jx && jy && w > 0 && h > 0 && stride >= w && ts >= 1 && ts <= 64);
    VS_ARG(w <= 65535 && h <= 65535);
    if (!vsi::device_ready()) return VS_ERR_HIP;
    hipStream_t s = (hipStream_t)stream;
    const size_t nt = (size_t)(w / ts) * (h / ts);
    Staged a, x, y, p, q;
    VS_TRY(a.in(in, img_span(w, h, stride, 1), mem, s));
    VS_TRY(x.out(lmx, nt * 4, mem));
    VS_TRY(y.out(lmy, nt * 4, mem));
    VS_TRY(p.out(jx, nt * 16, mem));
    VS_TRY(q.out(jy, nt * 16, mem));
    VS_HIP(vsk::keyframe(a.as<uint8_t>(), w, h, stride, ts, x.as<uint16_t>(), y.as<uint16_t>(), p.as<float>(),
                         q.as<float>(), 1, 0, 0, 0, s));
    return vsi::finish_outputs(mem, s, {&x, &y, &p, &q});
} VS_CATCH_ALL

int vs_sparse_warpdiff(const uint8_t* tmpl, const uint8_t* key, int w, int h, int stride, const uint16_t* lm, int tx,
                       int ty, float A, float B, float TX, float TY, uint16_t* out, int mem, void* stream) try {
    VS_DIMS(w, h);
    VS_ARG(tmpl && key && lm && out && w > 0 && h > 0 && stride >= w && tx > 0 && ty > 0);
    if (!vsi::device_ready()) return VS_ERR_HIP;
    hipStream_t s = (hipStream_t)stream;
    const size_t nt = (size_t)tx * ty;
    Staged a, b, c, o;
    VS_TRY(a.in(tmpl, img_span(w, h, stride, 1), mem, s));
    VS_TRY(b.in(key, img_span(w, h, stride, 1), mem, s));
    VS_TRY(c.in(lm, nt * 4, mem, s));
    VS_TRY(o.out(out, nt * 2, mem));
    VS_HIP(vsk::sparse_warpdiff(a.as<uint8_t>(), b.as<uint8_t>(), w, h, stride, c.as<uint16_t>(), (int)nt, A, B, TX, TY,
                                o.as<uint16_t>(), s));
    return vsi::finish_outputs(mem, s, {&o});
} VS_CATCH_ALL

int vs_sparse_ica(const uint8_t* tmpl, const uint8_t* key, int w, int h, int stride, const uint16_t* selx, int nx,
                  const uint16_t* sely, int ny, const float* jacx, const float* jacy, float A, float B, float TX, float TY,
                  double* out4, int mem, void* stream) try {
    VS_DIMS(w, h);
    VS_ARG(tmpl && key && out4 && w > 0 && h > 0 && stride >= w && nx >= 0 && ny >= 0);
    VS_ARG((nx == 0 || (selx && jacx)) && (ny == 0 || (sely && jacy)));
    if (!vsi::device_ready()) return VS_ERR_HIP;
    hipStream_t s = (hipStream_t)stream;
    Staged a, b, sx, sy, jx, jy, o;
    VS_TRY(a.in(tmpl, img_span(w, h, stride, 1), mem, s));
    VS_TRY(b.in(key, img_span(w, h, stride, 1), mem, s));
    VS_TRY(sx.in(selx, (size_t)nx * 4, mem, s));
    VS_TRY(sy.in(sely, (size_t)ny * 4, mem, s));
    VS_TRY(jx.in(jacx, (size_t)nx * 16, mem, s));
    VS_TRY(jy.in(jacy, (size_t)ny * 16, mem, s));
    VS_TRY(o.out(out4, 32, mem));
    VS_HIP(vsk::sparse_ica(a.as<uint8_t>(), b.as<uint8_t>(), w, h, stride, sx.as<uint16_t>(), nx, sy.as<uint16_t>(), ny,
                           jx.as<float>(), jy.as<float>(), A, B, TX, TY, o.as<double>(), s));
    return vsi::finish_outputs(mem, s, {&o});
} VS_CATCH_ALL

int vs_image_warp(const uint8_t* in, int w, int h, int stride, float A, float B, float TX, float TY, float* out, int ow,
                  int oh, int mem, void* stream) try {
    VS_DIMS(w, h); VS_DIMS(ow, oh);
    VS_ARG(in && out && w > 0 && h > 0 && stride >= w && ow > 0 && oh > 0);
    if (!vsi::device_ready()) return VS_ERR_HIP;
    hipStream_t s = (hipStream_t)stream;
    Staged a, o;
    VS_TRY(a.in(in, img_span(w, h, stride, 1), mem, s));
    VS_TRY(o.out(out, (size_t)ow * oh * 4, mem));
    VS_HIP(vsk::image_warp(a.as<uint8_t>(), w, h, stride, A, B, TX, TY, o.as<float>(), ow, oh, s));
    return vsi::finish_outputs(mem, s, {&o});
} VS_CATCH_ALL

// roi = NULL: the whole w x h output.  Otherwise dst holds only the window (roi->w x roi->h pixels per frame).
static int bgr_warp_common(const void* src, size_t src_fs, int n_frames, int w, int h, int src_stride, int channels,
                           int bits, const vs_transform* t, int mode, int border, int max_value, void* dst,
                           size_t dst_fs, int dst_stride, bool f32out, int mem, hipStream_t s, const vsk::Roi* roi_in = nullptr) {
    VS_DIMS(w, h);
    VS_ARG(src && dst && t && n_frames >= 1 && w > 0 && h > 0 && channels >= 1 && channels <= 4);
    VS_ARG(bits == 8 || bits == 16);
    const vsk::Roi roi = roi_in ? *roi_in : vsk::Roi{0, 0, w, h};
    VS_ARG(roi.x >= 0 && roi.y >= 0 && roi.w >= 1 && roi.h >= 1 && roi.w <= w && roi.h <= h && roi.x <= w - roi.w && roi.y <= h - roi.h);   // (no sum that could overflow)
    VS_ARG(src_stride >= w * channels && dst_stride >= roi.w * channels);
    VS_ARG(mode == VS_WARP_LANCZOS2 || mode == VS_WARP_BILINEAR || mode == VS_WARP_LANCZOS2_FAST || mode == VS_WARP_LANCZOS2_SEP || mode == VS_WARP_BILINEAR_CV);
    VS_ARG(border == VS_BORDER_CLAMP || border == VS_BORDER_CONSTANT);
    if (mode == VS_WARP_BILINEAR_CV && f32out) return set_error(VS_ERR_UNSUPPORTED, "VS_WARP_BILINEAR_CV has integer outputs only (cv::warpAffine on 8- / 16-bit frames)");
    VS_ARG(n_frames == 1 || (src_fs >= img_span(w, h, src_stride, channels) && dst_fs >= img_span(roi.w, roi.h, dst_stride, channels)));
    if (!vsi::device_ready()) return VS_ERR_HIP;
    const size_t esz = bits / 8, osz = f32out ? 4 : esz;
    if (mode == VS_WARP_BILINEAR_CV) {
        // cv::warpAffine's own arithmetic: t[i] is the FORWARD transform; its inverted matrix (six doubles = three ring slots per frame)
        // is what the kernels take
        VS_ARG(max_value >= 0 && max_value <= (bits == 8 ? 255 : 65535));
        ParamRing* ring = param_ring();
        if (!ring) return set_error(VS_ERR_UNSUPPORTED, "no current HIP device with index < 16");
        Staged a, o;
        const size_t in_bytes = ((size_t)(n_frames - 1) * src_fs + img_span(w, h, src_stride, channels)) * esz;
        VS_TRY(a.in(src, in_bytes, mem, s));
        VS_TRY(o.out_image(dst, (size_t)roi.w * channels * esz, (size_t)roi.h, (size_t)dst_stride * esz, (size_t)n_frames, dst_fs * esz, mem));
        // frames per launch group: their matrices fit one upload of the parameter ring (half its slots per call) and -- tuned kernel -- their
        // coordinate tables half of the table ring
        TableRing* tring = channels == 3 ? table_ring() : nullptr;
        const size_t tab_per = channels == 3 ? vsk::bgr_warp_cv_table_ints(bits, roi) : 0;
        int per_call = (int)(ParamRing::kSlots / 2 / 3);
        if (tab_per) per_call = (int)std::min<size_t>((size_t)per_call, std::max<size_t>(1, TableRing::kInts / 2 / tab_per));
        std::vector<double> Mv((size_t)std::min(n_frames, per_call) * 6 + 2);   // (+2: an upload is counted in 16-byte slots)
        for (int f0 = 0; f0 < n_frames; f0 += per_call) {
            const int nf = std::min(per_call, n_frames - f0);
            for (int i = 0; i < nf; i++) vs_cv_inverse_matrix(&t[f0 + i], w, h, &Mv[(size_t)i * 6]);
            const char* sp = (const char*)a.dev + (size_t)f0 * src_fs * esz;
            char* dp = (char*)o.dev + (size_t)f0 * dst_fs * esz;
            hipError_t e = hipErrorNotSupported;
            int* tdev = nullptr;
            float4* mdev = nullptr;
            // small calls (one frame of the drop-in pattern, the parity tests' batches): the matrices go to the table kernel as kernel arguments, no upload
            const bool by_value = n_frames <= vsk::kCvInlineFrames;
            if (channels == 3 && tring && tab_per * (size_t)nf <= TableRing::kInts / 2) {
                if (!by_value) VS_TRY(ring->upload((const float*)Mv.data(), ((size_t)nf * 6 * sizeof(double) + 15) / 16, s, &mdev));
                VS_TRY(tring->take(tab_per * (size_t)nf, s, &tdev));
                e = vsk::bgr_warp_cv_c3(sp, w, h, src_stride, bits, (const double*)mdev, by_value ? Mv.data() : nullptr, tdev, border, max_value, dp, dst_stride, nf, src_fs,
                                        dst_fs, roi, s);
            }
            if (e == hipErrorNotSupported) {
                if (!mdev) VS_TRY(ring->upload((const float*)Mv.data(), ((size_t)nf * 6 * sizeof(double) + 15) / 16, s, &mdev));
                e = vsk::bgr_warp_cv_generic(sp, w, h, src_stride, channels, bits, (const double*)mdev, border, max_value, dp, dst_stride, nf, src_fs, dst_fs, roi, s);
            }
            VS_HIP(e);
            if (mdev) VS_TRY(ring->fence(mdev, s));
            if (tdev) VS_TRY(tring->fence(tdev, s));
        }
        return vsi::finish_outputs(mem, s, {&o});
    }
    // kernel parameters of every frame, then (tuned 3-channel kernel) the extents its tile prologue uses: one upload of 2n float4
    static const bool host_extents = []() { const char* e = getenv("VS_WARP_HOST_EXTENTS"); return e ? atoi(e) != 0 : true; }();
    const bool tuned = channels == 3 && !f32out && max_value >= 0 && max_value <= (bits == 8 ? 255 : 65535);
    const bool with_extents = tuned && host_extents && (size_t)n_frames * 2 <= ParamRing::kSlots / 2;   // (the ring takes half its slots per call)
    std::vector<float> P((size_t)n_frames * (with_extents ? 8 : 4));
    for (int i = 0; i < n_frames; i++) vs_ul_params_warp(&t[i], w, h, &P[(size_t)i * 4]);
    if (with_extents) vsk::bgr_warp_c3_extents(P.data(), n_frames, roi, bits, mode, P.data() + (size_t)n_frames * 4);
    static const bool slot_hint = []() { const char* e = getenv("VS_WARP_SLOT_HINT"); return e ? atoi(e) != 0 : true; }();     // (A/B switch: 0 = the engine's overlapped warps may go compact too)
    const int compact = with_extents && !(slot_hint && vsi::warp_keeps_solver_slot()) ? vsk::bgr_warp_c3_compact_shape(P.data() + (size_t)n_frames * 4, n_frames) : 0;
    // Per-frame kernel parameters travel host -> device through a pinned ring (ParamRing below), so a
    // VS_MEM_DEVICE call stays asynchronous and never reads a host buffer that has gone out of scope.
    float4* pdev = nullptr;
    ParamRing* ring = param_ring();
    if (!ring) return set_error(VS_ERR_UNSUPPORTED, "no current HIP device with index < 16");
    if (P.size() / 4 <= (size_t)kParamBlockSlots) {          // small calls: by value through a one-workgroup kernel (see vs_k_param_block)
        ParamBlock blk{};
        memcpy(blk.v, P.data(), P.size() * sizeof(float));
        VS_TRY(ring->take(P.size() / 4, s, &pdev));
        hipLaunchKernelGGL(vs_k_param_block, dim3(1), dim3(kParamBlockSlots), 0, s, blk, pdev, (int)(P.size() / 4));
        VS_HIP(hipGetLastError());
    } else
        VS_TRY(ring->upload(P.data(), P.size() / 4, s, &pdev));
    Staged a, o;
    const size_t in_bytes = ((size_t)(n_frames - 1) * src_fs + img_span(w, h, src_stride, channels)) * esz;
    VS_TRY(a.in(src, in_bytes, mem, s));
    VS_TRY(o.out_image(dst, (size_t)roi.w * channels * osz, (size_t)roi.h, (size_t)dst_stride * osz, (size_t)n_frames, dst_fs * osz, mem));
    hipError_t e = hipErrorNotSupported;
    if (tuned)
        e = vsk::bgr_warp_c3(a.dev, w, h, src_stride, bits, pdev, with_extents ? pdev + n_frames : nullptr, mode, border, max_value, o.dev,
                             dst_stride, n_frames, src_fs, dst_fs, roi, compact, s);
    if (e == hipErrorNotSupported)   // layouts without a tuned kernel (other channel counts, float output): one thread per pixel, same arithmetic
        e = vsk::bgr_warp_generic(a.dev, w, h, src_stride, channels, bits, pdev, mode, border, max_value,
                                  o.dev, dst_stride, f32out, n_frames, src_fs, dst_fs, roi, s);
    VS_HIP(e);
    VS_TRY(ring->fence(pdev, s));
    return vsi::finish_outputs(mem, s, {&o});
}

int vs_bgr_image_warp(const void* src, int w, int h, int src_stride, int channels, int bits, const vs_transform* t,
                      int mode, int border, int max_value, void* dst, int dst_stride, int mem, void* stream) try {
    return bgr_warp_common(src, 0, 1, w, h, src_stride, channels, bits, t, mode, border, max_value, dst, 0, dst_stride,
                           false, mem, (hipStream_t)stream);
} VS_CATCH_ALL

int vs_bgr_image_warp_batch(const void* src, size_t src_fs, int n_frames, int w, int h, int src_stride, int channels,
                            int bits, const vs_transform* t, int mode, int border, int max_value, void* dst,
                            size_t dst_fs, int dst_stride, int mem, void* stream) try {
    return bgr_warp_common(src, src_fs, n_frames, w, h, src_stride, channels, bits, t, mode, border, max_value, dst,
                           dst_fs, dst_stride, false, mem, (hipStream_t)stream);
} VS_CATCH_ALL

int vs_bgr_image_warp_roi_batch(const void* src, size_t src_fs, int n_frames, int w, int h, int src_stride, int channels,
                                int bits, const vs_transform* t, int mode, int border, int max_value, int roi_x, int roi_y,
                                int roi_w, int roi_h, void* dst, size_t dst_fs, int dst_stride, int mem, void* stream) try {
    const vsk::Roi roi{roi_x, roi_y, roi_w, roi_h};
    return bgr_warp_common(src, src_fs, n_frames, w, h, src_stride, channels, bits, t, mode, border, max_value, dst,
                           dst_fs, dst_stride, false, mem, (hipStream_t)stream, &roi);
} VS_CATCH_ALL

int vs_bgr_image_warp_f32(const void* src, int w, int h, int src_stride, int channels, int bits, const vs_transform* t,
                          int mode, int border, float* dst, int dst_stride, int mem, void* stream) try {
    return bgr_warp_common(src, 0, 1, w, h, src_stride, channels, bits, t, mode, border, 0, dst, 0, dst_stride, true, mem,
                           (hipStream_t)stream);
} VS_CATCH_ALL

int vs_bgr_to_gray(const void* src, int w, int h, int src_stride, int bits, int shift_to_8, uint8_t* dst, int dst_stride,
                   int mem, void* stream) try {
    VS_DIMS(w, h);
    VS_ARG(src && dst && w > 0 && h > 0 && src_stride >= 3 * w && dst_stride >= w && (bits == 8 || bits == 16));
    VS_ARG(shift_to_8 >= 0 && shift_to_8 <= 8);
    if (!vsi::device_ready()) return VS_ERR_HIP;
    hipStream_t s = (hipStream_t)stream;
    Staged a, o;
    VS_TRY(a.in(src, img_span(w, h, src_stride, 3) * (bits / 8), mem, s));
    VS_TRY(o.out_image(dst, (size_t)w, (size_t)h, (size_t)dst_stride, 1, 0, mem));
    VS_HIP(vsk::bgr_to_gray(a.dev, w, h, src_stride, bits, shift_to_8, o.as<uint8_t>(), dst_stride, 1, 0, 0, s));
    return vsi::finish_outputs(mem, s, {&o});
} VS_CATCH_ALL

}  // extern "C"
