// vs_internal.hpp -- shared by the translation units behind the C ABI.
#pragma once

#include "../../include/vs_amd.h"

#include <stddef.h>
#include <stdint.h>
#include <initializer_list>

namespace vsi {
// records a thread-local message for vs_last_error() and returns `code`
int set_error(int code, const char* fmt, ...) __attribute__((format(printf, 2, 3)));
// No exception crosses the C ABI (include/vs_amd.h).  caught(): inside a catch block, turns the in-flight exception into VS_ERR_NOMEM and a
// message; guarded(f): runs f() and stops whatever it throws.  Every extern "C" entry point that can allocate on the host is a
// function-try-block ending in VS_CATCH_ALL; the engine calls also guard their inner stages, so that their failure protocol runs.
int caught() noexcept;
template <typename F>
inline int guarded(F&& f) noexcept {
    try { return f(); } catch (...) { return caught(); }
}
}  // namespace vsi
#define VS_CATCH_ALL catch (...) { return vsi::caught(); }
#define VS_CATCH_ALL_NULL catch (...) { (void)vsi::caught(); return nullptr; }

#ifdef __HIPCC__
#include <hip/hip_runtime.h>

#define VS_HIP(expr)                                                                                     \
    do {                                                                                                 \
        hipError_t _e = (expr);                                                                          \
        if (_e != hipSuccess)                                                                            \
            return vsi::set_error(VS_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
    } while (0)

namespace vsi {

// Every device / pinned-host allocation of the library goes through these two, so that tests can make the k-th one fail
// (vs_test_fail_alloc, include/vs_amd.h).  On failure *p is nullptr.
hipError_t dev_alloc(void** p, size_t bytes);
hipError_t pinned_alloc(void** p, size_t bytes);

// RAII device allocation used by the host-staged (VS_MEM_HOST) form of the kernel-level calls.
struct DevBuf {
    void* p = nullptr;
    size_t bytes = 0;
    DevBuf() = default;
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    ~DevBuf() { if (p) (void)hipFree(p); }
    hipError_t alloc(size_t n) {
        if (p) { (void)hipFree(p); p = nullptr; }
        bytes = n;
        return dev_alloc(&p, n ? n : 1);
    }
    template <typename T> T* as() const { return (T*)p; }
};

// Staging for the host-memory (VS_MEM_HOST) form of the kernel-level calls.  A block = device memory + a PINNED host mirror of the same size,
// leased from a process-wide pool (vs_capi.hip): no hipMalloc / hipFree per call, and the HIP runtime never sees the caller's pageable memory --
// inputs are copied into the mirror by the CPU and travel as one dense pinned H2D copy; outputs come back as one dense D2H copy into the mirror
// and the CPU scatters the rows into the caller's buffer after the stream has been synchronised.  (Until round 5 outputs went back with one
// hipMemcpy2DAsync per frame straight into pageable, possibly 2-byte-aligned caller memory: profiles/r06_flake.md.)
struct StageBlock {
    void* dev = nullptr;
    void* pin = nullptr;
    size_t cap = 0;
    int device = 0;
};
StageBlock* stage_acquire(size_t bytes);                   // nullptr on failure (last error set)
void stage_release(StageBlock* b);
// "the calling thread has synchronised with every copy and kernel it enqueued on a staged block": bumped by finish_outputs() and by entry
// points that synchronise themselves; a Staged that dies without having seen it (an error return) quiesces the device before its block is reused
unsigned long long host_sync_epoch();
void host_synced();

// An argument that lives wherever the caller said (`mem`): for device memory it is the caller's
// pointer; for host memory it is a staged device copy (uploaded for inputs, downloaded for outputs).
struct Staged {
    StageBlock* blk = nullptr;
    void* dev = nullptr;
    void* host = nullptr;
    size_t bytes = 0;
    bool is_out = false, staged = false, copied_back = false;
    unsigned long long epoch = 0;
    Staged() = default;
    Staged(const Staged&) = delete;
    Staged& operator=(const Staged&) = delete;
    ~Staged();
    int in(const void* ptr, size_t n, int mem, hipStream_t s);
    int out(void* ptr, size_t n, int mem);
    // an image output: `frames` images of `rows` rows of `row_bytes` bytes, `pitch` bytes from row to row and `frame_pitch` from
    // image to image.  Only the rows' own bytes reach the caller: the padding between them may be the caller's neighbouring pixels.
    int out_image(void* ptr, size_t row_bytes, size_t rows, size_t pitch, size_t frames, size_t frame_pitch, int mem);
    int finish(hipStream_t s);   // staged outputs: ONE dense D2H into the pinned mirror (async; complete() after the stream has been synchronised)
    void complete();             // mirror -> the caller's memory (CPU; rows only for pitched images)
    size_t row_bytes = 0, rows = 0, pitch = 0, frames = 0, frame_pitch = 0;   // set by out_image (pitch != row_bytes or gaps between frames)
    template <typename T> T* as() const { return (T*)dev; }
private:
    int lease(size_t n);
};
// the tail of every kernel-level call: D2H of the staged outputs, (host memory) synchronise, rows into the caller's buffers
int finish_outputs(int mem, hipStream_t s, std::initializer_list<Staged*> outs);

bool device_ready();   // true when a HIP device is usable (sets last error otherwise)
// Set by the engine around warp launches that run beside the NEXT group's alignment (vs_stabilizer_process_batch / _clips, overlapped): the small-footprint
// solver build moves into a CU as soon as ONE warp workgroup leaves it, which needs the warp's workgroup to hold at least the solver's 35 KB of LDS -- the
// standard 31 KB window does (with the CU's slack), the COMPACT six-wave instantiation's 26 KB does not (c5: 20.3 k -> 18.4 k frames/s with it).  Thread-local:
// a handle is used by one thread at a time, and the hint must not reach another thread's calls.
bool& warp_keeps_solver_slot();
// `s` is about to be destroyed: wait for and drop everything the library still tracks on it (bgr_image_warp's parameter ring
// keeps an event per in-flight call; an event must not outlive the stream it was recorded on)
hipError_t retire_stream(hipStream_t s);   // first error of the waits (the references are dropped either way)

}  // namespace vsi
#endif
