// ubench_format_load: does gfx950's texture path convert packed BGR bytes to floats on the way in, and at what rate?
//   buffer_load_format_xyzw through a buffer resource {stride 3 bytes, DATA_FORMAT 8_8_8_8, NUM_FORMAT USCALED, DST_SEL {R,G,B,1}}:
//   lane i of an idxen load gets {float(b[3i]), float(b[3i+1]), float(b[3i+2]), 1.0f} -- the float4 the Lanczos tile stores per pixel.
// (1) values against the bytes, incl. an index past num_records (must read {0,0,0,1}); (2) cycles per wave-level load, 4 and 8 waves per
// SIMD, rows of 80 pixels like the warp kernel's tile fill, against the kernel's present fill (global_load_dwordx3 + 16 v_cvt_f32_ubyte).
// build: hipcc --offload-arch=gfx950 -O3 -o tools/build/ubench_format_load tools/ubench_format_load.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>

typedef float f4 __attribute__((ext_vector_type(4)));
typedef int i4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x3 __attribute__((ext_vector_type(3)));

__device__ __forceinline__ i4 make_rsrc(const void* base, uint32_t stride, uint32_t records, uint32_t data_format) {
    const uint64_t b = (uint64_t)base;
    i4 r;
    r.x = (int)(uint32_t)b;
    r.y = (int)(((uint32_t)(b >> 32) & 0xffffu) | (stride << 16));
    r.z = (int)records;
    r.w = (int)(4u | (5u << 3) | (6u << 6) | (1u << 9) | (2u << 12) | (data_format << 15));   // dst_sel R G B 1, USCALED
    // the descriptor must live in SGPRs: make it uniform for the compiler
    r.x = __builtin_amdgcn_readfirstlane(r.x); r.y = __builtin_amdgcn_readfirstlane(r.y);
    r.z = __builtin_amdgcn_readfirstlane(r.z); r.w = __builtin_amdgcn_readfirstlane(r.w);
    return r;
}

__global__ void k_values(const uint8_t* src, f4* out, int n_px, int n_out) {
    const i4 rs = make_rsrc(src, 3, (uint32_t)n_px, 10);
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_out) return;
    f4 v;
    asm volatile("buffer_load_format_xyzw %0, %1, %2, 0 idxen\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(i), "s"(rs) : "memory");
    out[i] = v;
}

__global__ void k_values16(const uint16_t* src, f4* out, int n_px, int n_out) {
    const i4 rs = make_rsrc(src, 6, (uint32_t)n_px, 12);
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_out) return;
    f4 v;
    asm volatile("buffer_load_format_xyzw %0, %1, %2, 0 idxen\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(i), "s"(rs) : "memory");
    out[i] = v;
}

// one workgroup = 256 threads fills a 24 x 80 pixel "tile" (rows `pitch` bytes apart) per iteration, and sums what it got
template <int FORM>
__global__ __launch_bounds__(256) void k_rate(const uint8_t* src, int pitch, int rows_total, int iters, float* sink, unsigned long long* cyc) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    float acc = 0.f;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; it++) {
        const int row0 = (int)((blockIdx.x * 131u + (unsigned)it * 24u) % (unsigned)(rows_total - 24));
        if (FORM == 0) {
            // 1920 pixels = 30 wave loads; wave w takes loads w, w + 4, ...: pixel p = 64 * k + lane -> row p / 80, col p % 80
            f4 v[8];
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const int p = 64 * (wv + 4 * k) + lane;
                const int row = p / 80, col = p - 80 * row;
                const int idx = p < 1920 ? (row0 + row) * (pitch / 3) + col : 0x7fffffff;    // (pitch a multiple of 3 here)
                const i4 rs = make_rsrc(src, 3, (uint32_t)(rows_total * (pitch / 3)), 10);
                asm volatile("buffer_load_format_xyzw %0, %1, %2, 0 idxen" : "=v"(v[k]) : "v"(idx), "s"(rs) : "memory");
            }
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]) :: "memory");
#pragma unroll
            for (int k = 0; k < 8; k++) acc += v[k].x + v[k].y + v[k].z + v[k].w;
        } else {
            // the kernel's present fill: 480 items of 4 pixels (12 bytes), 2 per thread, 16 conversions each
#pragma unroll
            for (int s = 0; s < 2; s++) {
                const int item = threadIdx.x + 256 * s;
                if (item < 480) {
                    const int row = item / 20, g = item - 20 * row;
                    const u32x3 q = *(const u32x3*)(src + (size_t)(row0 + row) * pitch + 12 * g);
                    acc += (float)(q.x & 255) + (float)((q.x >> 8) & 255) + (float)((q.x >> 16) & 255) + (float)(q.x >> 24) +
                           (float)(q.y & 255) + (float)((q.y >> 8) & 255) + (float)((q.y >> 16) & 255) + (float)(q.y >> 24) +
                           (float)(q.z & 255) + (float)((q.z >> 8) & 255) + (float)((q.z >> 16) & 255) + (float)(q.z >> 24) + 4.0f;
                }
            }
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (acc == -1.f) sink[0] = acc;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "Error: %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

int main() {
    const int n_px = 1000;
    std::vector<uint8_t> h(3 * n_px + 8);
    for (size_t i = 0; i < h.size(); i++) h[i] = (uint8_t)((i * 37 + 11) & 255);
    uint8_t* d; f4* o;
    CK(hipMalloc((void**)&d, h.size())); CK(hipMalloc((void**)&o, sizeof(f4) * 1024));
    CK(hipMemcpy(d, h.data(), h.size(), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_values, dim3(4), dim3(256), 0, 0, d, o, n_px, 1024);
    std::vector<f4> r(1024);
    CK(hipMemcpy(r.data(), o, sizeof(f4) * 1024, hipMemcpyDeviceToHost));
    int bad = 0;
    for (int i = 0; i < 1024; i++) {
        const bool in = i < n_px;
        const float e[4] = {in ? (float)h[3 * i] : 0.f, in ? (float)h[3 * i + 1] : 0.f, in ? (float)h[3 * i + 2] : 0.f, 1.0f};
        if (r[i].x != e[0] || r[i].y != e[1] || r[i].z != e[2] || r[i].w != e[3]) { if (bad++ < 5) std::printf("u8 px %d: got %g %g %g %g want %g %g %g %g\n", i, r[i].x, r[i].y, r[i].z, r[i].w, e[0], e[1], e[2], e[3]); }
    }
    std::printf("u8 8_8_8_8 USCALED stride 3: %d of 1024 lanes wrong (24 of them past num_records)\n", bad);
    // 16-bit: 6-byte stride, 16_16_16_16
    std::vector<uint16_t> h16(3 * n_px + 4);
    for (size_t i = 0; i < h16.size(); i++) h16[i] = (uint16_t)((i * 2657 + 77) & 0xffff);
    uint16_t* d16; CK(hipMalloc((void**)&d16, h16.size() * 2));
    CK(hipMemcpy(d16, h16.data(), h16.size() * 2, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_values16, dim3(4), dim3(256), 0, 0, d16, o, n_px, 1024);
    CK(hipMemcpy(r.data(), o, sizeof(f4) * 1024, hipMemcpyDeviceToHost));
    bad = 0;
    for (int i = 0; i < 1024; i++) {
        const bool in = i < n_px;
        const float e[4] = {in ? (float)h16[3 * i] : 0.f, in ? (float)h16[3 * i + 1] : 0.f, in ? (float)h16[3 * i + 2] : 0.f, 1.0f};
        if (r[i].x != e[0] || r[i].y != e[1] || r[i].z != e[2] || r[i].w != e[3]) { if (bad++ < 5) std::printf("u16 px %d: got %g %g %g %g want %g %g %g %g\n", i, r[i].x, r[i].y, r[i].z, r[i].w, e[0], e[1], e[2], e[3]); }
    }
    std::printf("u16 16_16_16_16 USCALED stride 6: %d of 1024 lanes wrong\n", bad);

    // rate: a 4K frame's worth of rows, workgroups = 256 CUs x {4, 8} per CU
    const int pitch = 3840 * 3, rows = 2160;
    uint8_t* img; CK(hipMalloc((void**)&img, (size_t)pitch * rows + 64)); CK(hipMemset(img, 7, (size_t)pitch * rows + 64));
    float* sink; unsigned long long* cyc; CK(hipMalloc((void**)&sink, 4)); CK(hipMalloc((void**)&cyc, 8 * 4096));
    for (int per_cu : {4, 8}) {
        for (int form = 0; form < 2; form++) {
            const int wgs = 256 * per_cu, iters = 64;
            double best = 1e30; unsigned long long cmean = 0;
            for (int rep = 0; rep < 6; rep++) {
                hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
                CK(hipEventRecord(a, 0));
                if (form == 0) hipLaunchKernelGGL(k_rate<0>, dim3(wgs), dim3(256), 0, 0, img, pitch, rows, iters, sink, cyc);
                else hipLaunchKernelGGL(k_rate<1>, dim3(wgs), dim3(256), 0, 0, img, pitch, rows, iters, sink, cyc);
                CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b));
                float ms; CK(hipEventElapsedTime(&ms, a, b));
                if (ms < best) best = ms;
                std::vector<unsigned long long> c(wgs);
                CK(hipMemcpy(c.data(), cyc, 8 * wgs, hipMemcpyDeviceToHost));
                cmean = 0; for (auto v : c) cmean += v; cmean /= wgs;
            }
            const double tiles = (double)wgs * iters;
            std::printf("%s, %d workgroups per CU: %.3f ms for %.0f tile fills = %.1f ns per tile per CU (%.2f us per 4K frame's 8100 tiles over 256 CUs); %llu cycles per workgroup-iteration\n",
                        form == 0 ? "buffer_load_format_xyzw (1 per pixel)" : "global_load_dwordx3 + 16 cvt (per 4 pixels)", per_cu, best, tiles,
                        best * 1e6 / (tiles / 256.0), best * 1e3 / tiles * 8100.0, cmean / iters);
        }
    }
    return 0;
}
