// ubench_occupancy: does gfx950's VALU issue rate keep improving beyond 4 waves per SIMD?  (tools/ubench_issue.hip's kernel carries ~100
// VGPRs and 32 KB of LDS, so it stops at 4.)  Lean kernels, W = 1 ... 8 waves per SIMD (256 x W workgroups of 256 threads), settled clocks,
// wall time x in-kernel clock / instructions per SIMD:
//   fma16      16 independent v_fma_f32 per iteration
//   tap        the contracted Lanczos tap: ds_read_b128 (4 KB tile), v_mul_f32 (w2d), 4 v_fma_f32  -- x4 per iteration, one wait
//   mix        8 v_fma_f32 + 4 v_mul_f32 + 2 v_cvt_f32_ubyte0 + 2 v_sub_f32 per iteration (roughly the sampler's mix)
//   pkfma16 / pk|fma by wave / pk,fma alternating: is the packed + scalar mixing penalty per wave (instruction adjacency) or per SIMD?
// build: hipcc --offload-arch=gfx950 -O3 -o tools/build/ubench_occupancy tools/ubench_occupancy.hip
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "Error: %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
constexpr int ITERS = 4096;
typedef float f4 __attribute__((ext_vector_type(4)));
struct Stamp { unsigned long long t0, t1, r0, r1; };

template <int KIND>
__global__ __launch_bounds__(256, 8) void k(Stamp* stamps, float* out, float seed) {
    __shared__ f4 tile[256];
    tile[threadIdx.x] = f4{seed, seed * 2.f, seed * 3.f, 1.f};
    __syncthreads();
    typedef float f2 __attribute__((ext_vector_type(2)));
    float a[16];
    f2 pa[8];
#pragma unroll
    for (int i = 0; i < 16; i++) a[i] = seed + (float)i;
#pragma unroll
    for (int i = 0; i < 8; i++) pa[i] = f2{seed - (float)i, seed * (float)i};
    float x = seed * 0.5f, y = seed * 0.25f;
    unsigned u = __float_as_uint(seed) | 0x01020304u;
    const __attribute__((address_space(3))) f4* t = (const __attribute__((address_space(3))) f4*)tile + (threadIdx.x & 63);
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    if (KIND == 4) {
        // waves of alternate workgroups run 16 v_pk_fma_f32 / 16 v_fma_f32 per iteration: each SIMD hosts both kinds side by side, no wave mixes them
        // (workgroups are dealt round-robin over the 8 XCDs: bit 3 of the id alternates between the workgroups that share a CU)
        const f2 px = f2{x, y}, py = f2{y, x};
        if (((blockIdx.x >> 3) & 1) == 0) {
#pragma unroll 1
            for (int it = 0; it < ITERS; it++) {
#pragma unroll
                for (int i = 0; i < 16; i++) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(pa[i & 7]) : "v"(px), "v"(py));
            }
        } else {
#pragma unroll 1
            for (int it = 0; it < ITERS; it++) {
#pragma unroll
                for (int i = 0; i < 16; i++) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(x), "v"(y));
            }
        }
    } else
#pragma unroll 1
    for (int it = 0; it < ITERS; it++) {
        if (KIND == 0) {
#pragma unroll
            for (int i = 0; i < 16; i++) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(x), "v"(y));
        } else if (KIND == 1) {
            f4 v[4];
#pragma unroll
            for (int j = 0; j < 4; j++) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v[j]) : "v"(t), "n"(1024 * 0 + 16 * 64 * 0) : "memory");
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]));
#pragma unroll
            for (int j = 0; j < 4; j++) {
                float w;
                asm volatile("v_mul_f32 %0, %1, %2" : "=v"(w) : "v"(x), "v"(a[12 + j]));
                asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[0]) : "v"(w), "v"(v[j].x));
                asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[1]) : "v"(w), "v"(v[j].y));
                asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[2]) : "v"(w), "v"(v[j].z));
                asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[3]) : "v"(w), "v"(v[j].w));
            }
        } else if (KIND == 3 || KIND == 5) {
            // 3: every wave runs 16 v_pk_fma_f32; 5: every wave alternates v_pk_fma_f32 and v_fma_f32 instruction by instruction
            const f2 px = f2{x, y}, py = f2{y, x};
            if (KIND == 5) {
#pragma unroll
                for (int i = 0; i < 8; i++) {
                    asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(pa[i & 3]) : "v"(px), "v"(py));
                    asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[8 + i]) : "v"(x), "v"(y));
                }
            } else {
#pragma unroll
                for (int i = 0; i < 16; i++) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(pa[i & 7]) : "v"(px), "v"(py));
            }
        } else {
#pragma unroll
            for (int i = 0; i < 8; i++) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(x), "v"(y));
#pragma unroll
            for (int i = 8; i < 12; i++) asm volatile("v_mul_f32 %0, %1, %0" : "+v"(a[i]) : "v"(x));
            asm volatile("v_cvt_f32_ubyte0 %0, %1" : "=v"(a[12]) : "v"(u));
            asm volatile("v_cvt_f32_ubyte0 %0, %1" : "=v"(a[13]) : "v"(u));
            asm volatile("v_sub_f32 %0, %1, %0" : "+v"(a[14]) : "v"(y));
            asm volatile("v_sub_f32 %0, %1, %0" : "+v"(a[15]) : "v"(y));
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if ((threadIdx.x & 63) == 0) stamps[blockIdx.x * 4 + (threadIdx.x >> 6)] = Stamp{t0, t1, r0, r1};
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; i++) s += a[i];
#pragma unroll
    for (int i = 0; i < 8; i++) s += pa[i].x + pa[i].y;
    if (s == 12345.678f) out[0] = s;
}

template <int K> static int run(const char* name, int valu_per_iter, Stamp* dst, float* d) {
    for (int wps = 1; wps <= 8; wps++) {
        const int blocks = 256 * wps;
        hipEvent_t e0, e1, w0, w1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&w0)); CK(hipEventCreate(&w1));
        CK(hipEventRecord(w0));
        float warm = 0.f;
        while (warm < 100.f) {
            for (int r = 0; r < 8; r++) hipLaunchKernelGGL((k<K>), dim3(blocks), dim3(256), 0, 0, dst, d, 1.0f);
            CK(hipEventRecord(w1)); CK(hipEventSynchronize(w1)); CK(hipEventElapsedTime(&warm, w0, w1));
        }
        CK(hipEventRecord(e0));
        for (int r = 0; r < 8; r++) hipLaunchKernelGGL((k<K>), dim3(blocks), dim3(256), 0, 0, dst, d, 1.0f);
        CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
        float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 8;
        std::vector<Stamp> h((size_t)blocks * 4);
        CK(hipMemcpy(h.data(), dst, h.size() * sizeof(Stamp), hipMemcpyDeviceToHost));
        std::vector<double> clk;
        for (const Stamp& s : h) if (s.r1 > s.r0) clk.push_back((double)(s.t1 - s.t0) / (double)(s.r1 - s.r0) * 0.1);
        std::sort(clk.begin(), clk.end());
        const double ghz = clk.empty() ? 0.0 : clk[clk.size() / 2];
        const double n = (double)ITERS * valu_per_iter;
        std::printf("%-20s waves/SIMD %d: %7.3f ms wall | in-kernel clock %.3f GHz | %.2f cycles per VALU instruction per SIMD\n", name, wps, ms, ghz,
                    ms * 1e-3 * ghz * 1e9 / (n * wps));
    }
    return 0;
}
int main() {
    float* d; CK(hipMalloc(&d, 64));
    Stamp* st; CK(hipMalloc(&st, sizeof(Stamp) * 256 * 8 * 4));
    run<0>("fma16", 16, st, d); run<1>("tap", 20, st, d); run<2>("mix", 16, st, d);
    run<3>("pkfma16", 16, st, d); run<4>("pk|fma by wave", 16, st, d); run<5>("pk,fma alternating", 16, st, d);
    return 0;
}
