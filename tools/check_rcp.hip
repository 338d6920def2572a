// tools/check_rcp.hip -- exhaustive check of the reciprocal sequences the separable warp may use (vs_warp.hip rcp_rn), against the IEEE
// quotient 1.0f / den (hipcc's full division expansion, -fno-fast-math) for EVERY float in [0.5, 2).
//   seq A: v_rcp, one Newton step, two residual corrections (the compiler's own sequence without scaling / fix-up) -- 7 instructions
//   seq B: v_rcp, one Newton step, ONE residual correction                                                         -- 5 instructions
//   seq C: no v_rcp: r0 = 2 - den, two Newton steps (all fmas: reproducible on a CPU); compared on [0.9, 1.1] as an ACCURACY figure
// build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-fast-math -o tools/bin/check_rcp tools/check_rcp.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstring>

__device__ __forceinline__ float seq_a(float den) {
    float r = __builtin_amdgcn_rcpf(den);
    const float e = __builtin_fmaf(-den, r, 1.0f);
    r = __builtin_fmaf(e, r, r);
    float tt = __builtin_fmaf(-den, r, 1.0f);
    float q = __builtin_fmaf(tt, r, r);
    tt = __builtin_fmaf(-den, q, 1.0f);
    return __builtin_fmaf(tt, r, q);
}
__device__ __forceinline__ float seq_b(float den) {
    float r = __builtin_amdgcn_rcpf(den);
    const float e = __builtin_fmaf(-den, r, 1.0f);
    r = __builtin_fmaf(e, r, r);
    const float tt = __builtin_fmaf(-den, r, 1.0f);
    return __builtin_fmaf(tt, r, r);
}
__device__ __forceinline__ float seq_c(float den) {
    float r = 2.0f - den;
    float e = __builtin_fmaf(-den, r, 1.0f);
    r = __builtin_fmaf(e, r, r);
    e = __builtin_fmaf(-den, r, 1.0f);
    return __builtin_fmaf(e, r, r);
}

// counts[0..2]: values where seq A / B / C differ from 1.0f / den; counts[3]: largest |C - true| in ulps inside [0.9, 1.1]; counts[4]: values tested
__global__ void check(uint32_t lo, uint32_t hi, unsigned long long* counts) {
    const uint32_t bits = lo + blockIdx.x * blockDim.x + threadIdx.x;
    if (bits >= hi) return;
    const float den = __uint_as_float(bits);
    float one = 1.0f;
    asm volatile("" : "+v"(one));
    const float want = one / den;
    if (seq_a(den) != want) atomicAdd(&counts[0], 1ULL);
    if (seq_b(den) != want) atomicAdd(&counts[1], 1ULL);
    if (den >= 0.9f && den <= 1.1f) {
        const float c = seq_c(den);
        if (c != want) atomicAdd(&counts[2], 1ULL);
        const long long d = (long long)__float_as_uint(c) - (long long)__float_as_uint(want);
        atomicMax(&counts[3], (unsigned long long)(d < 0 ? -d : d));
        atomicAdd(&counts[5], 1ULL);
    }
    atomicAdd(&counts[4], 1ULL);
}

// the separable sampler's weight sum over EVERY fraction in [0, 1] (lanczos2_fma chains as in vs_device.hpp): min / max as ordered bit patterns
__device__ __forceinline__ float lz(float x) {
    const float x2 = x * x;
    float v = 0.000858519f;
    v = __builtin_fmaf(v, x2, -0.0158853f); v = __builtin_fmaf(v, x2, 0.128693f); v = __builtin_fmaf(v, x2, -0.583468f);
    v = __builtin_fmaf(v, x2, 1.52229f); v = __builtin_fmaf(v, x2, -2.05238f); v = __builtin_fmaf(v, x2, 0.999861f);
    return fabsf(x) >= 2.0f ? 0.0f : v;
}
__global__ void sum_range(uint32_t n, uint32_t* mnmx) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;       // bit pattern of the fraction: 0 .. 0x3f800000 (= 1.0f)
    if (i > n) return;
    const float f = __uint_as_float(i);
    const float s = (lz(-1.0f - f) + lz(0.0f - f)) + (lz(1.0f - f) + lz(2.0f - f));
    atomicMin(&mnmx[0], __float_as_uint(s));                        // s > 0: bit patterns order like the values
    atomicMax(&mnmx[1], __float_as_uint(s));
}

int main() {
    unsigned long long* d; unsigned long long h[6] = {};
    hipMalloc(&d, sizeof(h)); hipMemset(d, 0, sizeof(h));
    const uint32_t lo = 0x3f000000u, hi = 0x40000000u;         // [0.5, 2)
    hipLaunchKernelGGL(check, dim3((hi - lo + 255) / 256), dim3(256), 0, 0, lo, hi, d);
    if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed\n"); return 1; }
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("floats in [0.5, 2) tested: %llu\n", h[4]);
    printf("seq A (rcp + Newton + 2 corrections, 7 instr) != 1.0f/den : %llu\n", h[0]);
    printf("seq B (rcp + Newton + 1 correction,  5 instr) != 1.0f/den : %llu\n", h[1]);
    printf("seq C (2 - den, two Newton steps, 5 fma-class instr) on [0.9, 1.1]: %llu tested, %llu != 1.0f/den, worst %llu ulp\n", h[5], h[2], h[3]);
    uint32_t* m; uint32_t hm[2] = {0xffffffffu, 0u};
    hipMalloc(&m, sizeof(hm)); hipMemcpy(m, hm, sizeof(hm), hipMemcpyHostToDevice);
    const uint32_t n = 0x3f800000u;
    hipLaunchKernelGGL(sum_range, dim3(n / 256 + 1), dim3(256), 0, 0, n, m);
    if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed\n"); return 1; }
    hipMemcpy(hm, m, sizeof(hm), hipMemcpyDeviceToHost);
    float lo_s, hi_s; memcpy(&lo_s, &hm[0], 4); memcpy(&hi_s, &hm[1], 4);
    printf("sum of the four contracted weights over all %u fractions in [0, 1]: min %.9g max %.9g  ->  den = (sum wx)(sum wy) in [%.9g, %.9g]\n",
           n + 1, lo_s, hi_s, lo_s * lo_s, hi_s * hi_s);
    return 0;
}
