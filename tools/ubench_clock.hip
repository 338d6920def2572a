// Which clock does s_memtime count?  One wave spins for N ticks; HIP events give the wall time.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void spin(unsigned long long n, unsigned long long* out) {
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    unsigned long long t;
    do { t = __builtin_amdgcn_s_memtime(); } while (t - t0 < n);
    out[0] = t - t0;
    out[1] = __builtin_amdgcn_s_memrealtime() - r0;
}
int main() {
    unsigned long long* d; hipMalloc(&d, 16);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (unsigned long long n : {1000000ull, 10000000ull, 10000000ull}) {
        hipEventRecord(a); spin<<<1, 64>>>(n, d); hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        unsigned long long h[2]; hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
        printf("s_memtime ticks %llu  s_memrealtime ticks %llu  wall %.3f ms -> %.1f MHz (memtime), %.1f MHz (memrealtime)\n", h[0], h[1], ms, h[0] / ms / 1e3, h[1] / ms / 1e3);
    }
    return 0;
}
