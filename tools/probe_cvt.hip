// probe_cvt.hip -- what does v_cvt_pk_u8_f32 do with fractions, negatives and values above 255 on gfx950?
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(const float* in, unsigned* out, int n) {
    int i = threadIdx.x;
    if (i >= n) return;
    unsigned r = 0;
    asm volatile("v_cvt_pk_u8_f32 %0, %1, 0, %0" : "+v"(r) : "v"(in[i]));
    out[i] = r;
}
int main() {
    const float h[] = {0.0f, 0.49f, 0.5f, 0.51f, 0.999f, 1.0f, 1.5f, 2.5f, 3.5f, 254.5f, 254.99f, 255.0f, 255.5f, 256.0f, 300.0f, 1e9f, -0.5f, -1.0f, -100.f, 127.49999f, 127.5f, 128.5f};
    const int n = sizeof(h) / sizeof(h[0]);
    float* d; unsigned* o; unsigned ho[64];
    hipMalloc(&d, sizeof(h)); hipMalloc(&o, 256);
    hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, o, n);
    hipMemcpy(ho, o, n * 4, hipMemcpyDeviceToHost);
    for (int i = 0; i < n; i++) printf("%g -> %u\n", h[i], ho[i]);
    return 0;
}
