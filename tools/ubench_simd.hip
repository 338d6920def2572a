// Which SIMD does each wave of a 512-thread workgroup land on?  (HW_REG_HW_ID: wave_id[3:0], simd_id[5:4], cu_id[11:8])
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(512) void k(unsigned* out) {
    const unsigned hw = __builtin_amdgcn_s_getreg((15 << 11) | (0 << 6) | 4);
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 8 + (threadIdx.x >> 6)] = hw;
}
int main() {
    unsigned* d; (void)hipMalloc(&d, 4 * 8 * 64);
    for (int rep = 0; rep < 2; rep++) {
        k<<<rep ? 300 : 1, 512>>>(d);
        (void)hipDeviceSynchronize();
        unsigned h[8 * 64]; (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        for (int b = 0; b < (rep ? 3 : 1); b++) {
            printf("block %d:", b);
            for (int w = 0; w < 8; w++) printf("  w%d simd %u slot %u cu %u", w, (h[b * 8 + w] >> 4) & 3, h[b * 8 + w] & 15, (h[b * 8 + w] >> 8) & 15);
            printf("\n");
        }
    }
    return 0;
}
