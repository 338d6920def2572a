// Which CUs does a stream created with hipExtStreamCreateWithCUMask use?  Workgroups record HW_REG_XCC_ID and HW_REG_HW_ID
// (cu_id[11:8], sh_id[12], se_id[15:13]); printed per mask: XCDs used, distinct (xcc, se, sh, cu) count.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <set>
#include <vector>
__global__ __launch_bounds__(256) void k(unsigned* out) {
    const unsigned hw = __builtin_amdgcn_s_getreg((15 << 11) | (0 << 6) | 4);
    const unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20);
    // stay a little so that the grid spreads over every CU the stream may use
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    while (__builtin_amdgcn_s_memtime() - t0 < 20000) {}
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = hw; out[2 * blockIdx.x + 1] = xcc; }
}
static void run(const std::vector<uint32_t>& mask, const char* what) {
    hipStream_t s;
    if (hipExtStreamCreateWithCUMask(&s, (uint32_t)mask.size(), mask.data()) != hipSuccess) { printf("%s: create failed\n", what); return; }
    const int n = 4096;
    unsigned* d; (void)hipMalloc(&d, 8 * n);
    hipLaunchKernelGGL(k, dim3(n), dim3(256), 0, s, d);
    (void)hipStreamSynchronize(s);
    std::vector<unsigned> h(2 * n); (void)hipMemcpy(h.data(), d, 8 * n, hipMemcpyDeviceToHost);
    std::set<unsigned> cus; int per_xcc[8] = {0};
    std::set<unsigned> xccs;
    for (int i = 0; i < n; i++) {
        const unsigned hw = h[2 * i], xcc = h[2 * i + 1] & 15;
        const unsigned id = (xcc << 16) | (((hw >> 13) & 7) << 8) | (((hw >> 12) & 1) << 4) | ((hw >> 8) & 15);
        if (cus.insert(id).second) per_xcc[xcc & 7]++;
        xccs.insert(xcc);
    }
    printf("%-28s distinct CUs %3zu; per XCC:", what, cus.size());
    for (int x = 0; x < 8; x++) printf(" %d", per_xcc[x]);
    printf("\n");
    (void)hipFree(d); (void)hipStreamDestroy(s);
}
int main() {
    std::vector<uint32_t> all(8, 0xffffffffu);
    run(all, "all 256 bits");
    for (int n : {8, 16, 32, 64}) {
        std::vector<uint32_t> m(8, 0);
        for (int i = 0; i < n; i++) m[i / 32] |= 1u << (i % 32);
        char b[64]; snprintf(b, 64, "low %d bits", n); run(m, b);
        std::vector<uint32_t> c(8, 0xffffffffu);
        for (int i = 0; i < n; i++) c[i / 32] &= ~(1u << (i % 32));
        snprintf(b, 64, "all but low %d bits", n); run(c, b);
    }
    {   // every 8th bit
        std::vector<uint32_t> m(8, 0x01010101u);
        run(m, "every 8th bit (32 bits)");
    }
    return 0;
}
