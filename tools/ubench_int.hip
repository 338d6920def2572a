// ubench_int.hip -- issue costs of the INTEGER vector instructions the fixed-point warp (VS_WARP_BILINEAR_CV, vs_warp.hip) is made of, on gfx950 at
// settled clocks.  Same method as ubench_issue.hip (>= 120 ms of back-to-back launches first; wall time x in-kernel clock / instructions / waves
// per SIMD), 4 and 8 waves per SIMD.  Purpose: which of these run at the fp32 fma's rate (~2.5-2.8 cycles per wave-instruction per SIMD) and
// which at half of it -- an integer kernel is priced in THESE cycles.
//   build: hipcc --offload-arch=gfx950 -O2 -o tools/bin/ubench_int tools/ubench_int.hip      run: tools/bin/ubench_int
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
constexpr int ITERS = 4096;
#define OP3(name, i) asm volatile(name " %0, %1, %2, %0" : "+v"(u[i]) : "v"(u[(i + 1) & 15]), "v"(u[(i + 2) & 15]));
#define OP2(name, i) asm volatile(name " %0, %1, %0" : "+v"(u[i]) : "v"(u[(i + 1) & 15]));
#define DOT2(i) OP3("v_dot2_u32_u16", i)
#define DOT4(i) OP3("v_dot4_u32_u8", i)
#define MAD24(i) OP3("v_mad_u32_u24", i)
#define ADD3(i) OP3("v_add3_u32", i)
#define LSHLOR(i) OP3("v_lshl_or_b32", i)
#define ANDOR(i) OP3("v_and_or_b32", i)
#define BFI(i) OP3("v_bfi_b32", i)
#define PERM(i) OP3("v_perm_b32", i)
#define ALIGN(i) OP3("v_alignbit_b32", i)
#define PKMAD(i) OP3("v_pk_mad_u16", i)
#define MUL24(i) OP2("v_mul_u32_u24", i)
#define MULLO(i) asm volatile("v_mul_lo_u32 %0, %1, %0" : "+v"(u[i]) : "v"(u[(i + 1) & 15]));
#define LSHR(i) OP2("v_lshrrev_b32", i)
#define AND(i) OP2("v_and_b32", i)
#define ADDU(i) OP2("v_add_u32", i)
#define BFE(i) asm volatile("v_bfe_u32 %0, %0, 5, 5" : "+v"(u[i]));
#define ASHR(i) asm volatile("v_ashrrev_i32 %0, 10, %0" : "+v"(u[i]));
#define SDWA(i) asm volatile("v_mul_u32_u24_sdwa %0, %1, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD" : "+v"(u[i]) : "v"(u[(i + 1) & 15]));
#define FMA(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f[i]) : "v"(f[(i + 1) & 15]), "v"(f[(i + 2) & 15]));
#define REP16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)
struct Stamp { unsigned long long t0, t1, r0, r1; };

template <int KIND>
__global__ __launch_bounds__(256) void k(Stamp* stamps, unsigned* out, unsigned seed) {
    unsigned u[16]; float f[16];
#pragma unroll
    for (int i = 0; i < 16; i++) { u[i] = seed * 2654435761u + threadIdx.x * 17u + i; f[i] = (float)(seed + i) * 1e-3f; }
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < ITERS; it++) {
        if (KIND == 0) { REP16(DOT2) }
        if (KIND == 1) { REP16(DOT4) }
        if (KIND == 2) { REP16(MAD24) }
        if (KIND == 3) { REP16(MUL24) }
        if (KIND == 4) { REP16(PERM) }
        if (KIND == 5) { REP16(BFE) }
        if (KIND == 6) { REP16(LSHR) }
        if (KIND == 7) { REP16(ASHR) }
        if (KIND == 8) { REP16(ADD3) }
        if (KIND == 9) { REP16(LSHLOR) }
        if (KIND == 10) { REP16(ANDOR) }
        if (KIND == 11) { REP16(BFI) }
        if (KIND == 12) { REP16(AND) }
        if (KIND == 13) { REP16(ADDU) }
        if (KIND == 14) { REP16(ALIGN) }
        if (KIND == 15) { REP16(PKMAD) }
        if (KIND == 16) { REP16(MULLO) }
        if (KIND == 17) { REP16(SDWA) }
        if (KIND == 18) { REP16(FMA) }
        // mixes: does a half-rate integer op hide behind full-rate ones?
        if (KIND == 19) { DOT2(0) ADDU(1) DOT2(2) ADDU(3) DOT2(4) ADDU(5) DOT2(6) ADDU(7) DOT2(8) ADDU(9) DOT2(10) ADDU(11) DOT2(12) ADDU(13) DOT2(14) ADDU(15) }
        if (KIND == 20) { PERM(0) DOT2(1) PERM(2) DOT2(3) PERM(4) DOT2(5) PERM(6) DOT2(7) PERM(8) DOT2(9) PERM(10) DOT2(11) PERM(12) DOT2(13) PERM(14) DOT2(15) }
        if (KIND == 21) { PERM(0) FMA(1) PERM(2) FMA(3) PERM(4) FMA(5) PERM(6) FMA(7) PERM(8) FMA(9) PERM(10) FMA(11) PERM(12) FMA(13) PERM(14) FMA(15) }
        if (KIND == 22) { DOT2(0) FMA(1) DOT2(2) FMA(3) DOT2(4) FMA(5) DOT2(6) FMA(7) DOT2(8) FMA(9) DOT2(10) FMA(11) DOT2(12) FMA(13) DOT2(14) FMA(15) }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if ((threadIdx.x & 63) == 0) stamps[blockIdx.x * 4 + (threadIdx.x >> 6)] = Stamp{t0, t1, r0, r1};
    unsigned s = 0; float fs = 0.f;
#pragma unroll
    for (int i = 0; i < 16; i++) { s += u[i]; fs += f[i]; }
    if (s == 0x12345678u && fs == 1.5f) out[0] = 1;
}

template <int K> static int run(const char* name, Stamp* dst, unsigned* d) {
    for (int wps : {4, 8}) {
        const int blocks = 256 * wps;
        hipEvent_t e0, e1, w0, w1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&w0)); CK(hipEventCreate(&w1));
        CK(hipEventRecord(w0));
        float warm = 0.f;
        while (warm < 120.f) {
            for (int r = 0; r < 8; r++) hipLaunchKernelGGL((k<K>), dim3(blocks), dim3(256), 0, 0, dst, d, 1u);
            CK(hipEventRecord(w1)); CK(hipEventSynchronize(w1)); CK(hipEventElapsedTime(&warm, w0, w1));
        }
        CK(hipEventRecord(e0));
        for (int r = 0; r < 8; r++) hipLaunchKernelGGL((k<K>), dim3(blocks), dim3(256), 0, 0, dst, d, 1u);
        CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
        float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 8;
        std::vector<Stamp> h((size_t)blocks * 4);
        CK(hipMemcpy(h.data(), dst, h.size() * sizeof(Stamp), hipMemcpyDeviceToHost));
        std::vector<double> clk;
        for (const Stamp& s : h) if (s.r1 > s.r0) clk.push_back((double)(s.t1 - s.t0) / (double)(s.r1 - s.r0) * 0.1);
        std::sort(clk.begin(), clk.end());
        const double ghz = clk.empty() ? 0.0 : clk[clk.size() / 2];
        printf("%-34s waves/SIMD %d: %7.3f ms wall | in-kernel clock %.3f GHz | %.2f cycles per wave-instruction per SIMD\n", name, wps, ms, ghz,
               ms * 1e-3 * ghz * 1e9 / ((double)ITERS * 16 * wps));
    }
    return 0;
}
int main() {
    unsigned* d; CK(hipMalloc(&d, 64));
    Stamp* st; CK(hipMalloc(&st, sizeof(Stamp) * 256 * 8 * 4));
    printf("gfx950 integer VALU issue costs at settled clocks (256 x W workgroups of 256 threads, %d x 16 instructions per wave)\n", ITERS);
    run<18>("v_fma_f32 (yardstick)", st, d);
    run<0>("v_dot2_u32_u16", st, d); run<1>("v_dot4_u32_u8", st, d); run<2>("v_mad_u32_u24", st, d); run<3>("v_mul_u32_u24", st, d);
    run<17>("v_mul_u32_u24_sdwa (byte select)", st, d); run<16>("v_mul_lo_u32", st, d); run<15>("v_pk_mad_u16", st, d);
    run<4>("v_perm_b32", st, d); run<14>("v_alignbit_b32", st, d); run<5>("v_bfe_u32", st, d); run<6>("v_lshrrev_b32", st, d); run<7>("v_ashrrev_i32", st, d);
    run<8>("v_add3_u32", st, d); run<9>("v_lshl_or_b32", st, d); run<10>("v_and_or_b32", st, d); run<11>("v_bfi_b32", st, d);
    run<12>("v_and_b32", st, d); run<13>("v_add_u32", st, d);
    run<19>("dot2, add_u32 alternating", st, d); run<20>("perm, dot2 alternating", st, d); run<21>("perm, fma alternating", st, d); run<22>("dot2, fma alternating", st, d);
    return 0;
}
