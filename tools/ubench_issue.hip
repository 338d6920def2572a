// ubench_issue.hip -- VALU issue costs on gfx950 at SETTLED clocks (replaces the round-2 ubench_valu / ubench_mix figures, which
// were taken on 0.07-0.5 ms kernels: the card's shader clock needs ~40 ms of continuous work to settle, so those kernels never
// saw the clock they were converted at -- profiles/r03_exp_clock_settling.txt).
//
// Method: every variant is launched back to back for >= 120 ms before anything is timed; the timed launches follow without a gap.
// Three clocks are reported side by side so that no conversion rests on an assumed frequency:
//   wall      HIP events around 8 launches
//   ticks     s_memtime around the instruction loop of every wave (shader cycles), median over waves
//   clock     delta s_memtime / delta s_memrealtime x 100 MHz inside the kernel (MI355X_MICROARCH.md, DVFS give-back item 6)
// cycles per instruction per SIMD = median wave ticks x (waves per SIMD) / instructions per wave ... the waves of a SIMD share
// its issue port, so with W resident waves each wave's loop takes W times the SIMD's per-instruction cost.
//   build: hipcc --offload-arch=gfx950 -O2 -o tools/bin/ubench_issue tools/ubench_issue.hip      run: tools/bin/ubench_issue
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
constexpr int ITERS = 4096;
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
#define F(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
#define M(i) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
#define A(i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
#define P(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(pm), "v"(pc));
#define PM(i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(pm));
#define PA(i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(pc));
#define C(i) asm volatile("v_cvt_f32_ubyte1 %0, %1" : "=v"(b[i]) : "v"(ui[i]));
#define CU(i) asm volatile("v_cvt_u32_f32 %0, %1" : "=v"(ui[i]) : "v"(a[i]));
#define I(i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(ui[i]) : "v"(ui[(i + 1) & 15]));
#define X(i) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
#define PR(i) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(ui[i]) : "v"(ui[(i + 1) & 15]), "v"(ui[(i + 2) & 15]));
#define R(i) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));
#define FL(i) asm volatile("v_floor_f32 %0, %1" : "=v"(b[i]) : "v"(a[i]));
#define CM(i) asm volatile("v_cmp_ge_f32 vcc, %1, %2\n\tv_cndmask_b32 %0, %0, %3, vcc" : "+v"(a[i]) : "v"(b[i]), "v"(m), "v"(c) : "vcc");
#define DP(i) asm volatile("v_mov_b32_dpp %0, %1 row_shl:1 row_mask:0xf bank_mask:0xf" : "=v"(ui[i]) : "v"(ui[(i + 1) & 15]));
#define FD4(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i & 3]) : "v"(m), "v"(c));
#define FD2(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i & 1]) : "v"(m), "v"(c));
#define FD1(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[0]) : "v"(m), "v"(c));
#define PD4(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i & 3]) : "v"(pm), "v"(pc));
// one ds_read_b128 (conflict-free: lane * 16 bytes) -- LDS issue beside the vector stream
#define L(i) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(q[i & 3]) : "v"(laddr), "n"(((i) & 7) * 1024));
#define LW asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
// v_fma_mix_f32: fp32 fma whose second source is a HALF taken from the low / high 16 bits of a register (converted exactly), so
// that an LDS tile kept in fp16 (u8 / 10-bit samples are exact in fp16) needs half the ds_read traffic and no conversion
#define FM(i) asm volatile("v_fma_mix_f32 %0, %1, %2, %0 op_sel_hi:[0,1,0]" : "+v"(a[i]) : "v"(m), "v"(hq[(i) & 7]));
#define FMH(i) asm volatile("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[0,1,0]" : "+v"(a[i]) : "v"(m), "v"(hq[(i) & 7]));
#define L64(i) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(h2[i & 3]) : "v"(laddr8), "n"(((i) & 7) * 512));
#define L128H(i) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(q[i & 3]) : "v"(laddr), "n"(((i) & 7) * 1024));
#define REP16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)

struct Stamp { unsigned long long t0, t1, r0, r1; };

template <int KIND>
__global__ __launch_bounds__(256) void k(Stamp* stamps, unsigned long long* out, float seed) {
    __shared__ f4 lds[2048];                                           // 32 KB
    float a[16], b[16]; f2 p[16]; unsigned ui[16]; f4 q[4]; f2 h2[4]; unsigned hq[8];
#pragma unroll
    for (int i = 0; i < 16; i++) { a[i] = seed + i + threadIdx.x; b[i] = seed * i; p[i] = f2{a[i], b[i]}; ui[i] = threadIdx.x * 17 + i; }
    for (int i = threadIdx.x; i < 2048; i += 256) lds[i] = f4{seed, seed, seed, seed};
#pragma unroll
    for (int i = 0; i < 4; i++) q[i] = f4{0, 0, 0, 0};
    __syncthreads();
    const unsigned laddr = (unsigned)(uintptr_t)lds + (threadIdx.x & 63) * 16;
    const unsigned laddr8 = (unsigned)(uintptr_t)lds + (threadIdx.x & 63) * 8;
#pragma unroll
    for (int i = 0; i < 4; i++) h2[i] = f2{0, 0};
#pragma unroll
    for (int i = 0; i < 8; i++) hq[i] = 0x3c003c00u + threadIdx.x;
    float m = seed * 0.5f + 1.0f, c = seed + 0.25f;
    f2 pm = {m, m + 1.f}, pc = {c, c + 1.f};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < ITERS; it++) {
        if (KIND == 0) { REP16(F) }
        if (KIND == 1) { REP16(M) }
        if (KIND == 2) { REP16(A) }
        if (KIND == 3) { REP16(P) }
        if (KIND == 4) { REP16(PM) }
        if (KIND == 5) { REP16(PA) }
        if (KIND == 6) { REP16(C) }
        if (KIND == 7) { REP16(CU) }
        if (KIND == 8) { REP16(I) }
        if (KIND == 9) { REP16(X) }
        if (KIND == 10) { REP16(PR) }
        if (KIND == 11) { REP16(R) }
        if (KIND == 12) { REP16(FL) }
        if (KIND == 13) { CM(0) CM(1) CM(2) CM(3) CM(4) CM(5) CM(6) CM(7) }                     // 8 x (cmp + cndmask) = 16 instructions
        if (KIND == 14) { REP16(DP) }
        if (KIND == 15) { F(0) P(0) F(1) P(1) F(2) P(2) F(3) P(3) F(4) P(4) F(5) P(5) F(6) P(6) F(7) P(7) }
        if (KIND == 16) { F(0) F(1) F(2) P(0) F(3) F(4) F(5) P(1) F(6) F(7) F(8) P(2) F(9) F(10) F(11) P(3) }
        if (KIND == 17) { F(0) C(0) F(1) C(1) F(2) C(2) F(3) C(3) F(4) C(4) F(5) C(5) F(6) C(6) F(7) C(7) }
        if (KIND == 18) { REP16(FD4) }
        if (KIND == 19) { REP16(FD2) }
        if (KIND == 20) { REP16(FD1) }
        if (KIND == 21) { REP16(PD4) }
        if (KIND == 22) { M(0) A(0) M(1) A(1) M(2) A(2) M(3) A(3) M(4) A(4) M(5) A(5) M(6) A(6) M(7) A(7) }   // dependent mul -> add pairs (the un-contracted order), 8 chains
        if (KIND == 23) { PM(0) PA(0) PM(1) PA(1) PM(2) PA(2) PM(3) PA(3) PM(4) PA(4) PM(5) PA(5) PM(6) PA(6) PM(7) PA(7) }
        // the warp kernel's inner shape: one ds_read_b128 per 4 fma (contracted: 16 reads, 64 fma per pixel)
        if (KIND == 24) { L(0) F(0) F(1) F(2) F(3) L(1) F(4) F(5) F(6) F(7) L(2) F(8) F(9) F(10) F(11) L(3) F(12) F(13) F(14) F(15) LW }
        if (KIND == 25) { L(0) L(1) L(2) L(3) REP16(F) LW }
        if (KIND == 26) { L(0) L(1) L(2) L(3) L(4) L(5) L(6) L(7) LW }                          // LDS reads alone: 8 per iteration
        if (KIND == 27) { L(0) P(0) P(1) L(1) P(2) P(3) L(2) P(4) P(5) L(3) P(6) P(7) LW }     // the exact kernel: one read per 2 packed (4 flops-pairs)
        if (KIND == 28) { REP16(FM) }
        if (KIND == 29) { FM(0) FMH(1) FM(2) FMH(3) FM(4) FMH(5) FM(6) FMH(7) FM(8) FMH(9) FM(10) FMH(11) FM(12) FMH(13) FM(14) FMH(15) }
        // an fp16 tile: one ds_read_b64 per tap ({B,G,R,1} as four halves), or one ds_read_b128 per TWO taps
        if (KIND == 30) { L64(0) FM(0) FMH(1) FM(2) FMH(3) L64(1) FM(4) FMH(5) FM(6) FMH(7) L64(2) FM(8) FMH(9) FM(10) FMH(11) L64(3) FM(12) FMH(13) FM(14) FMH(15) LW }
        if (KIND == 31) { L128H(0) FM(0) FMH(1) FM(2) FMH(3) FM(4) FMH(5) FM(6) FMH(7) L128H(1) FM(8) FMH(9) FM(10) FMH(11) FM(12) FMH(13) FM(14) FMH(15) LW }
        if (KIND == 32) { L(0) F(0) F(1) F(2) A(3) L(1) F(4) F(5) F(6) A(7) L(2) F(8) F(9) F(10) A(11) L(3) F(12) F(13) F(14) A(15) LW }   // the contracted kernel's tap: read, 3 fma, 1 add
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if ((threadIdx.x & 63) == 0) stamps[blockIdx.x * 4 + (threadIdx.x >> 6)] = Stamp{t0, t1, r0, r1};
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; i++) s += a[i] + b[i] + p[i].x + p[i].y + (float)ui[i];
#pragma unroll
    for (int i = 0; i < 4; i++) s += q[i].x + q[i].y + q[i].z + q[i].w + h2[i].x + h2[i].y;
#pragma unroll
    for (int i = 0; i < 8; i++) s += (float)hq[i];
    if (s == 12345.678f) out[0] = 1;
}

template <int K> static int run(const char* name, int instr_per_iter, Stamp* dst, unsigned long long* d) {
    for (int wps : {1, 2, 4}) {      // (~100 VGPRs and 32 KB of LDS per workgroup: at most 4 waves per SIMD are resident)
        const int blocks = 256 * wps;
        hipEvent_t e0, e1, w0, w1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&w0)); CK(hipEventCreate(&w1));
        // settle: back-to-back launches for >= 120 ms
        CK(hipEventRecord(w0));
        float warm = 0.f;
        while (warm < 120.f) {
            for (int r = 0; r < 8; r++) hipLaunchKernelGGL((k<K>), dim3(blocks), dim3(256), 0, 0, dst, d, 1.0f);
            CK(hipEventRecord(w1)); CK(hipEventSynchronize(w1)); CK(hipEventElapsedTime(&warm, w0, w1));
        }
        CK(hipEventRecord(e0));
        for (int r = 0; r < 8; r++) hipLaunchKernelGGL((k<K>), dim3(blocks), dim3(256), 0, 0, dst, d, 1.0f);
        CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
        float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 8;
        std::vector<Stamp> h((size_t)blocks * 4);
        CK(hipMemcpy(h.data(), dst, h.size() * sizeof(Stamp), hipMemcpyDeviceToHost));
        std::vector<double> ticks, clk;
        for (const Stamp& s : h) { ticks.push_back((double)(s.t1 - s.t0)); if (s.r1 > s.r0) clk.push_back((double)(s.t1 - s.t0) / (double)(s.r1 - s.r0) * 0.1); }
        std::sort(ticks.begin(), ticks.end()); std::sort(clk.begin(), clk.end());
        const double tk = ticks[ticks.size() / 2], ghz = clk.empty() ? 0.0 : clk[clk.size() / 2];
        const double n = (double)ITERS * instr_per_iter;
        printf("%-46s waves/SIMD %d: %7.3f ms wall | wave loop %9.0f ticks | in-kernel clock %.3f GHz | %.2f cyc/instr/SIMD (ticks) | %.2f (wall x clock)\n",
               name, wps, ms, tk, ghz, tk / (n * wps), ms * 1e-3 * ghz * 1e9 / (n * wps));
    }
    return 0;
}
int main() {
    unsigned long long* d; CK(hipMalloc(&d, 64));
    Stamp* st; CK(hipMalloc(&st, sizeof(Stamp) * 256 * 8 * 4));
    printf("gfx950 VALU issue costs at settled clocks: 256 x W workgroups of 256 threads (W waves per SIMD), %d iterations x 16 instructions per wave\n", ITERS);
    printf("cyc/instr/SIMD (ticks) = median wave-loop ticks / instructions per wave / W   [W waves share the SIMD's issue port]\n");
    run<0>("v_fma_f32 x16 independent", 16, st, d); run<1>("v_mul_f32 x16", 16, st, d); run<2>("v_add_f32 x16", 16, st, d);
    run<3>("v_pk_fma_f32 x16", 16, st, d); run<4>("v_pk_mul_f32 x16", 16, st, d); run<5>("v_pk_add_f32 x16", 16, st, d);
    run<6>("v_cvt_f32_ubyte1 x16", 16, st, d); run<7>("v_cvt_u32_f32 x16", 16, st, d); run<8>("v_add_u32 x16", 16, st, d);
    run<9>("v_med3_f32 x16", 16, st, d); run<10>("v_perm_b32 x16", 16, st, d); run<11>("v_rcp_f32 x16", 16, st, d);
    run<12>("v_floor_f32 x16", 16, st, d); run<13>("v_cmp + v_cndmask x8", 16, st, d); run<14>("v_mov_b32_dpp x16", 16, st, d);
    run<15>("fma, pk_fma alternating", 16, st, d); run<16>("3 fma : 1 pk_fma", 16, st, d); run<17>("fma, cvt_ubyte alternating", 16, st, d);
    run<18>("fma, 4 dependent chains", 16, st, d); run<19>("fma, 2 dependent chains", 16, st, d); run<20>("fma, 1 dependent chain", 16, st, d);
    run<21>("pk_fma, 4 dependent chains", 16, st, d);
    run<22>("mul -> add dependent pairs, 8 chains", 16, st, d); run<23>("pk_mul -> pk_add dependent pairs, 8 chains", 16, st, d);
    run<24>("[ds_read_b128 + 4 fma] x4 + wait (20 instr)", 20, st, d); run<25>("4 ds_read_b128, 16 fma, wait (20 instr)", 20, st, d);
    run<26>("8 ds_read_b128 + wait (8 instr)", 8, st, d); run<27>("[ds_read_b128 + 2 pk_fma] x4 + wait (12 instr)", 12, st, d);
    run<28>("v_fma_mix_f32 (f16 lo source) x16", 16, st, d); run<29>("v_fma_mix_f32 lo / hi alternating x16", 16, st, d);
    run<30>("[ds_read_b64 + 4 fma_mix] x4 + wait (20 instr)", 20, st, d); run<31>("[ds_read_b128 + 8 fma_mix] x2 + wait (18 instr)", 18, st, d);
    run<32>("[ds_read_b128 + 3 fma + add] x4 + wait (20 instr)", 20, st, d);
    return 0;
}
