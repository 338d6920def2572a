// ubench_mix.hip -- issue cost of MIXED VALU streams on gfx950 (follow-up to ubench_valu.hip): does a scalar fp32
// instruction keep its 2-cycle issue when packed / conversion instructions sit next to it?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
constexpr int ITERS = 2048;
typedef float f2 __attribute__((ext_vector_type(2)));
#define F(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
#define M(i) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
#define A(i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
#define P(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(pm), "v"(pc));
#define PM(i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(pm));
#define PA(i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(pc));
#define C(i) asm volatile("v_cvt_f32_ubyte1 %0, %1" : "=v"(b[i]) : "v"(ui[i]));
#define I(i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(ui[i]) : "v"(ui[(i + 1) & 15]));
#define X(i) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
#define FD(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i & 3]) : "v"(m), "v"(c));     /* 4 dependent chains */
#define PD(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i & 3]) : "v"(pm), "v"(pc));
#define PD2(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i & 1]) : "v"(pm), "v"(pc));
#define PD1(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2\n\ts_nop 0" : "+v"(p[0]) : "v"(pm), "v"(pc));
#define FD1(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[0]) : "v"(m), "v"(c));
#define FD2(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i & 1]) : "v"(m), "v"(c));
template <int KIND>
__global__ __launch_bounds__(256) void k(unsigned long long* out, float seed) {
    float a[16], b[16]; f2 p[16]; unsigned ui[16];
#pragma unroll
    for (int i = 0; i < 16; i++) { a[i] = seed + i + threadIdx.x; b[i] = seed * i; p[i] = f2{a[i], b[i]}; ui[i] = threadIdx.x * 17 + i; }
    float m = seed * 0.5f + 1.0f, c = seed + 0.25f;
    f2 pm = {m, m + 1.f}, pc = {c, c + 1.f};
    for (int it = 0; it < ITERS; it++) {
        if (KIND == 0) { F(0) F(1) F(2) F(3) F(4) F(5) F(6) F(7) F(8) F(9) F(10) F(11) F(12) F(13) F(14) F(15) }
        if (KIND == 1) { P(0) P(1) P(2) P(3) P(4) P(5) P(6) P(7) P(8) P(9) P(10) P(11) P(12) P(13) P(14) P(15) }
        if (KIND == 2) { F(0) P(0) F(1) P(1) F(2) P(2) F(3) P(3) F(4) P(4) F(5) P(5) F(6) P(6) F(7) P(7) }
        if (KIND == 3) { F(0) F(1) F(2) F(3) F(4) F(5) F(6) F(7) P(0) P(1) P(2) P(3) P(4) P(5) P(6) P(7) }
        if (KIND == 4) { F(0) F(1) F(2) P(0) F(3) F(4) F(5) P(1) F(6) F(7) F(8) P(2) F(9) F(10) F(11) P(3) }
        if (KIND == 5) { M(0) A(0) M(1) A(1) M(2) A(2) M(3) A(3) M(4) A(4) M(5) A(5) M(6) A(6) M(7) A(7) }   /* dependent mul->add pairs, 8 chains */
        if (KIND == 6) { PM(0) PA(0) PM(1) PA(1) PM(2) PA(2) PM(3) PA(3) PM(4) PA(4) PM(5) PA(5) PM(6) PA(6) PM(7) PA(7) }
        if (KIND == 7) { PM(0) PM(1) PM(2) PM(3) PM(4) PM(5) PM(6) PM(7) PA(0) PA(1) PA(2) PA(3) PA(4) PA(5) PA(6) PA(7) }
        if (KIND == 8) { F(0) C(0) F(1) C(1) F(2) C(2) F(3) C(3) F(4) C(4) F(5) C(5) F(6) C(6) F(7) C(7) }
        if (KIND == 9) { F(0) I(0) F(1) I(1) F(2) I(2) F(3) I(3) F(4) I(4) F(5) I(5) F(6) I(6) F(7) I(7) }
        if (KIND == 10) { F(0) X(8) F(1) X(9) F(2) X(10) F(3) X(11) F(4) X(12) F(5) X(13) F(6) X(14) F(7) X(15) }
        if (KIND == 11) { FD(0) FD(1) FD(2) FD(3) FD(4) FD(5) FD(6) FD(7) FD(8) FD(9) FD(10) FD(11) FD(12) FD(13) FD(14) FD(15) }
        if (KIND == 12) { PD(0) PD(1) PD(2) PD(3) PD(4) PD(5) PD(6) PD(7) PD(8) PD(9) PD(10) PD(11) PD(12) PD(13) PD(14) PD(15) }
        if (KIND == 13) { PD2(0) PD2(1) PD2(2) PD2(3) PD2(4) PD2(5) PD2(6) PD2(7) PD2(8) PD2(9) PD2(10) PD2(11) PD2(12) PD2(13) PD2(14) PD2(15) }
        if (KIND == 14) { PD1(0) PD1(1) PD1(2) PD1(3) PD1(4) PD1(5) PD1(6) PD1(7) PD1(8) PD1(9) PD1(10) PD1(11) PD1(12) PD1(13) PD1(14) PD1(15) }
        if (KIND == 15) { FD1(0) FD1(1) FD1(2) FD1(3) FD1(4) FD1(5) FD1(6) FD1(7) FD1(8) FD1(9) FD1(10) FD1(11) FD1(12) FD1(13) FD1(14) FD1(15) }
        if (KIND == 16) { FD2(0) FD2(1) FD2(2) FD2(3) FD2(4) FD2(5) FD2(6) FD2(7) FD2(8) FD2(9) FD2(10) FD2(11) FD2(12) FD2(13) FD2(14) FD2(15) }
        if (KIND == 17) { M(0) M(1) M(2) M(3) M(4) M(5) M(6) M(7) A(0) A(1) A(2) A(3) A(4) A(5) A(6) A(7) }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; i++) s += a[i] + b[i] + p[i].x + p[i].y + (float)ui[i];
    if (s == 12345.678f) out[0] = 1;
}
template <int K> static int run(const char* name, unsigned long long* d) {
    for (int wps : {1, 2, 4}) {
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        hipLaunchKernelGGL((k<K>), dim3(256 * wps), dim3(256), 0, 0, d, 1.0f);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        for (int r = 0; r < 4; r++) hipLaunchKernelGGL((k<K>), dim3(256 * wps), dim3(256), 0, 0, d, 1.0f);
        CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
        float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 4;
        printf("%-44s waves/SIMD %d: %.3f ms -> %.2f cyc@2.4GHz per instr per SIMD\n", name, wps, ms, ms * 1e-3 * 2.4e9 / ((double)ITERS * 16 * wps));
    }
    return 0;
}
int main() {
    unsigned long long* d; CK(hipMalloc(&d, 64));
    run<0>("16 fma", d); run<1>("16 pk_fma", d); run<2>("fma,pk alternating", d); run<3>("8 fma then 8 pk", d); run<4>("3 fma : 1 pk", d);
    run<5>("mul->add dependent pairs (8 chains)", d); run<17>("8 mul then 8 dependent add", d); run<6>("pk_mul->pk_add dependent pairs", d); run<7>("8 pk_mul then 8 pk_add", d);
    run<8>("fma,cvt_ubyte alternating", d); run<9>("fma,add_u32 alternating", d); run<10>("fma,med3 alternating", d);
    run<11>("fma, 4 dependent chains", d); run<16>("fma, 2 dependent chains", d); run<15>("fma, 1 dependent chain", d);
    run<12>("pk_fma, 4 dependent chains", d); run<13>("pk_fma, 2 dependent chains", d); run<14>("pk_fma, 1 chain + s_nop", d);
    return 0;
}
