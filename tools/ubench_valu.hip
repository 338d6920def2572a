// ubench_valu.hip -- issue-rate probe for gfx950: how many cycles does a wave64 VALU instruction hold its SIMD,
// for scalar-fp32, packed-fp32, integer and conversion forms, at 1..8 waves per SIMD?  (DESIGN.md cost model of
// bgr_image_warp rests on this number.)  Build: hipcc --offload-arch=gfx950 -O3 tools/ubench_valu.hip -o ubench_valu
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

constexpr int ITERS = 2048;

// 16 independent instructions per iteration, no memory traffic.
#define REP16(OP) OP(0) OP(1) OP(2) OP(3) OP(4) OP(5) OP(6) OP(7) OP(8) OP(9) OP(10) OP(11) OP(12) OP(13) OP(14) OP(15)

template <int KIND>
__global__ __launch_bounds__(256) void k_valu(unsigned long long* out, float seed) {
    float a[16];
    float b[16];
#pragma unroll
    for (int i = 0; i < 16; i++) { a[i] = seed + i + threadIdx.x; b[i] = seed * i; }
    float m = seed * 0.5f + 1.0f, c = seed + 0.25f;
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 p[16];
#pragma unroll
    for (int i = 0; i < 16; i++) p[i] = f2{a[i], b[i]};
    f2 pm = {m, m + 1.f}, pc = {c, c + 1.f};
    unsigned ui[16];
#pragma unroll
    for (int i = 0; i < 16; i++) ui[i] = threadIdx.x * 17 + i;
    __syncthreads();
    unsigned long long t0;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
    for (int it = 0; it < ITERS; it++) {
        if (KIND == 0) {
#define OP(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
            REP16(OP)
#undef OP
        } else if (KIND == 1) {
#define OP(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(pm), "v"(pc));
            REP16(OP)
#undef OP
        } else if (KIND == 2) {
#define OP(i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(pm));
            REP16(OP)
#undef OP
        } else if (KIND == 3) {
#define OP(i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(pm));
            REP16(OP)
#undef OP
        } else if (KIND == 4) {
#define OP(i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(ui[i]) : "v"(ui[(i + 1) & 15]));
            REP16(OP)
#undef OP
        } else if (KIND == 5) {
#define OP(i) asm volatile("v_cvt_f32_ubyte1 %0, %1" : "=v"(a[i]) : "v"(ui[i]));
            REP16(OP)
#undef OP
        } else if (KIND == 6) {
#define OP(i) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
            REP16(OP)
#undef OP
        } else if (KIND == 7) {
#define OP(i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
            REP16(OP)
#undef OP
        } else if (KIND == 8) {   // 8 scalar fma + 8 packed fma interleaved
#define OP(i) asm volatile("v_fma_f32 %0, %0, %2, %3\n\tv_pk_fma_f32 %1, %1, %4, %5" : "+v"(a[i]), "+v"(p[i]) : "v"(m), "v"(c), "v"(pm), "v"(pc));
            OP(0) OP(1) OP(2) OP(3) OP(4) OP(5) OP(6) OP(7)
#undef OP
        } else if (KIND == 9) {   // v_fma_mix_f32 with an f16 operand (lo half of src0)
#define OP(i) asm volatile("v_fma_mix_f32 %0, %1, %2, %0 op_sel_hi:[1,0,0]" : "+v"(a[i]) : "v"(ui[i]), "v"(m));
            REP16(OP)
#undef OP
        } else if (KIND == 10) {  // v_dot2_f32_f16? not on gfx950 -> v_dot2c_f32_f16 ; use v_dot4_u32_u8
#define OP(i) asm volatile("v_dot4_u32_u8 %0, %1, %2, %0" : "+v"(ui[i]) : "v"(ui[(i + 1) & 15]), "v"(ui[(i + 2) & 15]));
            REP16(OP)
#undef OP
        } else if (KIND == 11) {  // v_perm_b32
#define OP(i) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(ui[i]) : "v"(ui[(i + 1) & 15]), "v"(ui[(i + 2) & 15]));
            REP16(OP)
#undef OP
        } else if (KIND == 12) {  // v_rcp_f32
#define OP(i) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));
            REP16(OP)
#undef OP
        } else if (KIND == 13) {  // v_cvt_pk_u8_f32
#define OP(i) asm volatile("v_cvt_pk_u8_f32 %0, %1, 1, %0" : "+v"(ui[i]) : "v"(a[i]));
            REP16(OP)
#undef OP
        } else if (KIND == 14) {  // v_floor_f32
#define OP(i) asm volatile("v_floor_f32 %0, %0" : "+v"(a[i]));
            REP16(OP)
#undef OP
        } else if (KIND == 15) {  // v_fract_f32
#define OP(i) asm volatile("v_fract_f32 %0, %0" : "+v"(a[i]));
            REP16(OP)
#undef OP
        } else if (KIND == 16) {  // v_mad_u32_u24
#define OP(i) asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(ui[i]) : "v"(ui[(i + 1) & 15]), "v"(ui[(i + 2) & 15]));
            REP16(OP)
#undef OP
        } else if (KIND == 17) {  // v_pk_fma_f32 with op_sel broadcast of the low half of src1
#define OP(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2 op_sel_hi:[1,0,1]" : "+v"(p[i]) : "v"(pm), "v"(pc));
            REP16(OP)
#undef OP
        } else if (KIND == 18) {  // DPP mov
#define OP(i) asm volatile("v_mov_b32_dpp %0, %1 row_shl:1 row_mask:0xf bank_mask:0xf" : "=v"(ui[i]) : "v"(ui[(i + 1) & 15]));
            REP16(OP)
#undef OP
        } else if (KIND == 19) {  // v_cvt_f32_u32
#define OP(i) asm volatile("v_cvt_f32_u32 %0, %1" : "=v"(a[i]) : "v"(ui[i]));
            REP16(OP)
#undef OP
        } else if (KIND == 20) {  // v_cndmask
#define OP(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(m));
            REP16(OP)
#undef OP
        } else if (KIND == 21) {  // v_med3_f32
#define OP(i) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
            REP16(OP)
#undef OP
        }
    }
    unsigned long long t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; i++) s += a[i] + p[i].x + p[i].y + (float)ui[i];
    if (s == 12345.678f) out[0] = 1;   // keep everything live
    if ((threadIdx.x & 63) == 0) out[1 + (blockIdx.x * 4 + (threadIdx.x >> 6))] = t1 - t0;
}

// LDS: 16 conflict-free ds_read_b128 per iteration (lane l reads slot l of a row), optional VALU beside it
template <int KIND>
__global__ __launch_bounds__(256) void k_lds(unsigned long long* out, float seed) {
    __shared__ float4 tile[4096];
    for (int i = threadIdx.x; i < 4096; i += 256) tile[i] = float4{seed, seed, seed, 1.f};
    __syncthreads();
    typedef float f4 __attribute__((ext_vector_type(4)));
    f4 acc = {0, 0, 0, 0};
    const unsigned base = (unsigned)(threadIdx.x & 63) * 16u + (unsigned)(threadIdx.x >> 6) * 1024u * 8u;
    unsigned long long t0;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
    for (int it = 0; it < ITERS; it++) {
        f4 v[16];
        if (KIND == 0) {
#define OP(i) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v[i]) : "v"(base), "i"((i & 7) * 1024 + (i >> 3) * 16));
            REP16(OP)
#undef OP
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        } else if (KIND == 1) {   // ds_read_b64
            typedef float f2 __attribute__((ext_vector_type(2)));
            f2 w[16];
            const unsigned b2 = (unsigned)(threadIdx.x & 63) * 8u + (unsigned)(threadIdx.x >> 6) * 1024u * 8u;
#define OP(i) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(w[i]) : "v"(b2), "i"((i & 7) * 1024 + (i >> 3) * 16));
            REP16(OP)
#undef OP
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int i = 0; i < 16; i++) v[i] = f4{w[i].x, w[i].y, 0, 0};
        }
#pragma unroll
        for (int i = 0; i < 16; i++) asm volatile("" :: "v"(v[i]));
        acc += v[it & 15];
    }
    unsigned long long t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");
    if (acc.x == 12345.678f) out[0] = 1;
    if ((threadIdx.x & 63) == 0) out[1 + (blockIdx.x * 4 + (threadIdx.x >> 6))] = t1 - t0;
}

template <typename F>
static int run(const char* name, F launch, int instr_per_iter, unsigned long long* d_out, std::vector<unsigned long long>& h) {
    for (int wps : {1, 2, 4, 8}) {
        const int blocks = 256 * wps;
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        launch(blocks);   // warm
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        launch(blocks);
        CK(hipEventRecord(e1));
        CK(hipDeviceSynchronize());
        float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
        CK(hipMemcpy(h.data(), d_out, sizeof(unsigned long long) * (1 + blocks * 4), hipMemcpyDeviceToHost));
        std::vector<unsigned long long> c(h.begin() + 1, h.begin() + 1 + blocks * 4);
        std::sort(c.begin(), c.end());
        const double med = (double)c[c.size() / 2];
        const double n = (double)ITERS * instr_per_iter;
        // per-SIMD issue interval if the waves of a SIMD ran concurrently for the whole span
        printf("%-28s waves/SIMD %d: wave cycles/instr %.2f  -> SIMD cycles/instr %.2f   (kernel %.3f ms => %.2f cyc/instr/SIMD @2.4GHz)\n", name, wps,
               med / n, med / n / wps, ms, ms * 1e-3 * 2.4e9 / (n * wps));
    }
    return 0;
}

int main() {
    unsigned long long* d_out;
    CK(hipMalloc(&d_out, sizeof(unsigned long long) * (1 + 256 * 8 * 4)));
    std::vector<unsigned long long> h(1 + 256 * 8 * 4);
#define RUNV(K, NAME, N) run(NAME, [&](int blocks) { hipLaunchKernelGGL((k_valu<K>), dim3(blocks), dim3(256), 0, 0, d_out, 1.0f); }, N, d_out, h)
    RUNV(0, "v_fma_f32", 16);
    RUNV(1, "v_pk_fma_f32", 16);
    RUNV(2, "v_pk_mul_f32", 16);
    RUNV(3, "v_pk_add_f32", 16);
    RUNV(17, "v_pk_fma_f32 op_sel", 16);
    RUNV(6, "v_mul_f32", 16);
    RUNV(7, "v_add_f32", 16);
    RUNV(4, "v_add_u32", 16);
    RUNV(16, "v_mad_u32_u24", 16);
    RUNV(5, "v_cvt_f32_ubyte1", 16);
    RUNV(19, "v_cvt_f32_u32", 16);
    RUNV(13, "v_cvt_pk_u8_f32", 16);
    RUNV(9, "v_fma_mix_f32", 16);
    RUNV(10, "v_dot4_u32_u8", 16);
    RUNV(11, "v_perm_b32", 16);
    RUNV(12, "v_rcp_f32", 16);
    RUNV(14, "v_floor_f32", 16);
    RUNV(15, "v_fract_f32", 16);
    RUNV(18, "v_mov_b32_dpp", 16);
    RUNV(20, "v_cndmask_b32", 16);
    RUNV(21, "v_med3_f32", 16);
    RUNV(8, "8 fma + 8 pk_fma", 16);
    run("ds_read_b128 x16 + wait", [&](int blocks) { hipLaunchKernelGGL((k_lds<0>), dim3(blocks), dim3(256), 0, 0, d_out, 1.0f); }, 16, d_out, h);
    run("ds_read_b64 x16 + wait", [&](int blocks) { hipLaunchKernelGGL((k_lds<1>), dim3(blocks), dim3(256), 0, 0, d_out, 1.0f); }, 16, d_out, h);
    return 0;
}
