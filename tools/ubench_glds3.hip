// ubench_glds3.hip -- can LDS-DMA stage packed 3-byte BGR pixels straight into the fixed-point bilinear kernel's tile format (one dword {B,G,R,x} per
// pixel)?  `global_load_lds_dword` with a PER-LANE source address base + 3 * pixel (byte-aligned, not dword-aligned) writes lane-linear dwords: if the
// hardware takes the unaligned dword reads, a row of the tile is one wave-instruction, no VGPRs and no formatting instructions (the kernel's fill is 5
// vector instructions per 12-byte item today: profiles/r05_warp_cv.md).  This program checks (1) that the bytes are right for every byte offset and row
// pitch, (2) what a tile fill of this form costs beside the register form (global_load_dwordx3 + 2 v_perm + ds_write_b128).
// build: hipcc --offload-arch=gfx950 -O3 -o tools/bin/ubench_glds3 tools/ubench_glds3.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(2); } } while (0)

constexpr int ROWS = 72, COLS = 72, PITCH_DW = 80;        // the 8-bit kernel's window: 72 x 72 staged pixels, 80-dword row pitch

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

// one workgroup = one window at (x0, y0) of a w x h packed-BGR image; four waves deal the rows; per row: lanes 0..63 -> columns 0..63, then lanes 0..7 ->
// columns 64..71.  mode 0: LDS-DMA dwords; mode 1: register form (12-byte items, the kernel's own fill).  Then every thread sums its share of the tile
// (so that the fill cannot be dropped) and, if `check`, compares every staged pixel with the image.
template <int MODE>
__global__ __launch_bounds__(256) void k_fill(const uint8_t* __restrict__ img, int pitch, int tiles_x, uint32_t* __restrict__ out, int check, int reps) {
    __shared__ __attribute__((aligned(16))) uint32_t tile[ROWS * PITCH_DW];
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int tx = blockIdx.x % tiles_x, ty = blockIdx.x / tiles_x;
    uint32_t acc = 0, bad = 0;
    for (int rep = 0; rep < reps; rep++) {
        const int x0 = tx * 64 + ((rep + blockIdx.x) & 3), y0 = ty * 64;       // every byte alignment of the window's first pixel
        const uint8_t* base = img + (size_t)y0 * pitch + (size_t)x0 * 3;
        if (MODE == 0) {
            for (int r = wv; r < ROWS; r += 4) {
                const uint8_t* row = base + (size_t)r * pitch;
                __builtin_amdgcn_global_load_lds((gptr_t)(row + 3 * lane), (lptr_t)(tile + r * PITCH_DW), 4, 0, 0);
                if (lane < 8) __builtin_amdgcn_global_load_lds((gptr_t)(row + 3 * (64 + lane)), (lptr_t)(tile + r * PITCH_DW + 64), 4, 0, 0);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else {
            typedef uint32_t u32x3 __attribute__((ext_vector_type(3)));
            typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
            // 18 groups of 4 pixels per row; the window's first pixel aligned down to a dword here (the kernel stages from a multiple of 4 pixels)
            const uint8_t* b4 = img + (size_t)y0 * pitch + (size_t)(tx * 64) * 3;
            for (int it = wv * 64 + lane; it < ROWS * 18; it += 256) {
                const int r = it / 18, g = it - 18 * r;
                const u32x3 q = *(const u32x3*)(b4 + (size_t)r * pitch + 12 * g);
                u32x4 px;
                px.x = q.x & 0x00ffffffu;
                px.y = __builtin_amdgcn_perm(q.y, q.x, 0x0c050403u);
                px.z = __builtin_amdgcn_perm(q.z, q.y, 0x0c040302u);
                px.w = q.z >> 8;
                *(u32x4*)(tile + r * PITCH_DW + 4 * g) = px;
            }
        }
        __syncthreads();
        for (int i = threadIdx.x; i < ROWS * COLS; i += 256) {
            const int r = i / COLS, c = i - COLS * r;
            const uint32_t v = tile[r * PITCH_DW + c] & 0x00ffffffu;
            acc += v;
            if (check && MODE == 0) {
                const uint8_t* p = base + (size_t)r * pitch + 3 * c;
                const uint32_t want = (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16);
                bad += v != want;
            }
        }
        __syncthreads();
    }
    atomicAdd(out, acc);
    if (bad) atomicAdd(out + 1, bad);
}

int main() {
    const int w = 3840 + 16, h = 2160 + 16, tiles_x = 59, tiles_y = 32;
    for (int pitch_extra = 0; pitch_extra < 4; pitch_extra++) {
        const int pitch = 3 * w + pitch_extra;                // every alignment of the rows too
        std::vector<uint8_t> host((size_t)pitch * h);
        uint32_t s = 12345u + pitch_extra;
        for (auto& b : host) { s = s * 1664525u + 1013904223u; b = (uint8_t)(s >> 24); }
        uint8_t* img; uint32_t* out;
        CK(hipMalloc(&img, host.size() + 64)); CK(hipMalloc(&out, 8));
        CK(hipMemcpy(img, host.data(), host.size(), hipMemcpyHostToDevice));
        CK(hipMemset(out, 0, 8));
        hipLaunchKernelGGL(k_fill<0>, dim3(tiles_x * tiles_y), dim3(256), 0, 0, img, pitch, tiles_x, out, 1, 4);
        CK(hipDeviceSynchronize());
        uint32_t r[2];
        CK(hipMemcpy(r, out, 8, hipMemcpyDeviceToHost));
        std::printf("pitch 3w+%d: LDS-DMA dword fill at byte-aligned sources: %u wrong pixels of %d\n", pitch_extra, r[1], tiles_x * tiles_y * 4 * ROWS * COLS);
        if (pitch_extra == 0) {
            for (int mode = 0; mode < 2; mode++) {
                for (int pass = 0; pass < 3; pass++) {
                    // settle the clock, then time
                    for (int i = 0; i < 40; i++) {
                        if (mode == 0) hipLaunchKernelGGL(k_fill<0>, dim3(tiles_x * tiles_y), dim3(256), 0, 0, img, pitch, tiles_x, out, 0, 16);
                        else hipLaunchKernelGGL(k_fill<1>, dim3(tiles_x * tiles_y), dim3(256), 0, 0, img, pitch, tiles_x, out, 0, 16);
                    }
                    CK(hipDeviceSynchronize());
                    const auto t0 = std::chrono::steady_clock::now();
                    const int n = 40;
                    for (int i = 0; i < n; i++) {
                        if (mode == 0) hipLaunchKernelGGL(k_fill<0>, dim3(tiles_x * tiles_y), dim3(256), 0, 0, img, pitch, tiles_x, out, 0, 16);
                        else hipLaunchKernelGGL(k_fill<1>, dim3(tiles_x * tiles_y), dim3(256), 0, 0, img, pitch, tiles_x, out, 0, 16);
                    }
                    CK(hipDeviceSynchronize());
                    const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / n;
                    const double bytes = (double)tiles_x * tiles_y * 16 * ROWS * COLS * 3;
                    std::printf("  %s fill + LDS read-back, %d tiles x 16: %.1f us per launch = %.2f TB/s of staged pixels (%.2f us per 4K frame's worth of tiles)\n",
                                mode == 0 ? "LDS-DMA dword (2 instructions per row)" : "register (dwordx3 + 2 v_perm + ds_write_b128)", tiles_x * tiles_y, us,
                                bytes / us / 1e6, us / 16.0 * 2040.0 / (tiles_x * tiles_y));
                }
            }
        }
        CK(hipFree(img)); CK(hipFree(out));
    }
    return 0;
}
