// vs_phase.hpp -- host interface of the phase-correlation kernels (vs_phase.hip).
#pragma once

#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace vsp {

constexpr int kMaxLine = 4096;     // longest transform line held in LDS (two float2 line buffers + the twiddle table = 96 KB)

// radix plan: all factors 5, then 3, then 4, then at most one 2 (oracle/vs_phase.cpp "Transform specification")
struct Plan {
    int n, passes;
    int radix[16];
};
struct Pair { int32_t prev_slot, cur_slot; };          // cv::phaseCorrelate(PhaseImage[prev], PhaseImage[cur])
struct Result { double dx, dy, response; };            // detected_shift, response (alignment.cpp:372-374)

int optimal_dft_size(int n);                             // cv::getOptimalDFTSize
bool make_plan(int n, Plan& p);

// Everything that depends only on the level-2 extent: padded sizes, plans, twiddle tables in device memory.
struct Context {
    int w = 0, h = 0;            // level-2 image
    int N = 0, M = 0, NC = 0;    // padded columns, rows, kept spectrum columns (N/2 + 1)
    Plan pn{}, pm{};
    float2* twN = nullptr;
    float2* twM = nullptr;
    mutable void* cands = nullptr;          // per-row-block peak candidates of the last correlate() (device)
    mutable size_t cands_bytes = 0;

    hipError_t configure(int width, int height, hipStream_t s);   // hipErrorInvalidValue: a padded extent over kMaxLine
    void destroy();
    size_t spec_frame() const { return (size_t)M * NC; }          // float2 elements per frame spectrum / per pair scratch
    size_t surface_elems() const { return (size_t)M * N; }        // floats per pair surface

    // half spectra of n_frames u8 images (frame f at img + f*img_frame bytes, row stride `stride`) -> spec[f]
    hipError_t spectra(const uint8_t* img, size_t img_frame, int stride, int n_frames, float2* spec, hipStream_t s) const;
    // per pair: surface (unshifted, unscaled) and {dx, dy, response}.  G: n_pairs*spec_frame() scratch
    hipError_t correlate(const float2* spec, const Pair* pairs_dev, int n_pairs, float2* G, float* surf, Result* results_dev,
                         hipStream_t s) const;
};

}  // namespace vsp
