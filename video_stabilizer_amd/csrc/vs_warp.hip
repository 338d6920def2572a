// vs_warp.hip -- tuned interleaved-BGR u8 bgr_image_warp kernels (placeholder until the tuned path lands).
#include "vs_kernels.hpp"

namespace vsk {
hipError_t bgr_warp_u8c3(const uint8_t*, int, int, int, const float4*, int, int, uint8_t*, int, int, size_t, size_t,
                         hipStream_t) {
    return hipErrorNotSupported;
}
}  // namespace vsk
