// vs_kernels.hpp -- host-callable launchers of the gfx950 kernels (all async on `s`).
// Batched launchers take frame strides in ELEMENTS and a frame count in grid.z / grid.y.
#pragma once

#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace vsk {

// output window of a warp: output pixel (x, y), x < w, y < h, is pixel (x + Roi.x, y + Roi.y) of the full output frame
struct Roi { int x, y, w, h; };

hipError_t calib_copy12(const void* src, void* dst, size_t bytes, hipStream_t s);

hipError_t pyr_down(const uint8_t* in, int w, int h, int in_stride, uint8_t* out, int ow, int oh, int out_stride,
                    int n_frames, size_t in_frame_stride, size_t out_frame_stride, hipStream_t s);
hipError_t bgr_to_gray(const void* src, int w, int h, int src_stride, int bits, int shift_to_8, uint8_t* dst,
                       int dst_stride, int n_frames, size_t src_frame_stride, size_t dst_frame_stride, hipStream_t s);
// BGR -> gray level 0 (dense, stride w) and level 1 (dense, stride w/2) of every frame's pyramid in one pass
hipError_t ingest_pyr(const void* src, int w, int h, int src_stride, int bits, int shift_to_8, uint8_t* g0, uint8_t* g1,
                      int n_frames, size_t src_frame_stride, size_t pyr_frame_stride, hipStream_t s);
hipError_t grad_xy(const uint8_t* in, int w, int h, int stride, float* gx, float* gy, hipStream_t s);
hipError_t grad_argmax(const float* gx, const float* gy, int w, int h, int ts, uint16_t* lmx, uint16_t* lmy,
                       hipStream_t s);
hipError_t sparse_jac(const float* gx, const float* gy, int w, int h, const uint16_t* lmx, const uint16_t* lmy, int nt,
                      float* jx, float* jy, hipStream_t s);
hipError_t keyframe(const uint8_t* img, int w, int h, int stride, int ts, uint16_t* lmx, uint16_t* lmy, float* jx,
                    float* jy, int n_frames, size_t img_frame_stride, size_t lm_frame_stride, size_t jac_frame_stride,
                    hipStream_t s, bool aos = false);   // aos: the engine's table layout ({x, y} pairs, float4 Jacobians), see vs_engine.hip
// every pyramid level of n_frames keyframes in ONE launch: pyr / lm / jac = slot of the first keyframe, levels at the given
// element offsets inside a slot (x-set table at lm_off, y-set 2*nt later; x-set Jacobians at jac_off, y-set 4*nt later);
// always the engine's layout: {x, y} pairs and float4 Jacobians per tile
struct KeyframeLevel { int w, h, ts, tx, ty, strips_x, blocks; size_t img_off, lm_off, jac_off; };
struct KeyframeLevels { int n; KeyframeLevel lv[16]; };
bool keyframe_levels_supported(const KeyframeLevels& L);
hipError_t keyframe_levels(const uint8_t* pyr, uint16_t* lm, float* jac, KeyframeLevels L, int n_frames, size_t pyr_frame_stride,
                           size_t lm_frame_stride, size_t jac_frame_stride, hipStream_t s);
hipError_t sparse_warpdiff(const uint8_t* tmpl, const uint8_t* key, int w, int h, int stride, const uint16_t* lm,
                           int nt, float A, float B, float TX, float TY, uint16_t* out, hipStream_t s);
hipError_t sparse_ica(const uint8_t* tmpl, const uint8_t* key, int w, int h, int stride, const uint16_t* selx, int nx,
                      const uint16_t* sely, int ny, const float* jacx, const float* jacy, float A, float B, float TX,
                      float TY, double* out, hipStream_t s);
hipError_t image_warp(const uint8_t* in, int w, int h, int stride, float A, float B, float TX, float TY, float* out,
                      int ow, int oh, hipStream_t s);
// params_dev: n_frames float4 {A,B,TX,TY} (upper-left based kernel arguments) in device memory
hipError_t bgr_warp_generic(const void* src, int w, int h, int src_stride, int channels, int bits,
                            const float4* params_dev, int mode, int border, int max_value, void* dst, int dst_stride,
                            bool f32out, int n_frames, size_t src_frame_stride, size_t dst_frame_stride, Roi roi, hipStream_t s);
// tuned interleaved 3-channel path, u8 or u16 (vs_warp.hip); hipErrorNotSupported when the grid would overflow
// compact: 0 the standard window; 1 / 2: every frame fits the 20-row windows of the contracted / separable forms' six- / seven-wave instantiations (bgr_warp_c3_compact_shape on the extents)
hipError_t bgr_warp_c3(const void* src, int w, int h, int src_stride, int bits, const float4* params_dev, const float4* extents_dev,
                       int mode, int border, int max_value, void* dst, int dst_stride, int n_frames, size_t src_fs, size_t dst_fs, Roi roi,
                       int compact, hipStream_t s);
int bgr_warp_c3_compact_shape(const float* E4, int n_frames);
// VS_WARP_BILINEAR_CV (cv::warpAffine's fixed-point bilinear): minv_dev = n_frames x 6 doubles, the output -> source matrix of each frame
// (vs_cv_inverse_matrix).  Generic: any channel count, u8 / u16 containers; tuned (vs_warp.hip): interleaved 8-bit BGR saturating at 255,
// hipErrorNotSupported for anything else
hipError_t bgr_warp_cv_generic(const void* src, int w, int h, int src_stride, int channels, int bits, const double* minv_dev, int border,
                               int max_value, void* dst, int dst_stride, int n_frames, size_t src_fs, size_t dst_fs, Roi roi, hipStream_t s);
// (tuned) tab_dev: device scratch of n_frames * bgr_warp_cv_table_ints(bits, roi) ints -- the per-frame coordinate tables, written by a small
// kernel in front of the warp launch on the same stream (cv::warpAffine's adelta / bdelta / row origins, made once per frame as OpenCV makes them)
// minv_host (n_frames x 6 doubles on the host) may stand in for minv_dev when n_frames <= kCvInlineFrames: the matrices then travel as kernel arguments
size_t bgr_warp_cv_table_ints(int bits, Roi roi);
constexpr int kCvInlineFrames = 64;          // 3 KiB of kernel arguments
hipError_t bgr_warp_cv_c3(const void* src, int w, int h, int src_stride, int bits, const double* minv_dev, const double* minv_host, int* tab_dev, int border, int max_value,
                          void* dst, int dst_stride, int n_frames, size_t src_fs, size_t dst_fs, Roi roi, hipStream_t s);
// host side of the tuned kernel's tile prologue: per frame {lo_x, hi_x, lo_y, hi_y} from the kernel parameters {A, B, TX, TY}, for the
// tile of the kernel that bgr_warp_c3 launches for (bits, mode)
void bgr_warp_c3_extents(const float* P4, int n_frames, Roi roi, int bits, int mode, float* E4);

}  // namespace vsk
