/* vs_halide_abi.h -- link-level drop-in for the reference's Halide AOT pipelines.
 *
 * The reference's imgproc.cpp includes sixteen generated headers (imgproc.cpp:9-24) and calls the functions they declare
 * with halide_buffer_t* arguments (imgproc.cpp:42, 62-76, 94-104, 112, 131, 140, 176-194).  libvs_halide_abi.so exports the
 * same sixteen C symbols with the same argument lists -- inputs in declaration order, scalars by value, outputs last, int
 * result, 0 = success (Halide's AOT calling convention; generators.cpp:56-739 for the declaration order) -- implemented on
 * libvs_amd.so's HIP kernels.  A maintainer relinks imgproc.cpp against this library instead of the sixteen generated static
 * libraries (CMakeLists.txt:156-271) and changes nothing else: INTEGRATION.md, "Route 0".
 *
 * The structs below are layout-identical to HalideRuntime.h's halide_buffer_t / halide_dimension_t / halide_type_t (Halide
 * 10 .. 19: the layout is part of Halide's stable C ABI).  HalideRuntime.h is not available in this build environment, so
 * the layout is restated here and pinned by static_asserts on every offset; where a translation unit has already included
 * HalideRuntime.h, define VS_HALIDE_ABI_USE_HALIDE_RUNTIME before this header and the real types are used instead.
 *
 * What the entry points accept: host-resident buffers (host != NULL; the reference runs Halide on the CPU and never sets a
 * device handle), every dimension with min == 0, dim[0].stride == 1, the element types and ranks of the generators'
 * Input / Output declarations.  Rows of u8 images may be padded (dim[1].stride >= dim[0].extent); float / u16 / f64 planes
 * must be dense, as Halide::Runtime::Buffer allocates them.  Anything else returns a halide_error_code_t value (non-zero:
 * the reference's wrappers only test `== 0`) and leaves vs_last_error() set.  Bounds-query mode (all-NULL buffers) is not
 * offered: imgproc.cpp never uses it.  Results are those of the corresponding vs_* kernel-level entry (include/vs_amd.h):
 * bit-identical to the CPU restatement of the generators.
 */
#ifndef VS_HALIDE_ABI_H
#define VS_HALIDE_ABI_H

#include <stddef.h>
#include <stdint.h>

#ifdef VS_HALIDE_ABI_USE_HALIDE_RUNTIME
typedef struct halide_buffer_t vs_halide_buffer_t;
#else
struct halide_device_interface_t;

/* halide_type_code_t: halide_type_int = 0, halide_type_uint = 1, halide_type_float = 2, halide_type_handle = 3 */
enum { VS_HALIDE_TYPE_INT = 0, VS_HALIDE_TYPE_UINT = 1, VS_HALIDE_TYPE_FLOAT = 2, VS_HALIDE_TYPE_HANDLE = 3 };

typedef struct vs_halide_type_t {
    uint8_t code;      /* halide_type_code_t */
    uint8_t bits;
    uint16_t lanes;
} vs_halide_type_t;

typedef struct vs_halide_dimension_t {
    int32_t min, extent, stride;
    uint32_t flags;
} vs_halide_dimension_t;

typedef struct vs_halide_buffer_t {
    uint64_t device;                                         /* opaque device handle, 0 = none */
    const struct halide_device_interface_t* device_interface;
    uint8_t* host;
    uint64_t flags;                                          /* halide_buffer_flag_host_dirty = 1, _device_dirty = 2 */
    vs_halide_type_t type;
    int32_t dimensions;
    vs_halide_dimension_t* dim;
    void* padding;
} vs_halide_buffer_t;

#ifdef __cplusplus
static_assert(sizeof(vs_halide_type_t) == 4 && sizeof(vs_halide_dimension_t) == 16, "halide_type_t / halide_dimension_t layout");
static_assert(sizeof(void*) != 8 || (sizeof(vs_halide_buffer_t) == 56 && offsetof(vs_halide_buffer_t, device_interface) == 8 &&
                                     offsetof(vs_halide_buffer_t, host) == 16 && offsetof(vs_halide_buffer_t, flags) == 24 &&
                                     offsetof(vs_halide_buffer_t, type) == 32 && offsetof(vs_halide_buffer_t, dimensions) == 36 &&
                                     offsetof(vs_halide_buffer_t, dim) == 40 && offsetof(vs_halide_buffer_t, padding) == 48),
              "halide_buffer_t layout (LP64)");
#endif
#endif /* VS_HALIDE_ABI_USE_HALIDE_RUNTIME */

/* the halide_error_code_t values these entry points return (HalideRuntime.h) */
enum {
    VS_HALIDE_OK = 0,
    VS_HALIDE_ERR_GENERIC = -1,                /* halide_error_code_generic_error: the HIP path failed (vs_last_error()) */
    VS_HALIDE_ERR_BAD_TYPE = -3,               /* halide_error_code_bad_type */
    VS_HALIDE_ERR_OUT_OF_BOUNDS = -4,          /* halide_error_code_access_out_of_bounds: extents that do not fit together */
    VS_HALIDE_ERR_CONSTRAINT = -8,             /* halide_error_code_constraint_violated: min != 0, stride of dim 0 != 1, padded plane */
    VS_HALIDE_ERR_BUFFER_NULL = -12,           /* halide_error_code_buffer_argument_is_null */
    VS_HALIDE_ERR_HOST_NULL = -34,             /* halide_error_code_host_is_null */
    VS_HALIDE_ERR_BAD_DIMENSIONS = -43         /* halide_error_code_bad_dimensions */
};

#ifdef __cplusplus
extern "C" {
#endif

/* generators.cpp:56-92 pyr_down: Input<Buffer<uint8_t>> input (2-D), Output<Buffer<uint8_t>> output (2-D); imgproc.cpp:112 */
int pyr_down(vs_halide_buffer_t* input, vs_halide_buffer_t* output);
/* generators.cpp:202-224 grad_xy: input u8 (2-D); output_x, output_y f32 (2-D); imgproc.cpp:140 */
int grad_xy(vs_halide_buffer_t* input, vs_halide_buffer_t* output_x, vs_halide_buffer_t* output_y);
/* generators.cpp:260-294 grad_argmax<tile_size> (CMakeLists.txt:212-253 instantiates 2, 4, .., 20): grad_x, grad_y f32 (2-D);
 * local_max_x, local_max_y u16 (3-D: tiles_x, tiles_y, 2); imgproc.cpp:176-194 */
int grad_argmax_2(vs_halide_buffer_t* grad_x, vs_halide_buffer_t* grad_y, vs_halide_buffer_t* local_max_x, vs_halide_buffer_t* local_max_y);
int grad_argmax_4(vs_halide_buffer_t* grad_x, vs_halide_buffer_t* grad_y, vs_halide_buffer_t* local_max_x, vs_halide_buffer_t* local_max_y);
int grad_argmax_6(vs_halide_buffer_t* grad_x, vs_halide_buffer_t* grad_y, vs_halide_buffer_t* local_max_x, vs_halide_buffer_t* local_max_y);
int grad_argmax_8(vs_halide_buffer_t* grad_x, vs_halide_buffer_t* grad_y, vs_halide_buffer_t* local_max_x, vs_halide_buffer_t* local_max_y);
int grad_argmax_10(vs_halide_buffer_t* grad_x, vs_halide_buffer_t* grad_y, vs_halide_buffer_t* local_max_x, vs_halide_buffer_t* local_max_y);
int grad_argmax_12(vs_halide_buffer_t* grad_x, vs_halide_buffer_t* grad_y, vs_halide_buffer_t* local_max_x, vs_halide_buffer_t* local_max_y);
int grad_argmax_14(vs_halide_buffer_t* grad_x, vs_halide_buffer_t* grad_y, vs_halide_buffer_t* local_max_x, vs_halide_buffer_t* local_max_y);
int grad_argmax_16(vs_halide_buffer_t* grad_x, vs_halide_buffer_t* grad_y, vs_halide_buffer_t* local_max_x, vs_halide_buffer_t* local_max_y);
int grad_argmax_18(vs_halide_buffer_t* grad_x, vs_halide_buffer_t* grad_y, vs_halide_buffer_t* local_max_x, vs_halide_buffer_t* local_max_y);
int grad_argmax_20(vs_halide_buffer_t* grad_x, vs_halide_buffer_t* grad_y, vs_halide_buffer_t* local_max_x, vs_halide_buffer_t* local_max_y);
/* generators.cpp:332-386 sparse_jac: grad_x, grad_y f32 (2-D); local_max_x, local_max_y u16 (3-D); output_x, output_y f32
 * (3-D: tiles_x, tiles_y, 4); imgproc.cpp:42 */
int sparse_jac(vs_halide_buffer_t* grad_x, vs_halide_buffer_t* grad_y, vs_halide_buffer_t* local_max_x, vs_halide_buffer_t* local_max_y,
               vs_halide_buffer_t* output_x, vs_halide_buffer_t* output_y);
/* generators.cpp:646-700 sparse_warpdiff: input_template, input_keyframe u8 (2-D); local_max u16 (3-D); A, B, TX, TY
 * (upper-left based: the caller converts, imgproc.cpp:98-103); output u16 (2-D: tiles_x, tiles_y); imgproc.cpp:94-104 */
int sparse_warpdiff(vs_halide_buffer_t* input_template, vs_halide_buffer_t* input_keyframe, vs_halide_buffer_t* local_max,
                    float A, float B, float TX, float TY, vs_halide_buffer_t* output);
/* generators.cpp:429-596 sparse_ica: input_template, input_keyframe u8 (2-D); selected_pixels_x / _y u16 (2-D: n, 2);
 * selected_jacobians_x / _y f32 (2-D: n, 4); A, B, TX, TY; output f64 (1-D: 4); imgproc.cpp:62-76 */
int sparse_ica(vs_halide_buffer_t* input_template, vs_halide_buffer_t* input_keyframe, vs_halide_buffer_t* selected_pixels_x,
               vs_halide_buffer_t* selected_pixels_y, vs_halide_buffer_t* selected_jacobians_x, vs_halide_buffer_t* selected_jacobians_y,
               float A, float B, float TX, float TY, vs_halide_buffer_t* output);
/* generators.cpp:126-164 image_warp: input u8 (2-D); A, B, TX, TY; output f32 (2-D); imgproc.cpp:131 */
int image_warp(vs_halide_buffer_t* input, float A, float B, float TX, float TY, vs_halide_buffer_t* output);

#ifdef __cplusplus
}
#endif
#endif /* VS_HALIDE_ABI_H */
