/*
 * vs_amd.h -- C ABI of libvs_amd.so: the MI355X (gfx950) alignment + warp engine.
 *
 * This is the drop-in boundary for the catid/video_stabilizer hot path.  Each entry point
 * names the reference interface it replaces (paths under the reference checkout).
 *
 * Conventions
 *   - extern "C", plain pointers and sizes.  Return: 0 = ok (or a documented positive value),
 *     negative = error; vs_last_error() returns a thread-local message.  No exceptions cross.
 *   - `mem` says where the caller's buffers live: VS_MEM_HOST (the library stages them through
 *     a pooled pinned mirror + device memory and synchronises: any alignment and pitch, pageable
 *     memory is fine; used by parity tests and one-off calls) or VS_MEM_DEVICE
 *     (device pointers; the call only enqueues work on `stream` and returns -- no sync).
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream).
 *   - Images are row-major; strides are in ELEMENTS.  "planar (tx,ty,c)" tables are laid out
 *     like the reference's Halide buffers: element (x,y,c) at c*tx*ty + y*tx + x.
 *   - The caller owns every buffer it passes.  Handles own their device memory and one HIP
 *     stream; a handle is single-threaded, distinct handles are independent.
 *   - There is NO CPU fallback: without a usable HIP device every compute entry point fails.
 */
#ifndef VS_AMD_H
#define VS_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VS_OK 0
#define VS_ERR_ARG (-1)
#define VS_ERR_HIP (-2)
#define VS_ERR_UNSUPPORTED (-3)
#define VS_ERR_STATE (-4)
#define VS_ERR_NOMEM (-5)        /* a host allocation failed (std::bad_alloc), or another C++ exception was stopped at this boundary: the call is
                                   abandoned under the error protocol of its family (engine calls: the running sequence ends, the handle stays usable) */

enum { VS_MEM_HOST = 0, VS_MEM_DEVICE = 1 };
/* Frame formats.  16-bit containers say how many bits the samples really use: the aligner derives its 8-bit luma with
 * gray >> (bits - 8) and the stabilizer's warp saturates at vs_format_max_value().  The reference itself is 8-bit only
 * (imgproc.cpp:207-209); BASELINE config 5 ("4K 10-bit BGR") is VS_FMT_BGR10.
 * (ABI 5 retired VS_FMT_BGR16 = 5, the first release's "10-bit luma, 65535 saturation" format: value 5 is an unknown format now --
 * declare the depth the samples really have.) */
enum { VS_FMT_GRAY8 = 0, VS_FMT_BGR8 = 1, VS_FMT_BGR10 = 2, VS_FMT_BGR12 = 3, VS_FMT_BGR16_FULL = 4 };
/* bits per sample the alignment luma assumes: 8, 10, 12 or 16; 0 for an unknown format */
int vs_format_bits(int format);
/* largest sample value the stabilizer's warp stores: 255, 1023, 4095, 65535; 0 for an unknown format */
int vs_format_max_value(int format);
/* VS_WARP_LANCZOS2: the reference sampler's sequence of fp32 roundings with no contraction (bit-identical to the CPU
 * restatement's VSO_WARP_LANCZOS2).
 * VS_WARP_LANCZOS2_FAST: opt-in, the CONTRACTED form of the same sampler -- every Horner step and every tap accumulation a
 * single fma (what the reference's own target string, which carries `fma` and no strict_float, lets its compiler emit), the
 * tap order and the correctly rounded division unchanged; bit-identical to the CPU restatement's VSO_WARP_LANCZOS2_CONTRACTED
 * (np.array_equal, every layout), 1.27x faster at 4K.
 * VS_WARP_LANCZOS2_SEP: opt-in, the SEPARABLE form -- the contracted form's weights and taps, summed rows first, then columns,
 * over the product of the two 1-D weight sums, one correctly rounded reciprocal for all channels (equal to generators.cpp:687-697
 * in real arithmetic; a reassociation inside the reference's own non-strict_float slack).  Bit-identical to the CPU restatement's
 * VSO_WARP_LANCZOS2_SEPARABLE; against the UN-contracted order: at most 1 LSB, >= 99.99 % of the samples identical (8- and 10-bit;
 * SURVEY 8(d)'s integer gate, tests/test_warp_gate_gpu.py).
 * VS_WARP_BILINEAR_CV: cv::warpAffine(INTER_LINEAR) as the reference's stabilizer calls it (stabilizer.cpp:97-99 ->
 * imgproc.cpp:446-484): OpenCV 4.x's classic FIXED-POINT path -- the matrix inverted in double, source coordinates in 1/32 pixel
 * (AB_BITS 10, INTER_BITS 5), 15-bit integer weights, (sum + 2^14) >> 15 for 8-bit samples (float weights and cvRound for 16-bit
 * containers).  Integer work: bit-identical to the CPU restatement's VSO_WARP_BILINEAR_CV, no tolerance.  "Parity unpinned
 * (OpenCV version)": the reference installs libopencv-dev unpinned and OpenCV is not in this image; the restatement follows the
 * published 4.5 / 4.6 source.  IN THIS MODE `t` IS THE TRANSFORM HANDED TO warpBySimilarityTransform -- the FORWARD map, which
 * cv::warpAffine inverts itself (every other mode takes the sampling map); integer output only (no _f32 form). */
enum { VS_WARP_LANCZOS2 = 0, VS_WARP_BILINEAR = 1, VS_WARP_LANCZOS2_FAST = 2, VS_WARP_LANCZOS2_SEP = 3, VS_WARP_BILINEAR_CV = 4 };
enum { VS_BORDER_CLAMP = 0, VS_BORDER_CONSTANT = 1 };
/* how the per-level "keep the best 80 %" subset is chosen (alignment.cpp:460-486) */
enum {
    VS_SELECT_STL_HOST = 0,   /* D2H + the host's std::nth_element, literally as the reference */
    VS_SELECT_DEVICE = 1,     /* on-device replica of libstdc++'s introselect: same set, same order (the default) */
    VS_SELECT_STABLE = 2      /* on the device under a documented, STL-independent rule: the tiles that are
                               * smallest by (abs_delta, tile index) -- ties on abs_delta go to the lower tile index -- in ascending
                               * tile order.  A set any conforming std::nth_element may produce; the survivors' ORDER (which the
                               * reference leaves to its STL, and which the fp64 sums follow) is fixed, so the transforms differ
                               * from the other two modes in the last bits -- and beyond, where the tied tiles differ.  No partition
                               * rounds (a histogram finds the cut, ballots place the survivors): a quarter of the selection
                               * time, one AlignNextFrame call in 0.20 ms instead of 0.245 at 1080p (0.29 instead of 0.38 at 4K).
                               * Bit-identical to the oracle's vso_select_smallest_stable / select rule 1. */
};

/* imgproc.hpp:40-46 SimilarityTransform (centre-based, double) */
typedef struct vs_transform { double A, B, TX, TY; } vs_transform;
typedef struct vs_point { double x, y; } vs_point;

/* alignment.hpp:5-41 VideoAlignerParams -- same fields, same defaults */
typedef struct vs_aligner_params {
    int    phase_correlate;            /* alignment.hpp:11: start TX,TY from cv::phaseCorrelate on pyramid level 2 */
    double phase_correlate_threshold;
    double threshold;
    float  smallest_fraction;
    int    max_iters;
    int    pyramid_min_width;
    int    pyramid_min_height;
    double max_displacement;
} vs_aligner_params;

/* stabilizer.hpp:13-30 VideoStabilizerParams (+ the two warp knobs this build defines) */
typedef struct vs_stabilizer_params {
    vs_aligner_params aligner;
    int    lag;
    int    smoother_memory;
    double lambda;
    int    enable_smoother;
    int    crop_pixels;
    double min_disp, max_disp;
    double min_decay, max_decay;
    int    warp_mode;     /* VS_WARP_*   (reference: cv::warpAffine INTER_LINEAR, imgproc.cpp:472) */
    int    warp_border;   /* VS_BORDER_* (reference: BORDER_CONSTANT black, imgproc.cpp:479-480) */
} vs_stabilizer_params;

const char* vs_last_error(void);
const char* vs_version(void);
/* ABI number of the structs and enums in this header.  It changes whenever a struct grows, an enum value moves or a default changes its
 * meaning (4: vs_align_info carries selected_x / selected_y / level_transform; 5: VS_FMT_BGR16 retired, VS_WARP_LANCZOS2_SEP = 3 and
 * VS_WARP_BILINEAR_CV = 4 added, vs_stabilizer_params_default's warp_mode is VS_WARP_BILINEAR_CV).  The engine writes sizeof(vs_align_info) bytes per frame
 * into caller arrays, so a caller built against another header must not go on: check vs_abi_version() == VS_ABI_VERSION once
 * after loading the library (the facade classes do, and throw).  vs_sizeof_align_info() is the size the LIBRARY was built with. */
#define VS_ABI_VERSION 5
int    vs_abi_version(void);
size_t vs_sizeof_align_info(void);
/* number of usable HIP devices (0 when there is none; never fails) */
int vs_device_count(void);

void vs_aligner_params_default(vs_aligner_params* p);
void vs_stabilizer_params_default(vs_stabilizer_params* p);

/* ------------------------------------------------------------------------------------------
 * Host-side scalar algebra (no device needed)
 * ------------------------------------------------------------------------------------------ */
/* SimilarityTransform::inverse / compose / warp / maxCornerDisplacement, imgproc.cpp:333-437 */
vs_transform vs_transform_inverse(const vs_transform* t);
vs_transform vs_transform_compose(const vs_transform* t1, const vs_transform* t2);   /* t1 then t2 */
vs_point     vs_transform_warp(const vs_transform* t, vs_point p);
vs_point     vs_transform_warp_center(const vs_transform* t, vs_point p, double cx, double cy);
double       vs_transform_max_corner_displacement(const vs_transform* t, double width, double height);
/* GradArgMax's tile-size rule, imgproc.cpp:151-162 */
int vs_tile_size(int w, int h);
/* centre-based double transform -> the float, upper-left based kernel arguments.
 * sparse: imgproc.cpp:69-75 / 98-103 (centre w/2,h/2).  warp: imgproc.cpp:125-131 (centre (w-1)/2,(h-1)/2) */
void vs_ul_params_sparse(const vs_transform* t, int w, int h, float out4[4]);
void vs_ul_params_warp(const vs_transform* t, int w, int h, float out4[4]);
/* VS_WARP_BILINEAR_CV: the 2x3 matrix of warpBySimilarityTransform(t) for a w x h frame (imgproc.cpp:457-466), inverted the way
 * cv::warpAffine inverts a matrix given without WARP_INVERSE_MAP (double precision, OpenCV's operation order): row-major
 * {M0, M1, M2; M3, M4, M5}, the map from an output pixel to its source position. */
void vs_cv_inverse_matrix(const vs_transform* t, int w, int h, double out6[6]);
/* L1SmootherCenter, smoother.hpp:10-30 / smoother.cpp:67-127 */
typedef struct vs_smoother vs_smoother;
vs_smoother* vs_smoother_create(int lag_behind, int lag_ahead, double lambda);
void vs_smoother_destroy(vs_smoother* s);
int  vs_smoother_update(vs_smoother* s, const vs_transform* meas, vs_transform* out_finalized); /* 1 = finalized */
void vs_tvl1_smooth(const double* data, int n, double lambda, int iterations, double* out);   /* smoother.cpp:18-65 */

/* ------------------------------------------------------------------------------------------
 * Kernel level: one entry point per Halide AOT function called from imgproc.cpp
 * ------------------------------------------------------------------------------------------ */
/* int pyr_down(in, out)                                   imgproc.cpp:112, generators.cpp:56-92 */
int vs_pyr_down(const uint8_t* in, int w, int h, int in_stride,
                uint8_t* out, int ow, int oh, int out_stride, int mem, void* stream);
/* int grad_xy(in, gx, gy)                                 imgproc.cpp:140, generators.cpp:202-224
 * gx, gy dense (w*h) */
int vs_grad_xy(const uint8_t* in, int w, int h, int stride, float* gx, float* gy, int mem, void* stream);
/* int grad_argmax_<ts>(gx, gy, local_max_x, local_max_y)  imgproc.cpp:174-195, generators.cpp:260-294
 * tile_size 1..64; outputs planar (w/ts, h/ts, 2) u16 */
int vs_grad_argmax(const float* gx, const float* gy, int w, int h, int tile_size,
                   uint16_t* local_max_x, uint16_t* local_max_y, int mem, void* stream);
/* int sparse_jac(gx, gy, lmx, lmy, out_x, out_y)          imgproc.cpp:42, generators.cpp:332-386
 * outputs planar (tx,ty,4) f32 */
int vs_sparse_jac(const float* gx, const float* gy, int w, int h,
                  const uint16_t* local_max_x, const uint16_t* local_max_y, int tx, int ty,
                  float* out_x, float* out_y, int mem, void* stream);
/* Fused keyframe pass (what the engine runs): grad_xy + grad_argmax + sparse_jac straight from
 * the u8 image, never materialising the gradient planes.  Same outputs, bit for bit, as the
 * three calls above (alignment.cpp:237-276). */
int vs_keyframe_fused(const uint8_t* in, int w, int h, int stride, int tile_size,
                      uint16_t* local_max_x, uint16_t* local_max_y, float* jac_x, float* jac_y,
                      int mem, void* stream);
/* int sparse_warpdiff(tmpl, key, local_max, A, B, TX, TY, out)   imgproc.cpp:94-104, generators.cpp:646-700 */
int vs_sparse_warpdiff(const uint8_t* tmpl, const uint8_t* key, int w, int h, int stride,
                       const uint16_t* local_max, int tx, int ty,
                       float A, float B, float TX, float TY, uint16_t* out, int mem, void* stream);
/* int sparse_ica(tmpl, key, selx, sely, jacx, jacy, A, B, TX, TY, out)  imgproc.cpp:62-76, generators.cpp:429-596
 * selx/sely planar (n,2) u16; jacx/jacy planar (n,4) f32; out = 4 doubles */
int vs_sparse_ica(const uint8_t* tmpl, const uint8_t* key, int w, int h, int stride,
                  const uint16_t* selx, int nx, const uint16_t* sely, int ny,
                  const float* jacx, const float* jacy,
                  float A, float B, float TX, float TY, double* out4, int mem, void* stream);
/* The keep-best-fraction selection of alignment.cpp:435-492 as a device op, for n_arrays independent
 * warpdiff tables (each tx*ty u16, row-major): flatten, std::nth_element on abs_delta, keep the first
 * size_t(tx*ty*fraction).  out_idx: n_arrays x (tx*ty) int32, the first `count` of each row are the
 * surviving tile indices (tile_y*tx+tile_x) in the exact order libstdc++'s nth_element leaves them
 * (an on-device replica of its introselect).  status[a] = 1 if libstdc++ would have taken its
 * heap-select fallback for array a (not replicated; callers then use the host).  Returns count. */
int vs_select_smallest(const uint16_t* warpdiff, int n_arrays, int tx, int ty, float fraction,
                       int32_t* out_idx, int32_t* status, int mem, void* stream);
/* The same step under VS_SELECT_STABLE's rule: the first `count` entries of each row are the tiles that are
 * smallest by (abs_delta, tile index), in ascending tile order -- the oracle's vso_select_smallest_stable.  Returns count. */
int vs_select_smallest_stable(const uint16_t* warpdiff, int n_arrays, int tx, int ty, float fraction,
                              int32_t* out_idx, int mem, void* stream);
/* cv::phaseCorrelate(a, b, cv::noArray(), &response) as VideoAligner calls it (alignment.cpp:372-374) on the CV_32F copy
 * of pyramid level 2 (alignment.cpp:225-229): two w x h u8 images -> result[3] = {shift.x, shift.y, response} in host
 * memory (the call synchronises).  Images are zero-padded to vs_optimal_dft_size (cv::getOptimalDFTSize: 2^a 3^b 5^c);
 * surface: NULL, or M*N floats (M, N = padded h, w; in `mem`) receiving the unshifted, unscaled correlation surface.
 * OpenCV is not part of the reference tree: the transform specification is this build's (oracle/vs_phase.cpp). */
int vs_phase_correlate(const uint8_t* a, const uint8_t* b, int w, int h, int stride, int mem, void* stream, float* surface,
                       double* result);
int vs_optimal_dft_size(int n);
/* int image_warp(in, A, B, TX, TY, out)                   imgproc.cpp:131, generators.cpp:126-164 */
int vs_image_warp(const uint8_t* in, int w, int h, int stride,
                  float A, float B, float TX, float TY, float* out, int ow, int oh, int mem, void* stream);
/* Streams passed to the bgr_image_warp entry points: the library keeps one event per in-flight call on the caller's stream (its
 * parameter ring) until a later call on the same thread retires it.  Before DESTROYING such a stream call vs_stream_retire(stream):
 * it waits for the library's work on that stream and drops every reference to it.  (Handles do this for their own streams.) */
int vs_stream_retire(void* stream);
/* bgr_image_warp: the full-frame colour warp (replaces warpBySimilarityTransform's cv::warpAffine,
 * imgproc.cpp:446-484 / stabilizer.cpp:97-99; the generator itself is absent from the reference,
 * schedules/bgr_image_warp.schedule.h is its orphan -- SURVEY D2).  `t` is the output->input
 * sampling map, centre-based about ((w-1)/2,(h-1)/2) as ImageWarp (imgproc.cpp:125-131).
 * src/dst interleaved with `channels` (1..4) per pixel; bits 8 (uint8_t) or 16 (uint16_t);
 * store rule floor(v+0.5) saturated to [0,max_value].  n_frames >= 1 frames are processed in one
 * launch: frame i at src + i*src_frame_stride (elements), transform t[i]. */
int vs_bgr_image_warp(const void* src, int w, int h, int src_stride, int channels, int bits,
                      const vs_transform* t, int mode, int border, int max_value,
                      void* dst, int dst_stride, int mem, void* stream);
int vs_bgr_image_warp_batch(const void* src, size_t src_frame_stride, int n_frames,
                            int w, int h, int src_stride, int channels, int bits,
                            const vs_transform* t /* host array, n_frames */, int mode, int border, int max_value,
                            void* dst, size_t dst_frame_stride, int dst_stride, int mem, void* stream);
/* The same, but only the window [roi_x, roi_x+roi_w) x [roi_y, roi_y+roi_h) of every output frame is computed and
 * stored (dst holds roi_w x roi_h pixels per frame, row stride dst_stride, frame stride dst_frame_stride): bit-identical
 * to cropping the full warp.  This is how the stabilizer applies crop_pixels (stabilizer.cpp:102-109) without warping the
 * margin or copying the frame. */
int vs_bgr_image_warp_roi_batch(const void* src, size_t src_frame_stride, int n_frames, int w, int h, int src_stride,
                                int channels, int bits, const vs_transform* t, int mode, int border, int max_value,
                                int roi_x, int roi_y, int roi_w, int roi_h,
                                void* dst, size_t dst_frame_stride, int dst_stride, int mem, void* stream);
/* same sampling, float output (typed like image_warp); dst interleaved f32 */
int vs_bgr_image_warp_f32(const void* src, int w, int h, int src_stride, int channels, int bits,
                          const vs_transform* t, int mode, int border,
                          float* dst, int dst_stride, int mem, void* stream);
/* cv::cvtColor(BGR2GRAY) stand-in (alignment.cpp:212): (B*3735+G*19235+R*9798+16384)>>15, then >>shift_to_8 */
int vs_bgr_to_gray(const void* src, int w, int h, int src_stride, int bits, int shift_to_8,
                   uint8_t* dst, int dst_stride, int mem, void* stream);

/* Test hook, not part of the reference's surface: fault injection for the library's own device / pinned-host allocations.
 * vs_test_fail_alloc(k), k > 0: the k-th allocation the library makes from now on (any handle, any thread) fails once with
 * out-of-memory, and the call it belongs to returns VS_ERR_HIP; k < 0: the |k|-th allocation THROWS std::bad_alloc instead -- a host
 * allocation failing at that point of the call, which must come back as VS_ERR_NOMEM (no exception crosses this boundary); k = 0
 * disarms.  Returns the number of allocations made since the
 * previous call of this function.  The environment variable VS_TEST_FAIL_ALLOC=k (honoured only together with VS_TEST_HOOKS=1) arms it at load time for programs that cannot
 * call it.  tests/test_alloc_failure_gpu.py walks k over every allocation of the engine-level calls. */
/* VS_TEST_POISON_ALLOC=<byte> in the environment (read once; honoured only together with VS_TEST_HOOKS=1): every fresh device allocation starts filled with that byte, so that a result
 * which depends on memory the library never wrote changes with the byte (tests/test_uninitialised_memory_gpu.py). */
int vs_test_fail_alloc(int k);

/* Debug build only (tools/build_variant.sh bounds: -DVS_DEBUG_BOUNDS; the reference's "bounds asserts in debug kernels").
 * In that build the LDS / scratch arrays of the selection, gather, exchange, warp-tile and FFT-line code are indexed through a
 * checked accessor: an out-of-range index is recorded (first one per source file: site id, index, limit, workgroup, thread) and
 * the access is redirected to element 0 -- reported, never executed, nothing traps.  vs_debug_bounds_check() synchronises the
 * device, returns the number of violations since the previous call (0 = clean) with the first one described in vs_last_error(),
 * and clears the record.  vs_debug_bounds_selftest() commits one violation on purpose (site 900, index 11 of 8) and returns 202.
 * In the regular library both return VS_ERR_UNSUPPORTED, and the kernels carry no checks (identical instruction streams). */
int vs_debug_bounds_check(void);
int vs_debug_bounds_selftest(void);

/* Profiling aid, not part of the reference's surface: device-to-device copy of floor(bytes/12)*12 bytes
 * with 12-byte accesses per lane, used to calibrate rocprofv3's FETCH_SIZE / WRITE_SIZE for the warp
 * kernel's access width (tools/calibrate_counters.py). */
int vs_calib_copy12(const void* src_dev, void* dst_dev, size_t bytes, void* stream);
/* Profiling aid: the shader clock as a kernel sees it.  Launches one VALU-bound probe kernel (2048 workgroups of dependent fmas,
 * ~0.3 ms) on `stream` of the current device and returns delta s_memtime / delta s_memrealtime x 100 MHz of its first wave in
 * *shader_mhz (MI355X_MICROARCH.md: the pair of counters that shows DVFS give-back); synchronises the stream.  bench.py issues it
 * directly behind its timed loop, so the figure is the clock the timed kernels ran at. */
int vs_shader_clock_probe(void* stream, double* shader_mhz);

/* ------------------------------------------------------------------------------------------
 * Engine level: VideoAligner (alignment.hpp:51-99) and VideoStabilizer (stabilizer.hpp:32-56)
 * ------------------------------------------------------------------------------------------ */
typedef struct vs_aligner vs_aligner;
/* per-frame detail of the last align call(s), for parity tests and the per-stage report */
typedef struct vs_align_info {
    int32_t status;          /* 1 aligned, 0 not aligned */
    int32_t fail_reason;     /* 0 ok, 1 first frame, 2 max iterations, 3 over displacement */
    int32_t fail_level;
    int32_t levels;
    int32_t iterations[16];
    double  condition[16];
    double  phase_dx, phase_dy, phase_response;   /* cv::phaseCorrelate result of the pair (phase_correlate only, else 0) */
    /* the custom metrics of the reference's PerformanceMetrics (alignment.cpp:489-490 "SelectedPointsX_/Y_<level>") and the
     * estimate each level ended on (centre-based, that level's pixels, before the x2 of TX,TY at :683-687): set for every
     * level the pair reached; a level that ran out of iterations (fail_reason 2) leaves its transform 0 */
    int32_t selected_x[16], selected_y[16];
    vs_transform level_transform[16];
} vs_align_info;

vs_aligner* vs_aligner_create(const vs_aligner_params* params /* NULL = defaults */, int device);
void vs_aligner_destroy(vs_aligner* a);
/* Error protocol of the engine-level calls: a call that returns < 0 (a refused allocation: VS_ERR_HIP, an unsupported size, ...)
 * leaves the handle consistent and usable -- no buffer is lost or freed twice -- and ENDS the running sequence: the next frame
 * is the first frame of a new sequence, exactly what a fresh handle would make of it.  (The reference: a failed kernel call
 * returns false and sets LastWidth = -1, so the next AlignNextFrame re-initialises, alignment.cpp:357-367.) */
/* New handles start in VS_SELECT_DEVICE, or in the mode the environment variable VS_SELECT_MODE=0|1|2 names (read once per
 * process, reported once on stderr when it takes effect: the modes differ in the last bits of the transforms); set_select_mode
 * overrides either.  vs_aligner_get_select_mode returns the mode in force (or VS_ERR_ARG). */
int  vs_aligner_set_select_mode(vs_aligner* a, int select_mode);
int  vs_aligner_get_select_mode(const vs_aligner* a);
/* Which build of the per-pair solver kernel a batch (>= 32 frame pairs, levels of <= 26000 tiles) runs through.  The
 * results are bit-identical either way.
 *   VS_BATCH_EXCLUSIVE (default)  one 512-thread workgroup per pair, a whole CU each: fastest when nothing else is running.
 *   VS_BATCH_SHARED               the small-footprint build (256 threads, <= 128 VGPRs, <= 36 KB LDS: the footprint of one
 *                                 bgr_image_warp workgroup; levels whose selection arrays do not fit select on global scratch).  For callers that keep the GPU busy on another stream while the
 *                                 alignment runs -- typically the warp of the previous clip (stabilizer.cpp:97-99 after
 *                                 :19): the pairs move into CUs as the other grid's workgroups retire instead of waiting
 *                                 for whole CUs to drain. */
enum { VS_BATCH_EXCLUSIVE = 0, VS_BATCH_SHARED = 1 };
int  vs_aligner_set_batch_mode(vs_aligner* a, int batch_mode);
/* forget the sequence: the next frame is treated as the first frame of a new clip (device memory is kept) */
int  vs_aligner_reset(vs_aligner* a);
/* Stream ordering of the engine-level calls.  A handle works on its own non-blocking stream (vs_aligner_stream), which
 * is NOT ordered against the caller's streams, the NULL stream included.  With VS_MEM_DEVICE the frames must therefore be
 * complete before an align / process call -- either the producer stream has been synchronised, or
 * vs_aligner_wait_stream(a, producer_stream) was called after the last producer enqueue: everything enqueued on
 * producer_stream up to that point then happens before whatever the handle enqueues afterwards.  In the other direction
 * nothing is needed: align / process calls return after their device work has finished (they hand results to the host). */
void* vs_aligner_stream(const vs_aligner* a);
int   vs_aligner_wait_stream(vs_aligner* a, void* producer_stream);
/* VideoAligner::AlignNextFrame (alignment.hpp:55-58, alignment.cpp:334-704).
 * returns 1 aligned / 0 not aligned (first frame, no convergence, over displacement) / <0 error.  When not aligned,
 * *out holds what the reference leaves in `transform`: identity for the first frame, else the estimate reached when the
 * level gave up (VideoStabilizer passes it on to the smoother regardless, stabilizer.cpp:18-44).
 * `params` may change per call like the reference's third argument (NULL = the creation params). */
int  vs_aligner_align_next(vs_aligner* a, const void* frame, int w, int h, int stride, int format, int mem,
                           const vs_aligner_params* params, vs_transform* out);
/* Batched form: exactly the results of n successive vs_aligner_align_next calls on this handle
 * (state carries over between calls), but every stage runs as one launch over all frames /
 * frame pairs.  frames: n frames, frame i at frames + i*frame_stride (elements).  out[n],
 * status[n] (1/0 per frame).  Returns the number of aligned frames, or <0 on error. */
int  vs_aligner_align_batch(vs_aligner* a, const void* frames, size_t frame_stride, int n,
                            int w, int h, int stride, int format, int mem,
                            const vs_aligner_params* params, vs_transform* out, int32_t* status);
/* Many independent clips at once: n_clips clips of frames_per_clip frames, back to back (clip c, frame k at index
 * c*frames_per_clip + k).  Results = every clip aligned by its own fresh VideoAligner (frame 0 of each clip: status 0,
 * fail_reason 1), computed together so that short clips still fill the GPU.  Resets the handle's running sequence. */
int  vs_aligner_align_clips(vs_aligner* a, const void* frames, size_t frame_stride, int n_clips, int frames_per_clip,
                            int w, int h, int stride, int format, int mem,
                            const vs_aligner_params* params, vs_transform* out, int32_t* status);
/* detail for frame i of the most recent align_next (i = 0) / align_batch call */
int  vs_aligner_get_info(const vs_aligner* a, int i, vs_align_info* info);
/* device pointers / dims of internal per-level state of the most recent call (parity tests) */
int  vs_aligner_level_dims(const vs_aligner* a, int level, int* w, int* h, int* tiles_x, int* tiles_y, int* tile_size);
/* copies out (to host) level images / keypoint tables of frame i of the most recent call */
int  vs_aligner_read_level_image(const vs_aligner* a, int i, int level, uint8_t* out);
int  vs_aligner_read_level_argmax(const vs_aligner* a, int i, int level, int set, uint16_t* out);
int  vs_aligner_read_level_jacobian(const vs_aligner* a, int i, int level, int set, float* out);

/* Opt-in per-stage timing (the reference's compiled-out PerformanceMetrics / TIME_FUNCTION,
 * alignment.cpp:10-147).  Device stages are bracketed with hipEvents on the handle's stream, the
 * selection stage is host wall-clock.  Values accumulate until reset. */
enum {
    VS_STAGE_INGEST = 0,     /* H2D (host frames) + BGR->gray / gray copy   "ConvertToBGR"       */
    VS_STAGE_PYR_DOWN = 1,   /* all pyr_down launches                       "PyrDown_i"          */
    VS_STAGE_KEYFRAME = 2,   /* fused grad/argmax/jacobian launches         "GradXY_i".."SparseJacobian_i" */
    VS_STAGE_WARPDIFF = 3,   /*                                             "SparseWarpDiff_X/Y_i" */
    VS_STAGE_SELECT = 4,     /* keep-best-80% incl. its copies              "NthElement_i"       */
    VS_STAGE_GATHER = 5,     /*                                             "JacobianSetup_i"    */
    VS_STAGE_GN = 6,         /* Hessian + solve + all iterations            "ICAIteration_i_iter" */
    VS_STAGE_PHASE = 7,      /* level-2 spectra + cross-power/inverse/peak  "PhaseCorrelation"   */
    VS_STAGE_COUNT = 8
};
typedef struct vs_stage_timings {
    double ms[VS_STAGE_COUNT];        /* accumulated milliseconds per stage */
    int64_t launches[VS_STAGE_COUNT]; /* kernel launches per stage */
    int64_t frames;                   /* frames processed while timing was on */
    int64_t gn_iterations;            /* Gauss-Newton iterations summed over pairs and levels */
} vs_stage_timings;
int  vs_aligner_enable_timing(vs_aligner* a, int enable);   /* also resets the accumulators */
int  vs_aligner_get_timings(vs_aligner* a, vs_stage_timings* out);

typedef struct vs_stabilizer vs_stabilizer;
vs_stabilizer* vs_stabilizer_create(const vs_stabilizer_params* params /* NULL = defaults */, int device);
void vs_stabilizer_destroy(vs_stabilizer* s);
/* VideoStabilizer::processFrame (stabilizer.hpp:39, stabilizer.cpp:9-117).  frame: interleaved BGR
 * u8 (VS_FMT_BGR8) or u16 (VS_FMT_BGR10 / BGR12 / BGR16_FULL).  out: (w-2*crop)*(h-2*crop)*3 elements, dense.
 * returns 1 when an output frame was written (0 for the first `lag` frames), <0 on error. */
int  vs_stabilizer_process(vs_stabilizer* s, const void* frame, int w, int h, int stride, int format, int mem,
                           void* out, int* out_w, int* out_h);
/* Batched form: exactly n successive vs_stabilizer_process calls (one batched alignment, the reference's scalar
 * bookkeeping in order on the host, batched warps).  frames: frame i at frames + i*frame_stride (elements);
 * has_output[i] = 1 when input frame i produced an output, written dense at out + i*out_frame_stride (elements,
 * >= out_w*out_h*3).  Returns the number of output frames, or <0. */
int  vs_stabilizer_process_batch(vs_stabilizer* s, const void* frames, size_t frame_stride, int n, int w, int h, int stride,
                                 int format, int mem, void* out, size_t out_frame_stride, int32_t* has_output,
                                 int* out_w, int* out_h);
/* Many independent clips at once: n_clips clips of frames_per_clip frames back to back (clip c, frame k at index
 * c*frames_per_clip + k, outputs at the same index).  Results = every clip through its own fresh VideoStabilizer, with the
 * alignment and the warps of all clips batched together.  The handle is reset before the first and after the last clip. */
int  vs_stabilizer_process_clips(vs_stabilizer* s, const void* frames, size_t frame_stride, int n_clips, int frames_per_clip,
                                 int w, int h, int stride, int format, int mem, void* out, size_t out_frame_stride,
                                 int32_t* has_output, int* out_w, int* out_h);
int  vs_stabilizer_reset(vs_stabilizer* s);   /* start a new clip; device buffers are kept */
/* stream ordering for VS_MEM_DEVICE frames: see vs_aligner_wait_stream */
void* vs_stabilizer_stream(const vs_stabilizer* s);
int   vs_stabilizer_wait_stream(vs_stabilizer* s, void* producer_stream);
/* the selection rule of the stabilizer's aligner (VS_SELECT_*, see vs_aligner_set_select_mode); takes effect with the next frame */
int   vs_stabilizer_set_select_mode(vs_stabilizer* s, int mode);
int   vs_stabilizer_get_select_mode(const vs_stabilizer* s);
void vs_stabilizer_state(const vs_stabilizer* s, vs_transform* last_meas, vs_transform* accum, int* last_success);

#ifdef __cplusplus
}
#endif
#endif
