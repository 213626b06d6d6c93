/*
 * l3d.h -- C ABI of the MI355X (gfx950) stereo-depth + laser centre-line library
 * (libl3d_hip.so).  Every entry point is extern "C", takes plain pointers and
 * sizes, returns 0 on success or a negative l3d_status, and never throws; the
 * text of the last error on the calling thread is available from
 * l3d_last_error().
 *
 * The reference project has no FFI of its own: its hot path is a sequence of
 * calls into the OpenCV / NumPy wheels.  Each entry point below therefore cites
 * the reference call site (file:line under the reference repository) whose
 * third-party call it replaces.
 *
 * Conventions
 *   - "host" entry points take caller-owned host buffers and copy in/out.
 *   - "_device" entry points take device pointers that are already resident in
 *     HBM ([batch][H][W] contiguous) and enqueue on exactly the `stream` given
 *     (a hipStream_t passed as void*; NULL = HIP's default stream), so they are
 *     ordered with the caller's own work on that stream.  They do not
 *     synchronise; the host variants run on the handle's stream and do.
 *   - A handle owns all scratch memory it needs (allocated at create time for
 *     `max_batch` frames) and one stream.  A handle is not thread-safe, and two
 *     calls on one handle must not overlap in time (they share its scratch);
 *     distinct handles are independent.
 *   - Images are single-channel u8 unless stated; disparities are int16 fixed
 *     point with 4 fractional bits, exactly as cv2.StereoSGBM.compute returns.
 */
#ifndef L3D_H
#define L3D_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
    L3D_OK = 0,
    L3D_ERR_INVALID_ARG = -1,
    L3D_ERR_UNSUPPORTED = -2,
    L3D_ERR_HIP = -3,
    L3D_ERR_NO_DEVICE = -4,
    L3D_ERR_CAPACITY = -5
} l3d_status;

const char *l3d_last_error(void);
const char *l3d_version(void);
/* number of visible HIP devices (0 and L3D_ERR_NO_DEVICE when there is none) */
int l3d_device_count(int *n);

/* ---------------------------------------------------------------------------
 * Semi-global block matching, cv2.STEREO_SGBM_MODE_SGBM_3WAY semantics.
 * Field names and meaning are cv2.StereoSGBM_create's keyword arguments
 * (camera/single_usb_stereo_camera.py:252-274).
 * ------------------------------------------------------------------------- */
typedef struct {
    int minDisparity;
    int numDisparities;    /* multiple of 16 in [16, 256] */
    int blockSize;         /* odd, 1..21 */
    int P1, P2;
    int disp12MaxDiff;
    int preFilterCap;
    int uniquenessRatio;
    int speckleWindowSize;
    int speckleRange;
    int mode;              /* 2 = cv2.STEREO_SGBM_MODE_SGBM_3WAY (what the reference passes, :273); 0 = MODE_SGBM (5 paths),
                              1 = MODE_HH (8 paths); 3 = MODE_HH4 is not built.
                              Modes 0 and 1 are not on the reference's path; their parity with OpenCV is unpinned (checked
                              against the CPU restatement of computeDisparitySGBM only).  They keep the aggregated-cost and
                              the summed-cost volume of whole images in device memory:
                              2 * max_batch * H * (W - numDisparities) * Dp * 2 bytes (Dp = numDisparities rounded up to what
                              the lane layout holds, at most + 16), against one volume of H + 4 rows for mode 2, and their
                              row passes read the volume with a stride (about 4-5 x slower than mode 2). */
} l3d_sgbm_params;

typedef struct l3d_sgbm l3d_sgbm;

/* replaces cv2.StereoSGBM_create(...) (camera/single_usb_stereo_camera.py:252)
 * and cv2.ximgproc.createRightMatcher's product (:277) */
int l3d_sgbm_create(const l3d_sgbm_params *params, int width, int height, int max_batch, int device,
                    l3d_sgbm **out);
void l3d_sgbm_destroy(l3d_sgbm *h);

/* replaces stereo_matcher.compute(left_gray, right_gray)
 * (camera/single_usb_stereo_camera.py:324-325).  Host buffers; row_stride and
 * img_stride in bytes; disp is [batch][H][W] int16, contiguous. */
int l3d_sgbm_compute(l3d_sgbm *h, const uint8_t *left, const uint8_t *right, int batch,
                     ptrdiff_t row_stride, ptrdiff_t img_stride, int16_t *disp);
/* same, device-resident operands.  The work is ordered on `stream` as seen from outside: it starts when `stream`
 * reaches the call and `stream` continues when the result is complete.  From 8 frames on, a batch runs as two
 * halves on two streams owned by the handle, forked from and joined to `stream` by events (environment
 * L3D_SGBM_SPLIT=0 keeps everything on `stream`).  One call at a time per handle. */
int l3d_sgbm_compute_device(l3d_sgbm *h, const uint8_t *d_left, const uint8_t *d_right, int batch,
                            int16_t *d_disp, void *stream);

/* The frame-pair step of the reference's get_frames / compute_depth in one call, device-resident
 * (camera/single_usb_stereo_camera.py:319-346, readme.md:367): BGR -> gray of both views, the matcher, /16,
 * reprojectImageTo3D Z and the validity masks.  d_left_bgr, d_right_bgr [batch][H][W][3]; d_depth [batch][H][W]
 * float; d_disp [batch][H][W] int16 or NULL (the disparity then stays in the handle); Q16_host as for
 * l3d_depth_from_disp16_device.  Results equal l3d_bgr2gray_device x2 + l3d_sgbm_compute_device +
 * l3d_depth_from_disp16_device; here the matcher's first kernel converts the colour frames on the way in (no gray
 * plane is written) and the reprojection runs on the handle's two streams with its half of the batch. */
int l3d_sgbm_compute_depth_device(l3d_sgbm *h, const uint8_t *d_left_bgr, const uint8_t *d_right_bgr, int batch,
                                  const double *Q16_host, int16_t *d_disp, float *d_depth, void *stream);

/* The tail of StereoSGBM::compute on its own: 3x3 median (replicated border) of
 * raw [batch][H][W] int16 disparity maps, then the speckle filter with the
 * handle's speckleWindowSize / speckleRange and newVal = 16*(minDisparity-1)
 * (skipped when speckleWindowSize <= 0).  Host pointers.  Same call site as
 * l3d_sgbm_compute (camera/single_usb_stereo_camera.py:324); exported so that
 * the post-filters can be driven with hand-made maps. */
int l3d_sgbm_postfilter(l3d_sgbm *h, const int16_t *raw, int batch, int16_t *disp);

/* Stage buffers of the most recent compute on this handle, copied to host
 * after synchronising (for stage-level parity checks):
 *   0 = aggregated BT cost C      [batch][H][W1][D] int16
 *   1 = top->down path L          [batch][H][W1][D] int16
 *   2 = disparity before median   [batch][H][W]     int16
 * bytes must not exceed the stage's size for the last batch. */
int l3d_sgbm_fetch_stage(l3d_sgbm *h, int stage, void *host_dst, size_t bytes);
/* Per-kernel device timing of the compute calls on this handle, taken with HIP
 * events recorded on the stream the kernels are launched on.  profile(h, 1)
 * switches recording on and clears the totals, profile(h, 0) switches it off;
 * profile_read waits for outstanding calls and returns the accumulated
 * milliseconds per kernel (profile_kernel_count entries, named by
 * profile_kernel_name) and the number of compute calls covered.  No reference
 * counterpart: measurement only. */
int l3d_sgbm_profile(l3d_sgbm *h, int enable);
int l3d_sgbm_profile_read(l3d_sgbm *h, double *total_ms, int n, long *calls);
const char *l3d_sgbm_profile_kernel_name(int k);
int l3d_sgbm_profile_kernel_count(void);
/* out8 = {minX1, W1, stripe_sz, stripe_overlap, invalid16, ftzero, P1, P2} */
int l3d_sgbm_geometry(const l3d_sgbm_params *params, int width, int height, int *out8);

/* ---------------------------------------------------------------------------
 * cv2.ximgproc.DisparityWLSFilter (confidence + fast global smoother).
 * ------------------------------------------------------------------------- */
typedef struct {
    double lambda;        /* wls_filter.setLambda  (camera/...:281), reference 8000 */
    double sigma_color;   /* wls_filter.setSigmaColor (:282), reference 1.5 */
    int min_disparity;    /* of the left matcher */
    int num_disparities;
    int lrc_thresh;       /* 24 */
    int disc_radius;      /* ceil(0.5 * blockSize) */
    float disc_rolloff;   /* 0.001 */
    int fgs_iters;        /* 3 */
    float fgs_attenuation;/* 0.25 */
} l3d_wls_params;

typedef struct l3d_wls l3d_wls;

/* replaces cv2.ximgproc.createDisparityWLSFilter(left_matcher) + setLambda +
 * setSigmaColor (camera/single_usb_stereo_camera.py:280-282) */
int l3d_wls_create(const l3d_wls_params *params, int width, int height, int max_batch, int device,
                   l3d_wls **out);
void l3d_wls_destroy(l3d_wls *h);
/* replaces wls_filter.filter(dL, left_gray, disparity_map_right=dR) (:328-332) */
int l3d_wls_filter(l3d_wls *h, const int16_t *disp_left, const int16_t *disp_right, const uint8_t *guide,
                   int batch, int16_t *out);
int l3d_wls_filter_device(l3d_wls *h, const int16_t *d_disp_left, const int16_t *d_disp_right,
                          const uint8_t *d_guide, int batch, int16_t *d_out, void *stream);
/* confidence map (x255) of the most recent filter call, [batch][H][W] float */
int l3d_wls_fetch_confidence(l3d_wls *h, float *host_dst, size_t bytes);

/* ---------------------------------------------------------------------------
 * Colour conversion + disparity -> depth.
 * ------------------------------------------------------------------------- */
/* replaces cv2.cvtColor(img, cv2.COLOR_BGR2GRAY)
 * (camera/single_usb_stereo_camera.py:320-321).  [batch][H][W][3] -> [batch][H][W] */
int l3d_bgr2gray_device(const uint8_t *d_bgr, int width, int height, int batch, uint8_t *d_gray,
                        int device, void *stream);
int l3d_bgr2gray(const uint8_t *bgr, int width, int height, int batch, uint8_t *gray, int device);

/* replaces `.astype(float32)/16`, cv2.reprojectImageTo3D(disparity, Q)[:,:,2] and
 * the three validity masks (camera/single_usb_stereo_camera.py:335-346); with
 * Q == NULL the uncalibrated fallback of :349-357.  depth [batch][H][W] float;
 * xyz (may be NULL, needs Q) [batch][H][W][3] float = unmasked
 * reprojectImageTo3D output. */
int l3d_depth_from_disp16_device(const int16_t *d_disp16, const double *Q16_host, int width, int height,
                                 int batch, float *d_depth, float *d_xyz, int device, void *stream);
int l3d_depth_from_disp16(const int16_t *disp16, const double *Q16, int width, int height, int batch,
                          float *depth, float *xyz, int device);

/* ---------------------------------------------------------------------------
 * Steger ridge extractor (FastStegerExtractor, core/laser_extractor.py:103-261).
 * ------------------------------------------------------------------------- */
typedef struct l3d_steger l3d_steger;

/* replaces FastStegerExtractor.__init__ (core/laser_extractor.py:109-136) */
int l3d_steger_create(double sigma, int brightness_threshold, int max_width, int max_height,
                      int max_batch, int device, l3d_steger **out);
void l3d_steger_destroy(l3d_steger *h);
/* replaces FastStegerExtractor.extract_centerline / extract_batch
 * (core/laser_extractor.py:160-261, 279-285): cv2.cvtColor + GaussianBlur +
 * 5 x filter2D + per-pixel np.linalg.eig loop.
 * img: [batch][H][W][channels] u8, channels 1 or 3 (BGR); roi = {x,y,w,h} or NULL.
 * xy: [batch][capacity][2] float (x, y), raster order; counts[batch] receives the
 * number of points found per image (if > capacity only the first `capacity` are
 * stored and L3D_ERR_CAPACITY is returned). */
int l3d_steger_extract(l3d_steger *h, const uint8_t *img, int channels, int width, int height, int batch,
                       const int *roi, float *xy, int capacity, int *counts);
/* device operands; d_xy [batch][capacity][2] float, d_counts [batch] int.
 * All calls on one handle share its work buffers and must be ordered on the device: use one stream, or order the
 * streams with events.  (A call leaves the buffers in the state the next call expects; a call on another stream than
 * the previous one, and a call after one that failed, clear them first -- which is only safe once the previous call's
 * kernels have run.) */
int l3d_steger_extract_device(l3d_steger *h, const uint8_t *d_img, int channels, int width, int height,
                              int batch, const int *roi, float *d_xy, int capacity, int *d_counts,
                              void *stream);

/* l3d_steger_extract_device without a handle and without any allocation.  No counterpart in the reference: it is for
 * callers that cannot allocate device memory -- code under HIP graph capture, tight serving loops, harnesses that
 * forbid hipMalloc.  It calls no hipMalloc / hipFree or other allocating routine; the work buffers (tile list, per-cell
 * counts and points, filter taps) live in one device array inside the library (8.6 MB per GPU, held until the process
 * ends).  Limit: batch * height * ceil(width / 64) <= 16384 (for example one 1280x720 frame or four 640x360 frames);
 * larger jobs are refused with L3D_ERR_CAPACITY.  Arguments and results as l3d_steger_extract_device, with sigma and
 * brightness_threshold passed per call and `device` naming the GPU the pointers live on.  Every call on one GPU shares
 * that GPU's array: calls that overlap in time on one GPU (two streams, two threads) race; calls on one stream do not. */
int l3d_steger_extract_device_static(double sigma, int brightness_threshold, const uint8_t *d_img, int channels,
                                     int width, int height, int batch, const int *roi, float *d_xy, int capacity,
                                     int *d_counts, int device, void *stream);

/* ---------------------------------------------------------------------------
 * HSV / brightness laser extractor (SimpleLaserExtractor,
 * core/laser_extractor.py:14-100).
 * ------------------------------------------------------------------------- */
typedef struct l3d_simple l3d_simple;

/* replaces SimpleLaserExtractor.__init__ (core/laser_extractor.py:20-43); the
 * HSV range, threshold and min_area are per-call arguments */
int l3d_simple_create(int max_width, int max_height, int device, l3d_simple **out);
void l3d_simple_destroy(l3d_simple *h);
/* replaces SimpleLaserExtractor.extract_centerline (core/laser_extractor.py:45-100):
 * cvtColor(BGR2HSV) + inRange, cvtColor(BGR2GRAY) + threshold, bitwise_and,
 * morphologyEx CLOSE + OPEN (3x3), findContours(RETR_EXTERNAL) + contourArea >
 * min_area + drawContours(filled), per-row intensity-weighted centroid.
 * bgr [H][W][3] u8; hsv_lower / hsv_upper 3 ints (inclusive, H in [0,180));
 * xy receives up to `height` (x, y) pairs of doubles in ascending y, *count how
 * many.  masks (may be NULL): 3 planes of W*H bytes = combined mask, mask after
 * morphology, final filled mask (for stage-level checks). */
int l3d_simple_extract(l3d_simple *h, const uint8_t *bgr, int width, int height, const int *hsv_lower,
                       const int *hsv_upper, int brightness_threshold, double min_area, double *xy,
                       int *count, uint8_t *masks);
/* device-resident image; results still land in host memory (xy, count) after a
 * synchronisation of `stream` */
int l3d_simple_extract_device(l3d_simple *h, const uint8_t *d_bgr, int width, int height,
                              const int *hsv_lower, const int *hsv_upper, int brightness_threshold,
                              double min_area, double *xy, int *count, void *stream);
/* The same for a stack of `batch` images [batch][height][width][3] in one sequence of launches (the reference calls
 * extract_centerline once per frame, main.py:172; a caller with frames in hand batches them).  The handle comes from
 * l3d_simple_create_batch (l3d_simple_create: max_batch = 1; 28 bytes of device memory per pixel of the stack).
 * xy [batch][height][2] doubles, counts [batch]; image f's centres are the first counts[f] pairs of its slice. */
int l3d_simple_create_batch(int max_width, int max_height, int max_batch, int device, l3d_simple **out);
int l3d_simple_extract_batch(l3d_simple *h, const uint8_t *bgr, int width, int height, int batch, const int *hsv_lower,
                             const int *hsv_upper, int brightness_threshold, double min_area, double *xy, int *counts);
int l3d_simple_extract_batch_device(l3d_simple *h, const uint8_t *d_bgr, int width, int height, int batch,
                                    const int *hsv_lower, const int *hsv_upper, int brightness_threshold,
                                    double min_area, double *xy, int *counts, void *stream);

/* ---------------------------------------------------------------------------
 * Steger variants (improved_steger.py).  Points are float32 (x, y) pairs.
 * ------------------------------------------------------------------------- */
/* replaces ImprovedStegerExtractor.__init__ (improved_steger.py:18-37); also the
 * Steger stage of HybridLaserExtractor (sigma only, :232-248) */
int l3d_steger_create_improved(double sigma, int brightness_threshold, double response_threshold,
                               int max_width, int max_height, int max_batch, int device, l3d_steger **out);
/* replaces ImprovedStegerExtractor.extract_centerline (per_row = 0, improved_steger.py:39-126:
 * every accepted pixel, raster order) and .extract_centerline_optimized (per_row = 1,
 * :128-223: per image row the accepted pixel with the largest |eigenvalue|, y = row).
 * Same argument meaning as l3d_steger_extract, without a ROI. */
int l3d_steger_extract_improved(l3d_steger *h, const uint8_t *img, int channels, int width, int height,
                                int batch, int per_row, float *xy, int capacity, int *counts);
int l3d_steger_extract_improved_device(l3d_steger *h, const uint8_t *d_img, int channels, int width, int height,
                                       int batch, int per_row, float *d_xy, int capacity, int *d_counts,
                                       void *stream);
/* replaces HybridLaserExtractor.extract_centerline (improved_steger.py:250-344): HSV +
 * brightness mask, 3x3 close and open (run by `mask_maker`), image zeroed outside the
 * mask, Steger per row on the mask's pixels with |offset_x| <= 0.5.  One BGR image;
 * mask (may be NULL) receives the cleaned mask [height][width]. */
int l3d_hybrid_extract(l3d_simple *mask_maker, l3d_steger *h, const uint8_t *bgr, int width, int height,
                       const int *hsv_lower, const int *hsv_upper, int brightness_threshold, float *xy,
                       int capacity, int *count, uint8_t *mask);
int l3d_hybrid_extract_device(l3d_simple *mask_maker, l3d_steger *h, const uint8_t *d_bgr, int width, int height,
                              const int *hsv_lower, const int *hsv_upper, int brightness_threshold,
                              float *d_xy, int capacity, int *d_count, void *stream);

/* ---------------------------------------------------------------------------
 * Stereo rectification: maps and bilinear remap.
 * ------------------------------------------------------------------------- */
typedef struct l3d_remap l3d_remap;

/* A remapper from caller-supplied float32 maps [height][width] (host pointers),
 * the form cv2.remap takes them in (camera/single_usb_stereo_camera.py:313-314). */
int l3d_remap_create(const float *map_x, const float *map_y, int width, int height, int device, l3d_remap **out);
/* replaces cv2.initUndistortRectifyMap(cameraMatrix, distCoeffs, R, newCameraMatrix,
 * size, CV_32FC1) (camera/single_usb_stereo_camera.py:190-206): the maps are
 * computed on the device.  camera_matrix 3x3 row-major; dist_coeffs k1 k2 p1 p2
 * [k3 [k4 k5 k6 [s1 s2 s3 s4]]] (n_dist = 0, 4, 5, 8 or 12); inv_new_camera_R =
 * inverse of newCameraMatrix[:, :3] * R, 3x3 row-major (a 3x3 inverse the caller
 * does on the host). */
int l3d_remap_create_rectify(const double *camera_matrix, const double *dist_coeffs, int n_dist,
                             const double *inv_new_camera_R, int width, int height, int device, l3d_remap **out);
void l3d_remap_destroy(l3d_remap *h);
/* the float32 maps of the handle, [height][width] each (host pointers) */
int l3d_remap_fetch_maps(l3d_remap *h, float *map_x, float *map_y);
/* replaces cv2.remap(src, map_x, map_y, cv2.INTER_LINEAR) with the default constant
 * border 0 (camera/single_usb_stereo_camera.py:313-314) for `batch` u8 images of
 * `channels` (1 or 3) interleaved channels.  Strides are in bytes, so one half of a
 * side-by-side or top-bottom camera frame (the reference's _split_frame, :143-150)
 * is addressed in place.  dst [batch][height][width][channels] (may be NULL);
 * gray [batch][height][width] (may be NULL; 3-channel sources only) receives
 * cvtColor(BGR2GRAY) of the remapped image (:320-321). */
int l3d_remap_apply(l3d_remap *h, const uint8_t *src, int channels, int src_width, int src_height,
                    ptrdiff_t src_row_stride, ptrdiff_t src_img_stride, int batch, uint8_t *dst, uint8_t *gray);
/* same, device-resident operands, enqueued on `stream` */
int l3d_remap_apply_device(l3d_remap *h, const uint8_t *d_src, int channels, int src_width, int src_height,
                           ptrdiff_t src_row_stride, ptrdiff_t src_img_stride, int batch, uint8_t *d_dst,
                           uint8_t *d_gray, void *stream);

/* ---------------------------------------------------------------------------
 * Laser pixel -> 3-D (Reconstructor, core/reconstruction.py).
 * ------------------------------------------------------------------------- */
/* replaces Reconstructor.reconstruct_from_depth (core/reconstruction.py:145-182).
 * uv: n (u,v) doubles; depth [H][W] float; K row-major 3x3; xyz n*3 doubles
 * (first *n_out rows valid, input order kept). */
int l3d_reconstruct_from_depth(const double *uv, int n, const float *depth, int width, int height,
                               const double *K9, double *xyz, int *n_out, int device);
/* replaces Reconstructor.reconstruct_point / _refraction_correction /
 * reconstruct_laser_line (core/reconstruction.py:30-143).  xyz n*3 doubles, rows
 * of NaN where the reference returns NaN. */
int l3d_reconstruct_plane(const double *uv, int n, const double *Kinv9, const double *plane4,
                          int refraction, double n_water, double *xyz, int device);

/* The two calls above for laser pixels that are still in HBM, i.e. the device output of l3d_steger_extract_device
 * (d_xy [batch][capacity][2] float32, d_counts [batch]) -- the frame loop of main.py:172-176 without a round trip
 * through host memory.  Frame b back-projects through d_depth[b] ([batch][height][width] float32).  No compaction:
 * d_xyz [batch][capacity][3] doubles, d_keep [batch][capacity] = 1 where the reference appends the point (slots
 * past a frame's count: 0).  The float32 coordinates are widened to double first, which is what the reference's
 * Python floats hold. */
int l3d_reconstruct_from_depth_device(const float *d_xy, const int *d_counts, int capacity, int batch,
                                      const float *d_depth, int width, int height, const double *K9, double *d_xyz,
                                      int *d_keep, void *stream);
/* Reconstructor.reconstruct_from_depth (core/reconstruction.py:145-182) for one frame whose depth map is still in HBM
 * and whose laser pixels are doubles (the row centroids of SimpleLaserExtractor, core/laser_extractor.py:88-98, are
 * float64): d_uv n (u,v) doubles, d_depth [height][width] float32, d_xyz [n][3] doubles, d_keep [n] = 1 where the
 * reference appends the point.  No compaction. */
int l3d_reconstruct_from_depth_uv64_device(const double *d_uv, int n, const float *d_depth, int width, int height,
                                           const double *K9, double *d_xyz, int *d_keep, void *stream);
/* d_xyz [batch][capacity][3] doubles: rows of NaN where the reference returns NaN and past a frame's count */
int l3d_reconstruct_plane_device(const float *d_xy, const int *d_counts, int capacity, int batch,
                                 const double *Kinv9, const double *plane4, int refraction, double n_water,
                                 double *d_xyz, void *stream);

/* replaces ImprovedLaserReconstructor.reconstruct_from_disparity (window_size = 0,
 * improved_reconstruction.py:37-86) and .reconstruct_with_interpolation (window_size odd >= 1:
 * median of the valid disparities of the window, :88-152).  uv: n (x, y) doubles; disparity
 * [height][width] float32 (or float64 with disparity_is_f64); Q16: the 4x4 reprojection matrix,
 * of which fx = Q[2][3], baseline = 1/Q[3][2], cx = -Q[0][3], cy = -Q[1][3] are used (:26-30);
 * xyz: float32 triples of the kept points (first *n_out rows, input order kept). */
int l3d_triangulate_disparity(const double *uv, int n, const void *disparity, int disparity_is_f64, int width,
                              int height, const double *Q16, int window_size, double min_disparity, float *xyz,
                              int *n_out, int device);
/* same on device-resident operands (float32 map), no compaction: d_xyz [n][3], d_keep [n] (1 where
 * the reference appends the point) */
int l3d_triangulate_disparity_device(const double *d_uv, int n, const float *d_disparity, int width, int height,
                                     const double *Q16, int window_size, double min_disparity, float *d_xyz,
                                     int *d_keep, void *stream);
/* replaces ImprovedLaserReconstructor.create_laser_depth_map (improved_reconstruction.py:154-186)
 * with image_shape = the disparity map's shape: depth [height][width] float32, zero except at the
 * rounded laser pixels. */
int l3d_laser_depth_map(const double *uv, int n, const void *disparity, int disparity_is_f64, int width, int height,
                        const double *Q16, float *depth, int device);

/* ---------------------------------------------------------------------------
 * Point-cloud post-processing (PointCloudProcessor without Open3D, utils/point_cloud.py).
 * Host pointers; points are n x 3 doubles.
 * ------------------------------------------------------------------------- */
/* replaces PointCloudProcessor._simple_voxel_downsample behind voxel_downsample
 * (utils/point_cloud.py:31-79): one point per occupied voxel floor(p / voxel_size), the
 * mean of the voxel's points, in order of each voxel's first point.  out: room for n points. */
int l3d_voxel_downsample(const double *points, int n, double voxel_size, double *out, int *n_out, int device);
/* replaces PointCloudProcessor._simple_outlier_removal behind statistical_outlier_removal
 * (utils/point_cloud.py:81-132): a point is kept when mean < mean + std_ratio * std of its
 * nb_neighbors nearest-neighbour distances (true unless the deviation vanishes or the
 * cloud has fewer than nb_neighbors + 1 points); clouds smaller than nb_neighbors are
 * returned unchanged.  out: room for n points. */
int l3d_sor_simple(const double *points, int n, int nb_neighbors, double std_ratio, double *out, int *n_out,
                   int device);
/* nearest-neighbour distance of every point: the cKDTree.query(points, k=2)[:, 1] of
 * compute_point_cloud_metrics (utils/point_cloud.py:270-276). */
int l3d_nn_distances(const double *points, int n, double *nn, int device);

#ifdef __cplusplus
}
#endif
#endif /* L3D_H */
