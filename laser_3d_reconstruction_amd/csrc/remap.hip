// Stereo rectification for gfx950: map generation and bilinear remap.
//
// Replaces cv2.initUndistortRectifyMap (camera/single_usb_stereo_camera.py:190-206) and
// cv2.remap(..., cv2.INTER_LINEAR) of the split camera frame (:143-150, 312-314) of the reference.
//
// Data layout in HBM (per handle): mapx, mapy [H][W] f32 (what the reference keeps as map_left_x/...),
// and fix [H][W] int2 = the maps in the 1/32-pixel fixed point that remap actually works in, computed
// once.  The remap kernel is a gather: 8 B of fixed-point map + 4 taps x C bytes (cache hits for a
// smooth map) in, C bytes (+1 for the optional gray plane) out per pixel; HBM-bound.
#include "common.h"
#include <cmath>
#include <new>

namespace l3d {

struct RectifyArgs {
    double K[9];
    double dist[12]; // k1 k2 p1 p2 k3 k4 k5 k6 s1 s2 s3 s4
    double iR[9];    // inverse of newCameraMatrix[:, :3] * R
};

// s = cvRound(map * 32): product in float, rounded half to even; NaN / out of int range give INT_MIN
__device__ __forceinline__ int remap_fix(float m)
{
    const double v = (double)(m * 32.0f);
    if (!(v > -2147483648.0 && v < 2147483647.0)) return (int)0x80000000;
    return (int)rint(v);
}

__global__ void __launch_bounds__(256) k_rectify_map(RectifyArgs a, int W, int H, float *__restrict__ mapx, float *__restrict__ mapy,
                                                     int2 *__restrict__ fix)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x, i = blockIdx.y;
    if (j >= W) return;
    const double u0 = a.K[2], v0 = a.K[5], fx = a.K[0], fy = a.K[4];
    const double k1 = a.dist[0], k2 = a.dist[1], p1 = a.dist[2], p2 = a.dist[3], k3 = a.dist[4], k4 = a.dist[5], k5 = a.dist[6], k6 = a.dist[7];
    const double s1 = a.dist[8], s2 = a.dist[9], s3 = a.dist[10], s4 = a.dist[11];
    const double _x = j * a.iR[0] + (i * a.iR[1] + a.iR[2]);
    const double _y = j * a.iR[3] + (i * a.iR[4] + a.iR[5]);
    const double _w = j * a.iR[6] + (i * a.iR[7] + a.iR[8]);
    const double w = 1. / _w, x = _x * w, y = _y * w;
    const double x2 = x * x, y2 = y * y;
    const double r2 = x2 + y2, _2xy = 2 * x * y;
    const double kr = (1 + ((k3 * r2 + k2) * r2 + k1) * r2) / (1 + ((k6 * r2 + k5) * r2 + k4) * r2);
    const double xd = (x * kr + p1 * _2xy + p2 * (r2 + 2 * x2) + s1 * r2 + s2 * r2 * r2);
    const double yd = (y * kr + p1 * (r2 + 2 * y2) + p2 * _2xy + s3 * r2 + s4 * r2 * r2);
    const float mx = (float)(fx * xd + u0), my = (float)(fy * yd + v0);
    const size_t o = (size_t)i * W + j;
    mapx[o] = mx;
    mapy[o] = my;
    fix[o] = make_int2(remap_fix(mx), remap_fix(my));
}

__global__ void __launch_bounds__(256) k_map_to_fix(const float *__restrict__ mapx, const float *__restrict__ mapy, size_t n, int2 *__restrict__ fix)
{
    const size_t o = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (o >= n) return;
    fix[o] = make_int2(remap_fix(mapx[o]), remap_fix(mapy[o]));
}

// Bilinear remap of u8 images with CH interleaved channels, constant border 0.  The integer part of a map
// entry is saturated to int16 and the four weights are (32-fx)(32-fy)*32 ... fx*fy*32 (sum 2^15), as in
// OpenCV's fixed-point path.  With CH == 3 a gray plane of the remapped pixel can be written as well
// (the same value cvtColor(BGR2GRAY) gives on the remapped image).
template <int CH>
__global__ void __launch_bounds__(256) k_remap(const uint8_t *__restrict__ src, int sw, int sh, size_t src_row, size_t src_img,
                                               const int2 *__restrict__ fix, int dw, int dh, uint8_t *__restrict__ dst,
                                               uint8_t *__restrict__ gray)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x, i = blockIdx.y, b = blockIdx.z;
    if (j >= dw) return;
    const int2 s = fix[(size_t)i * dw + j];
    const int fx = s.x & 31, fy = s.y & 31;
    const int ix = min(max(s.x >> 5, -32768), 32767), iy = min(max(s.y >> 5, -32768), 32767);
    const int w00 = (32 - fx) * (32 - fy) * 32, w01 = fx * (32 - fy) * 32, w10 = (32 - fx) * fy * 32, w11 = fx * fy * 32;
    const uint8_t *img = src + (size_t)b * src_img;
    const bool x0 = ix >= 0 && ix < sw, x1 = ix + 1 >= 0 && ix + 1 < sw, y0 = iy >= 0 && iy < sh, y1 = iy + 1 >= 0 && iy + 1 < sh;
    const uint8_t *r0 = img + (size_t)(y0 ? iy : 0) * src_row, *r1 = img + (size_t)(y1 ? iy + 1 : 0) * src_row;
    const size_t c0 = (size_t)(x0 ? ix : 0) * CH, c1 = (size_t)(x1 ? ix + 1 : 0) * CH;
    int out[CH];
#pragma unroll
    for (int c = 0; c < CH; c++) {
        const int p00 = (x0 && y0) ? r0[c0 + c] : 0, p01 = (x1 && y0) ? r0[c1 + c] : 0;
        const int p10 = (x0 && y1) ? r1[c0 + c] : 0, p11 = (x1 && y1) ? r1[c1 + c] : 0;
        out[c] = (p00 * w00 + p01 * w01 + p10 * w10 + p11 * w11 + (1 << 14)) >> 15;
    }
    const size_t o = ((size_t)b * dh + i) * dw + j;
    if (dst) {
#pragma unroll
        for (int c = 0; c < CH; c++) dst[o * CH + c] = (uint8_t)out[c];
    }
    if (CH == 3 && gray) gray[o] = (uint8_t)(((unsigned)out[2] * 9798u + (unsigned)out[1] * 19235u + (unsigned)out[0] * 3735u + (1u << 14)) >> 15);
}

} // namespace l3d

using namespace l3d;

struct l3d_remap {
    int device = 0, W = 0, H = 0;
    hipStream_t stream = nullptr;
    float *mapx = nullptr, *mapy = nullptr;
    int2 *fix = nullptr;
    uint8_t *stage_src = nullptr, *stage_dst = nullptr; // for the host entry point, grown on demand
    size_t stage_src_bytes = 0, stage_dst_bytes = 0;
};

static int remap_new(int width, int height, int device, l3d_remap **out)
{
    L3D_REQUIRE(out, "l3d_remap: null output");
    *out = nullptr;
    L3D_REQUIRE(width >= 1 && height >= 1, "l3d_remap: bad map size %dx%d", width, height);
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        set_error("l3d_remap: no HIP device available");
        return L3D_ERR_NO_DEVICE;
    }
    L3D_REQUIRE(device >= 0 && device < ndev, "l3d_remap: device %d out of range (%d devices)", device, ndev);
    l3d_remap *h = new (std::nothrow) l3d_remap();
    L3D_REQUIRE(h, "l3d_remap: out of host memory");
    h->device = device; h->W = width; h->H = height;
    DeviceGuard dg(device);
    const size_t n = (size_t)width * height;
    hipError_t e = hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipMalloc((void **)&h->mapx, n * sizeof(float));
    if (e == hipSuccess) e = hipMalloc((void **)&h->mapy, n * sizeof(float));
    if (e == hipSuccess) e = hipMalloc((void **)&h->fix, n * sizeof(int2));
    if (e != hipSuccess) {
        set_error("l3d_remap: allocation failed: %s", hipGetErrorString(e));
        l3d_remap_destroy(h);
        return L3D_ERR_HIP;
    }
    *out = h;
    return L3D_OK;
}

extern "C" void l3d_remap_destroy(l3d_remap *h)
{
    if (!h) return;
    DeviceGuard dg(h->device);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    void *bufs[] = {h->mapx, h->mapy, h->fix, h->stage_src, h->stage_dst};
    for (void *p : bufs)
        if (p) (void)hipFree(p);
    if (h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
}

extern "C" int l3d_remap_create(const float *map_x, const float *map_y, int width, int height, int device, l3d_remap **out)
{
    L3D_REQUIRE(map_x && map_y, "l3d_remap_create: null map");
    int rc = remap_new(width, height, device, out);
    if (rc) return rc;
    l3d_remap *h = *out;
    DeviceGuard dg(device);
    const size_t n = (size_t)width * height;
    hipError_t e = hipMemcpyAsync(h->mapx, map_x, n * sizeof(float), hipMemcpyHostToDevice, h->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(h->mapy, map_y, n * sizeof(float), hipMemcpyHostToDevice, h->stream);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_map_to_fix, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, h->stream, h->mapx, h->mapy, n, h->fix);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    if (e != hipSuccess) {
        set_error("l3d_remap_create: %s", hipGetErrorString(e));
        l3d_remap_destroy(h);
        *out = nullptr;
        return L3D_ERR_HIP;
    }
    return L3D_OK;
}

extern "C" int l3d_remap_create_rectify(const double *camera_matrix, const double *dist_coeffs, int n_dist, const double *inv_new_camera_R,
                                        int width, int height, int device, l3d_remap **out)
{
    L3D_REQUIRE(camera_matrix && inv_new_camera_R && (dist_coeffs || n_dist == 0), "l3d_remap_create_rectify: null argument");
    if (!(n_dist == 0 || n_dist == 4 || n_dist == 5 || n_dist == 8 || n_dist == 12)) {
        set_error("l3d_remap_create_rectify: %d distortion coefficients (0, 4, 5, 8 or 12 are supported; the tilted-sensor pair is not)", n_dist);
        return L3D_ERR_UNSUPPORTED;
    }
    int rc = remap_new(width, height, device, out);
    if (rc) return rc;
    l3d_remap *h = *out;
    DeviceGuard dg(device);
    RectifyArgs a;
    for (int i = 0; i < 9; i++) { a.K[i] = camera_matrix[i]; a.iR[i] = inv_new_camera_R[i]; }
    for (int i = 0; i < 12; i++) a.dist[i] = i < n_dist ? dist_coeffs[i] : 0.0;
    hipLaunchKernelGGL(k_rectify_map, dim3(cdiv(width, 256), height), dim3(256), 0, h->stream, a, width, height, h->mapx, h->mapy, h->fix);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    if (e != hipSuccess) {
        set_error("l3d_remap_create_rectify: %s", hipGetErrorString(e));
        l3d_remap_destroy(h);
        *out = nullptr;
        return L3D_ERR_HIP;
    }
    return L3D_OK;
}

extern "C" int l3d_remap_fetch_maps(l3d_remap *h, float *map_x, float *map_y)
{
    L3D_REQUIRE(h && map_x && map_y, "l3d_remap_fetch_maps: null argument");
    DeviceGuard dg(h->device);
    const size_t bytes = (size_t)h->W * h->H * sizeof(float);
    L3D_HIP_CHECK(hipMemcpy(map_x, h->mapx, bytes, hipMemcpyDeviceToHost));
    L3D_HIP_CHECK(hipMemcpy(map_y, h->mapy, bytes, hipMemcpyDeviceToHost));
    return L3D_OK;
}

static int remap_launch(l3d_remap *h, const uint8_t *d_src, int channels, int src_w, int src_h, size_t src_row, size_t src_img, int batch,
                        uint8_t *d_dst, uint8_t *d_gray, hipStream_t s)
{
    dim3 grid(cdiv(h->W, 256), h->H, batch);
    if (channels == 3)
        hipLaunchKernelGGL(k_remap<3>, grid, dim3(256), 0, s, d_src, src_w, src_h, src_row, src_img, h->fix, h->W, h->H, d_dst, d_gray);
    else
        hipLaunchKernelGGL(k_remap<1>, grid, dim3(256), 0, s, d_src, src_w, src_h, src_row, src_img, h->fix, h->W, h->H, d_dst, (uint8_t *)nullptr);
    L3D_HIP_CHECK(hipGetLastError());
    return L3D_OK;
}

static int remap_check(const char *who, l3d_remap *h, const void *src, int channels, int src_w, int src_h, ptrdiff_t src_row, ptrdiff_t src_img,
                       int batch, const void *dst, const void *gray)
{
    L3D_REQUIRE(h && src && (dst || gray), "%s: null argument", who);
    L3D_REQUIRE(channels == 1 || channels == 3, "%s: %d channels (1 or 3 are supported)", who, channels);
    L3D_REQUIRE(!(gray && channels != 3), "%s: the gray plane needs a 3-channel source", who);
    L3D_REQUIRE(src_w >= 1 && src_h >= 1 && batch >= 1, "%s: bad source size %dx%d batch %d", who, src_w, src_h, batch);
    L3D_REQUIRE(src_row >= (ptrdiff_t)src_w * channels && (batch == 1 || src_img >= src_row * (src_h - 1) + (ptrdiff_t)src_w * channels),
                "%s: strides smaller than the image", who);
    return L3D_OK;
}

extern "C" int l3d_remap_apply_device(l3d_remap *h, const uint8_t *d_src, int channels, int src_w, int src_h, ptrdiff_t src_row_stride,
                                      ptrdiff_t src_img_stride, int batch, uint8_t *d_dst, uint8_t *d_gray, void *stream)
{
    int rc = remap_check("l3d_remap_apply_device", h, d_src, channels, src_w, src_h, src_row_stride, src_img_stride, batch, d_dst, d_gray);
    if (rc) return rc;
    DeviceGuard dg(h->device);
    return remap_launch(h, d_src, channels, src_w, src_h, (size_t)src_row_stride, (size_t)src_img_stride, batch, d_dst, d_gray, (hipStream_t)stream);
}

extern "C" int l3d_remap_apply(l3d_remap *h, const uint8_t *src, int channels, int src_w, int src_h, ptrdiff_t src_row_stride,
                               ptrdiff_t src_img_stride, int batch, uint8_t *dst, uint8_t *gray)
{
    int rc = remap_check("l3d_remap_apply", h, src, channels, src_w, src_h, src_row_stride, src_img_stride, batch, dst, gray);
    if (rc) return rc;
    DeviceGuard dg(h->device);
    // the source is copied as stored (strides included): one contiguous span per call
    const size_t span = (size_t)src_img_stride * (batch - 1) + (size_t)src_row_stride * (src_h - 1) + (size_t)src_w * channels;
    const size_t npx = (size_t)h->W * h->H * batch;
    const size_t out_bytes = npx * channels + npx; // remapped image, then the gray plane
    if (span > h->stage_src_bytes) {
        if (h->stage_src) (void)hipFree(h->stage_src);
        h->stage_src = nullptr; h->stage_src_bytes = 0;
        L3D_HIP_CHECK(hipMalloc((void **)&h->stage_src, span));
        h->stage_src_bytes = span;
    }
    if (out_bytes > h->stage_dst_bytes) {
        if (h->stage_dst) (void)hipFree(h->stage_dst);
        h->stage_dst = nullptr; h->stage_dst_bytes = 0;
        L3D_HIP_CHECK(hipMalloc((void **)&h->stage_dst, out_bytes));
        h->stage_dst_bytes = out_bytes;
    }
    L3D_HIP_CHECK(hipMemcpyAsync(h->stage_src, src, span, hipMemcpyHostToDevice, h->stream));
    uint8_t *d_dst = h->stage_dst, *d_gray = gray ? h->stage_dst + npx * channels : nullptr;
    rc = remap_launch(h, h->stage_src, channels, src_w, src_h, (size_t)src_row_stride, (size_t)src_img_stride, batch, d_dst, d_gray, h->stream);
    if (rc) return rc;
    if (dst) L3D_HIP_CHECK(hipMemcpyAsync(dst, d_dst, npx * channels, hipMemcpyDeviceToHost, h->stream));
    if (gray) L3D_HIP_CHECK(hipMemcpyAsync(gray, d_gray, npx, hipMemcpyDeviceToHost, h->stream));
    L3D_HIP_CHECK(hipStreamSynchronize(h->stream));
    return L3D_OK;
}
