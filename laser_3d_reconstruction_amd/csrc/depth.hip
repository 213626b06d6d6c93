// Colour conversion, disparity -> depth, and laser-pixel -> 3-D kernels for gfx950.
//
// Replaces cv2.cvtColor(BGR2GRAY), `.astype(float32)/16`, cv2.reprojectImageTo3D and the
// validity masks of the reference (camera/single_usb_stereo_camera.py:320-321, 335-357) and
// Reconstructor.reconstruct_from_depth / reconstruct_point (core/reconstruction.py:30-182).
// All of these are element-wise, HBM-bound kernels (the gray conversion handles 4 adjacent
// pixels per thread so that its loads and stores are whole dwords).
#include "common.h"
#include <cmath>
#include <cstring>
#include <vector>

namespace l3d {

// 8-bit BGR -> gray in 15-bit fixed point: Y = (R*9798 + G*19235 + B*3735 + 2^14) >> 15
__device__ __forceinline__ uint32_t gray_px(uint32_t b, uint32_t g, uint32_t r)
{
    return (r * 9798u + g * 19235u + b * 3735u + (1u << 14)) >> 15;
}

__global__ void __launch_bounds__(256) k_bgr2gray(const uint8_t *__restrict__ bgr, size_t npx, uint8_t *__restrict__ gray)
{
    const size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x; // group of 4 pixels
    const size_t p0 = q * 4;
    if (p0 >= npx) return;
    if (p0 + 4 <= npx) {
        const uint32_t *s = reinterpret_cast<const uint32_t *>(bgr + p0 * 3); // 12-byte aligned to 4
        const uint32_t w0 = s[0], w1 = s[1], w2 = s[2];
        const uint32_t y0 = gray_px(w0 & 0xff, (w0 >> 8) & 0xff, (w0 >> 16) & 0xff);
        const uint32_t y1 = gray_px(w0 >> 24, w1 & 0xff, (w1 >> 8) & 0xff);
        const uint32_t y2 = gray_px((w1 >> 16) & 0xff, w1 >> 24, w2 & 0xff);
        const uint32_t y3 = gray_px((w2 >> 8) & 0xff, (w2 >> 16) & 0xff, w2 >> 24);
        *reinterpret_cast<uint32_t *>(gray + p0) = y0 | (y1 << 8) | (y2 << 16) | (y3 << 24);
    } else {
        for (size_t p = p0; p < npx; p++) gray[p] = (uint8_t)gray_px(bgr[3 * p], bgr[3 * p + 1], bgr[3 * p + 2]);
    }
}

struct QMat {
    double q[16];
};

// Z of cv2.reprojectImageTo3D for float disparity d at (x, y): the homogeneous point is
// computed in double, cast to float, then divided by the double W.  Without an xyz output only
// the Z and W rows of Q are evaluated (one double division per pixel instead of three).
// Block = 64 columns x 4 rows.
template <bool XYZ>
__global__ void __launch_bounds__(256) k_depth(const int16_t *__restrict__ disp16, int W, int H, int B, QMat Q, int calibrated,
                                               float *__restrict__ depth, float *__restrict__ xyz)
{
    const int x = blockIdx.x * 64 + threadIdx.x;
    const int y = blockIdx.y * 4 + threadIdx.y, b = blockIdx.z;
    if (x >= W || y >= H) return;
    const size_t i = ((size_t)b * H + y) * W + x;
    const float d = (float)disp16[i] / 16.0f;
    float z;
    if (calibrated) {
        const double dd = (double)d, xd = (double)x, yd = (double)y;
        double hp[4];
#pragma unroll
        for (int r = XYZ ? 0 : 2; r < 4; r++) {
            double s = 0.0;
            s += Q.q[4 * r + 0] * xd;
            s += Q.q[4 * r + 1] * yd;
            s += Q.q[4 * r + 2] * dd;
            s += Q.q[4 * r + 3] * 1.0;
            hp[r] = s;
        }
        if (XYZ) {
            xyz[3 * i] = (float)((double)(float)hp[0] / hp[3]);
            xyz[3 * i + 1] = (float)((double)(float)hp[1] / hp[3]);
        }
        const float Z = (float)((double)(float)hp[2] / hp[3]);
        if (XYZ) xyz[3 * i + 2] = Z;
        z = Z;
        if (z < 0.0f) z = 0.0f;
        if (z > 10.0f) z = 0.0f;
        if (d <= 0.0f) z = 0.0f;
    } else {
        z = 0.0f;
        if (d > 0.0f) z = (float)((0.06 * 350) / (double)d);
        if (z > 10.0f) z = 0.0f;
    }
    depth[i] = z;
}

// When Z does not depend on the pixel position -- the Z and W rows of Q have no x / y terms, which is the form
// cv2.stereoRectify produces, or there is no Q at all -- the depth of a pixel is a function of its 16-bit disparity
// alone.  k_depth_lut_build evaluates that function for all 65536 disparity values with exactly the operations of
// k_depth (at x = y = 0: the x / y products are zeros added first, which changes nothing), and k_depth_lut_apply
// turns a disparity map into depth by table look-up: one streaming pass with no double-precision division in it.
__global__ void __launch_bounds__(256) k_depth_lut_build(QMat Q, int calibrated, float *__restrict__ lut)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;   // 0 .. 65535 <-> disparity value i - 32768
    const float d = (float)(int16_t)(i - 32768) / 16.0f;
    float z;
    if (calibrated) {
        const double dd = (double)d, xd = 0.0, yd = 0.0;
        double hp[4];
#pragma unroll
        for (int r = 2; r < 4; r++) {
            double s = 0.0;
            s += Q.q[4 * r + 0] * xd;
            s += Q.q[4 * r + 1] * yd;
            s += Q.q[4 * r + 2] * dd;
            s += Q.q[4 * r + 3] * 1.0;
            hp[r] = s;
        }
        z = (float)((double)(float)hp[2] / hp[3]);
        if (z < 0.0f) z = 0.0f;
        if (z > 10.0f) z = 0.0f;
        if (d <= 0.0f) z = 0.0f;
    } else {
        z = 0.0f;
        if (d > 0.0f) z = (float)((0.06 * 350) / (double)d);
        if (z > 10.0f) z = 0.0f;
    }
    lut[i] = z;
}

__global__ void __launch_bounds__(256) k_depth_lut_apply(const int16_t *__restrict__ disp16, size_t n, const float *__restrict__ lut,
                                                         float *__restrict__ depth)
{
    const size_t i0 = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i0 >= n) return;
    if (i0 + 4 <= n && (((uintptr_t)(disp16 + i0)) & 7) == 0 && (((uintptr_t)(depth + i0)) & 15) == 0) {
        const uint2 w = *reinterpret_cast<const uint2 *>(disp16 + i0);
        float4 o;
        o.x = lut[(int)(int16_t)(w.x & 0xffffu) + 32768];
        o.y = lut[(int)(int16_t)(w.x >> 16) + 32768];
        o.z = lut[(int)(int16_t)(w.y & 0xffffu) + 32768];
        o.w = lut[(int)(int16_t)(w.y >> 16) + 32768];
        *reinterpret_cast<float4 *>(depth + i0) = o;
    } else {
        for (size_t i = i0; i < n && i < i0 + 4; i++) depth[i] = lut[(int)disp16[i] + 32768];
    }
}

// reconstruct_from_depth for one laser pixel (u, v): true and xyz[0..2] written where the reference appends a point
__device__ __forceinline__ bool from_depth_px(double u, double v, const float *__restrict__ depth, int W, int H, double fx, double fy, double cx,
                                              double cy, double *__restrict__ xyz)
{
    if (!(u == u && v == v && fabs(u) < 2147483000.0 && fabs(v) < 2147483000.0)) return false;
    const long iu = (long)u, iv = (long)v; // truncation toward zero, as Python int()
    if (!(0 <= iu && iu < W && 0 <= iv && iv < H)) return false;
    const float dep = depth[(size_t)iv * W + iu];
    if (!(dep > 0)) return false;
    const double z = (double)dep / 1000.0;
    xyz[0] = (u - cx) * z / fx;
    xyz[1] = (v - cy) * z / fy;
    xyz[2] = z;
    return true;
}

// reconstruct_from_depth for n laser pixels: keep[i] = 1 and xyz[i] valid where the reference
// appends a point.  Compaction (order preserving) happens on the host: n is a few thousand.
__global__ void __launch_bounds__(256) k_from_depth(const double *__restrict__ uv, int n, const float *__restrict__ depth, int W, int H,
                                                    double fx, double fy, double cx, double cy, double *__restrict__ xyz, int *__restrict__ keep)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    keep[i] = from_depth_px(uv[2 * i], uv[2 * i + 1], depth, W, H, fx, fy, cx, cy, xyz + 3 * (size_t)i) ? 1 : 0;
}

// The same for a batch of frames whose laser pixels are the extractor's device output: xy [B][cap][2] float32,
// counts [B]; frame b back-projects through depth map b.  Slots past a frame's count get keep = 0.
__global__ void __launch_bounds__(256) k_from_depth_batch(const float *__restrict__ xy, const int *__restrict__ counts, int cap,
                                                          const float *__restrict__ depth, int W, int H, double fx, double fy, double cx, double cy,
                                                          double *__restrict__ xyz, int *__restrict__ keep)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x, b = blockIdx.y;
    if (i >= cap) return;
    const size_t idx = (size_t)b * cap + i;
    int k = 0;
    if (i < min(counts[b], cap))
        k = from_depth_px((double)xy[2 * idx], (double)xy[2 * idx + 1], depth + (size_t)b * W * H, W, H, fx, fy, cx, cy, xyz + 3 * idx) ? 1 : 0;
    keep[idx] = k;
}

// ImprovedLaserReconstructor (improved_reconstruction.py:17-186): triangulation straight from the disparity map.
struct TriArgs {
    double fx, baseline, cx, cy;
    double min_disparity;
    int W, H, window; // window 0 = nearest pixel, else odd window edge for the median
};

// Python's int(round(v)): round half to even; false for NaN / out of int range
__device__ __forceinline__ bool py_round_int(double v, int &out)
{
    if (!(v == v) || !(fabs(v) < 2147483000.0)) return false;
    out = (int)rint(v);
    return true;
}

template <typename T>
__device__ __forceinline__ bool disp_valid(T v, double min_disparity)
{
    const double d = (double)v;
    return d >= min_disparity && !isnan(d) && !isinf(d);
}

// value of rank r (0-based, ascending) among the valid disparities of the window: counted, not sorted, so no
// per-thread array is needed (windows are at most a few dozen pixels and only laser pixels are processed)
template <typename T>
__device__ T window_rank(const T *__restrict__ disp, const TriArgs &a, int px, int py, int hw, int r)
{
    for (int y = py - hw; y <= py + hw; y++)
        for (int x = px - hw; x <= px + hw; x++) {
            const T v = disp[(size_t)y * a.W + x];
            if (!disp_valid(v, a.min_disparity)) continue;
            int less = 0, equal = 0;
            for (int yy = py - hw; yy <= py + hw; yy++)
                for (int xx = px - hw; xx <= px + hw; xx++) {
                    const T u = disp[(size_t)yy * a.W + xx];
                    if (!disp_valid(u, a.min_disparity)) continue;
                    less += u < v;
                    equal += u == v;
                }
            if (less <= r && r < less + equal) return v;
        }
    return (T)0;
}

template <typename T>
__global__ void __launch_bounds__(256) k_triangulate(const double *__restrict__ uv, int n, const T *__restrict__ disp, TriArgs a,
                                                     float *__restrict__ xyz, int *__restrict__ keep)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    keep[i] = 0;
    int px, py;
    if (!py_round_int(uv[2 * i], px) || !py_round_int(uv[2 * i + 1], py)) return;
    double d;
    if (a.window == 0) {
        if (px < 0 || px >= a.W || py < 0 || py >= a.H) return;
        const T v = disp[(size_t)py * a.W + px];
        if (!disp_valid(v, a.min_disparity)) return;
        d = (double)v;
    } else {
        const int hw = a.window / 2;
        if (px < hw || px >= a.W - hw || py < hw || py >= a.H - hw) return;
        int k = 0;
        for (int y = py - hw; y <= py + hw; y++)
            for (int x = px - hw; x <= px + hw; x++) k += disp_valid(disp[(size_t)y * a.W + x], a.min_disparity);
        if (k == 0) return;
        if (k & 1) d = (double)window_rank(disp, a, px, py, hw, k / 2);
        else {
            // np.median averages the two middle values in the map's own type
            const T lo = window_rank(disp, a, px, py, hw, k / 2 - 1), hi = window_rank(disp, a, px, py, hw, k / 2);
            d = (double)((T)(lo + hi) / (T)2);
        }
    }
    const double Z = (a.fx * a.baseline) / d;
    const double X = (px - a.cx) * Z / a.fx, Y = (py - a.cy) * Z / a.fx; // Y / fx as well: the reference's formula
    if (Z > 0 && Z < 10.0) {
        xyz[3 * i] = (float)X; xyz[3 * i + 1] = (float)Y; xyz[3 * i + 2] = (float)Z;
        keep[i] = 1;
    }
}

template <typename T>
__global__ void __launch_bounds__(256) k_laser_depth(const double *__restrict__ uv, int n, const T *__restrict__ disp, TriArgs a,
                                                     float *__restrict__ depth)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int px, py;
    if (!py_round_int(uv[2 * i], px) || !py_round_int(uv[2 * i + 1], py)) return;
    if (px < 0 || px >= a.W || py < 0 || py >= a.H) return;
    const double d = (double)disp[(size_t)py * a.W + px];
    if (d > 1.0 && !isnan(d)) {
        const double z = (a.fx * a.baseline) / d;
        if (0 < z && z < 10.0) depth[(size_t)py * a.W + px] = (float)z; // several points on one pixel write the same value
    }
}

struct PlaneArgs {
    double kinv[9];
    double plane[4];
    double n_water;
    int refraction;
};

// reconstruct_point (+ _refraction_correction) for pixel (u, v): o[0..2] = the point, NaN where the reference returns NaN
__device__ __forceinline__ void plane_px(double u, double v, const PlaneArgs &a, double *__restrict__ o)
{
    double r[3];
#pragma unroll
    for (int k = 0; k < 3; k++) r[k] = a.kinv[3 * k] * u + a.kinv[3 * k + 1] * v + a.kinv[3 * k + 2] * 1.0;
    const double nr = sqrt(r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);
    r[0] /= nr; r[1] /= nr; r[2] /= nr;
    const double pa = a.plane[0], pb = a.plane[1], pc = a.plane[2], pd = a.plane[3];
    const double den = pa * r[0] + pb * r[1] + pc * r[2];
    const double qnan = __builtin_nan("");
    if (fabs(den) < 1e-10) { o[0] = o[1] = o[2] = qnan; return; }
    const double t = -pd / den;
    if (t < 0) { o[0] = o[1] = o[2] = qnan; return; }
    double P[3] = {t * r[0], t * r[1], t * r[2]};
    if (a.refraction) {
        const double n1 = 1.0, n2 = a.n_water;
        const double cos1 = -(r[2]);
        const double sin1 = sqrt(1 - cos1 * cos1);
        const double sin2 = (n1 / n2) * sin1;
        if (!(sin2 > 1)) {
            const double cos2 = sqrt(1 - sin2 * sin2);
            double q[3] = {(n1 / n2) * r[0], (n1 / n2) * r[1], (n1 / n2) * r[2] + ((n1 / n2) * cos1 - cos2)};
            const double nq = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2]);
            q[0] /= nq; q[1] /= nq; q[2] /= nq;
            const double den2 = pa * q[0] + pb * q[1] + pc * q[2];
            if (!(fabs(den2) < 1e-10)) {
                const double t2 = -pd / den2;
                P[0] = t2 * q[0]; P[1] = t2 * q[1]; P[2] = t2 * q[2];
            }
        }
    }
    o[0] = P[0]; o[1] = P[1]; o[2] = P[2];
}

__global__ void __launch_bounds__(256) k_plane(const double *__restrict__ uv, int n, PlaneArgs a, double *__restrict__ xyz)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    plane_px(uv[2 * i], uv[2 * i + 1], a, xyz + 3 * (size_t)i);
}

// The same for the extractor's device output (see k_from_depth_batch); slots past a frame's count are filled with NaN
__global__ void __launch_bounds__(256) k_plane_batch(const float *__restrict__ xy, const int *__restrict__ counts, int cap, PlaneArgs a,
                                                     double *__restrict__ xyz)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x, b = blockIdx.y;
    if (i >= cap) return;
    const size_t idx = (size_t)b * cap + i;
    double *o = xyz + 3 * idx;
    if (i < min(counts[b], cap)) plane_px((double)xy[2 * idx], (double)xy[2 * idx + 1], a, o);
    else o[0] = o[1] = o[2] = __builtin_nan("");
}

static int check_device(int device, const char *who)
{
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        set_error("%s: no HIP device available", who);
        return L3D_ERR_NO_DEVICE;
    }
    if (device < 0 || device >= ndev) {
        set_error("%s: device %d out of range (%d devices)", who, device, ndev);
        return L3D_ERR_INVALID_ARG;
    }
    return L3D_OK;
}

// RAII device buffer for the host-pointer entry points
struct DevBuf {
    void *p = nullptr;
    hipError_t alloc(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 1); }
    ~DevBuf() { if (p) (void)hipFree(p); }
};

} // namespace l3d

using namespace l3d;

extern "C" int l3d_bgr2gray_device(const uint8_t *d_bgr, int width, int height, int batch, uint8_t *d_gray, int device, void *stream)
{
    L3D_REQUIRE(d_bgr && d_gray, "l3d_bgr2gray_device: null argument");
    L3D_REQUIRE(width > 0 && height > 0 && batch > 0, "l3d_bgr2gray_device: bad size");
    L3D_REQUIRE(((uintptr_t)d_bgr & 3) == 0 && ((uintptr_t)d_gray & 3) == 0, "l3d_bgr2gray_device: pointers must be 4-byte aligned");
    int rc = check_device(device, "l3d_bgr2gray_device");
    if (rc) return rc;
    DeviceGuard dg(device);
    const size_t npx = (size_t)width * height * batch;
    const size_t groups = (npx + 3) / 4;
    hipLaunchKernelGGL(k_bgr2gray, dim3((unsigned)((groups + 255) / 256)), dim3(256), 0, (hipStream_t)stream, d_bgr, npx, d_gray);
    L3D_HIP_CHECK(hipGetLastError());
    return L3D_OK;
}

extern "C" int l3d_bgr2gray(const uint8_t *bgr, int width, int height, int batch, uint8_t *gray, int device)
{
    L3D_REQUIRE(bgr && gray, "l3d_bgr2gray: null argument");
    L3D_REQUIRE(width > 0 && height > 0 && batch > 0, "l3d_bgr2gray: bad size");
    int rc = check_device(device, "l3d_bgr2gray");
    if (rc) return rc;
    DeviceGuard dg(device);
    const size_t npx = (size_t)width * height * batch;
    DevBuf in, out;
    L3D_HIP_CHECK(in.alloc(npx * 3));
    L3D_HIP_CHECK(out.alloc(npx));
    L3D_HIP_CHECK(hipMemcpy(in.p, bgr, npx * 3, hipMemcpyHostToDevice));
    rc = l3d_bgr2gray_device((const uint8_t *)in.p, width, height, batch, (uint8_t *)out.p, device, nullptr);
    if (rc) return rc;
    L3D_HIP_CHECK(hipDeviceSynchronize());
    L3D_HIP_CHECK(hipMemcpy(gray, out.p, npx, hipMemcpyDeviceToHost));
    return L3D_OK;
}

extern "C" int l3d_depth_from_disp16_device(const int16_t *d_disp16, const double *Q16_host, int width, int height, int batch,
                                            float *d_depth, float *d_xyz, int device, void *stream)
{
    L3D_REQUIRE(d_disp16 && d_depth, "l3d_depth_from_disp16_device: null argument");
    L3D_REQUIRE(width > 0 && height > 0 && batch > 0, "l3d_depth_from_disp16_device: bad size");
    L3D_REQUIRE(!(d_xyz && !Q16_host), "l3d_depth_from_disp16_device: xyz output needs Q");
    int rc = check_device(device, "l3d_depth_from_disp16_device");
    if (rc) return rc;
    DeviceGuard dg(device);
    QMat Q;
    for (int i = 0; i < 16; i++) Q.q[i] = Q16_host ? Q16_host[i] : 0.0;
    const dim3 grid(cdiv(width, 64), cdiv(height, 4), batch), block(64, 4);
    if (d_xyz)
        hipLaunchKernelGGL(k_depth<true>, grid, block, 0, (hipStream_t)stream, d_disp16, width, height, batch, Q, 1, d_depth, d_xyz);
    else
        hipLaunchKernelGGL(k_depth<false>, grid, block, 0, (hipStream_t)stream, d_disp16, width, height, batch, Q, Q16_host ? 1 : 0, d_depth,
                           d_xyz);
    L3D_HIP_CHECK(hipGetLastError());
    return L3D_OK;
}

// Depth by table look-up for callers that can keep the 65536-entry table (256 KB of device memory): `lut_Q` (17
// doubles) remembers what the table was built for, [16] = 1 when it holds a calibrated table, 0 for the fallback, -1 when
// empty.  Returns L3D_ERR_UNSUPPORTED (and does nothing) when Z depends on the pixel position: the caller then uses
// l3d_depth_from_disp16_device.
int depth_from_disp16_lut(const int16_t *d_disp16, const double *Q16_host, size_t npx, float *d_depth, float *d_lut, double *lut_Q,
                          hipStream_t stream)
{
    if (Q16_host && (Q16_host[8] != 0.0 || Q16_host[9] != 0.0 || Q16_host[12] != 0.0 || Q16_host[13] != 0.0)) return L3D_ERR_UNSUPPORTED;
    bool same = lut_Q[16] == (Q16_host ? 1.0 : 0.0);
    for (int i = 0; i < 16 && same && Q16_host; i++) same = std::memcmp(&lut_Q[i], &Q16_host[i], sizeof(double)) == 0;
    if (!same) {
        QMat Q;
        for (int i = 0; i < 16; i++) Q.q[i] = Q16_host ? Q16_host[i] : 0.0;
        // The table is rebuilt only when Q changes (once per calibration).  Later calls may come on other streams and
        // skip the build, so the build is waited for here: whichever stream applies the table next reads a complete
        // one.  The record of what the table holds is updated only once the build has succeeded.
        lut_Q[16] = -1.0;
        hipLaunchKernelGGL(k_depth_lut_build, dim3(256), dim3(256), 0, stream, Q, Q16_host ? 1 : 0, d_lut);
        L3D_HIP_CHECK(hipGetLastError());
        L3D_HIP_CHECK(hipStreamSynchronize(stream));
        for (int i = 0; i < 16; i++) lut_Q[i] = Q.q[i];
        lut_Q[16] = Q16_host ? 1.0 : 0.0;
    }
    if (npx)   // (npx = 0: only make sure the table is there)
        hipLaunchKernelGGL(k_depth_lut_apply, dim3((unsigned)((npx / 4 + 256) / 256)), dim3(256), 0, stream, d_disp16, npx, d_lut, d_depth);
    L3D_HIP_CHECK(hipGetLastError());
    return L3D_OK;
}

extern "C" int l3d_depth_from_disp16(const int16_t *disp16, const double *Q16, int width, int height, int batch, float *depth,
                                     float *xyz, int device)
{
    L3D_REQUIRE(disp16 && depth, "l3d_depth_from_disp16: null argument");
    L3D_REQUIRE(width > 0 && height > 0 && batch > 0, "l3d_depth_from_disp16: bad size");
    int rc = check_device(device, "l3d_depth_from_disp16");
    if (rc) return rc;
    DeviceGuard dg(device);
    const size_t npx = (size_t)width * height * batch;
    DevBuf in, out, out3;
    L3D_HIP_CHECK(in.alloc(npx * 2));
    L3D_HIP_CHECK(out.alloc(npx * 4));
    if (xyz) L3D_HIP_CHECK(out3.alloc(npx * 12));
    L3D_HIP_CHECK(hipMemcpy(in.p, disp16, npx * 2, hipMemcpyHostToDevice));
    rc = l3d_depth_from_disp16_device((const int16_t *)in.p, Q16, width, height, batch, (float *)out.p, xyz ? (float *)out3.p : nullptr, device, nullptr);
    if (rc) return rc;
    L3D_HIP_CHECK(hipDeviceSynchronize());
    L3D_HIP_CHECK(hipMemcpy(depth, out.p, npx * 4, hipMemcpyDeviceToHost));
    if (xyz) L3D_HIP_CHECK(hipMemcpy(xyz, out3.p, npx * 12, hipMemcpyDeviceToHost));
    return L3D_OK;
}

extern "C" int l3d_reconstruct_from_depth(const double *uv, int n, const float *depth, int width, int height, const double *K9,
                                          double *xyz, int *n_out, int device)
{
    L3D_REQUIRE(n_out, "l3d_reconstruct_from_depth: null argument");
    *n_out = 0;
    L3D_REQUIRE(n >= 0 && width > 0 && height > 0, "l3d_reconstruct_from_depth: bad size");
    if (n == 0) return L3D_OK;
    L3D_REQUIRE(uv && depth && K9 && xyz, "l3d_reconstruct_from_depth: null argument");
    int rc = check_device(device, "l3d_reconstruct_from_depth");
    if (rc) return rc;
    DeviceGuard dg(device);
    DevBuf duv, ddep, dxyz, dkeep;
    L3D_HIP_CHECK(duv.alloc((size_t)n * 16));
    L3D_HIP_CHECK(ddep.alloc((size_t)width * height * 4));
    L3D_HIP_CHECK(dxyz.alloc((size_t)n * 24));
    L3D_HIP_CHECK(dkeep.alloc((size_t)n * 4));
    L3D_HIP_CHECK(hipMemcpy(duv.p, uv, (size_t)n * 16, hipMemcpyHostToDevice));
    L3D_HIP_CHECK(hipMemcpy(ddep.p, depth, (size_t)width * height * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_from_depth, dim3(cdiv(n, 256)), dim3(256), 0, nullptr, (const double *)duv.p, n, (const float *)ddep.p, width, height,
                       K9[0], K9[4], K9[2], K9[5], (double *)dxyz.p, (int *)dkeep.p);
    L3D_HIP_CHECK(hipGetLastError());
    std::vector<double> tmp((size_t)n * 3);
    std::vector<int> keep(n);
    L3D_HIP_CHECK(hipMemcpy(tmp.data(), dxyz.p, (size_t)n * 24, hipMemcpyDeviceToHost));
    L3D_HIP_CHECK(hipMemcpy(keep.data(), dkeep.p, (size_t)n * 4, hipMemcpyDeviceToHost));
    int m = 0;
    for (int i = 0; i < n; i++)
        if (keep[i]) {
            xyz[3 * m] = tmp[3 * i]; xyz[3 * m + 1] = tmp[3 * i + 1]; xyz[3 * m + 2] = tmp[3 * i + 2];
            m++;
        }
    *n_out = m;
    return L3D_OK;
}

extern "C" int l3d_reconstruct_plane(const double *uv, int n, const double *Kinv9, const double *plane4, int refraction, double n_water,
                                     double *xyz, int device)
{
    L3D_REQUIRE(n >= 0, "l3d_reconstruct_plane: bad size");
    if (n == 0) return L3D_OK;
    L3D_REQUIRE(uv && Kinv9 && plane4 && xyz, "l3d_reconstruct_plane: null argument");
    int rc = check_device(device, "l3d_reconstruct_plane");
    if (rc) return rc;
    DeviceGuard dg(device);
    DevBuf duv, dxyz;
    L3D_HIP_CHECK(duv.alloc((size_t)n * 16));
    L3D_HIP_CHECK(dxyz.alloc((size_t)n * 24));
    L3D_HIP_CHECK(hipMemcpy(duv.p, uv, (size_t)n * 16, hipMemcpyHostToDevice));
    PlaneArgs a;
    for (int i = 0; i < 9; i++) a.kinv[i] = Kinv9[i];
    for (int i = 0; i < 4; i++) a.plane[i] = plane4[i];
    a.n_water = n_water;
    a.refraction = refraction;
    hipLaunchKernelGGL(k_plane, dim3(cdiv(n, 256)), dim3(256), 0, nullptr, (const double *)duv.p, n, a, (double *)dxyz.p);
    L3D_HIP_CHECK(hipGetLastError());
    L3D_HIP_CHECK(hipMemcpy(xyz, dxyz.p, (size_t)n * 24, hipMemcpyDeviceToHost));
    return L3D_OK;
}

extern "C" int l3d_reconstruct_from_depth_device(const float *d_xy, const int *d_counts, int capacity, int batch, const float *d_depth,
                                                 int width, int height, const double *K9, double *d_xyz, int *d_keep, void *stream)
{
    L3D_REQUIRE(capacity >= 0 && batch >= 1 && width > 0 && height > 0, "l3d_reconstruct_from_depth_device: bad size");
    if (capacity == 0) return L3D_OK;
    L3D_REQUIRE(d_xy && d_counts && d_depth && K9 && d_xyz && d_keep, "l3d_reconstruct_from_depth_device: null argument");
    hipLaunchKernelGGL(k_from_depth_batch, dim3(cdiv(capacity, 256), batch), dim3(256), 0, (hipStream_t)stream, d_xy, d_counts, capacity, d_depth,
                       width, height, K9[0], K9[4], K9[2], K9[5], d_xyz, d_keep);
    L3D_HIP_CHECK(hipGetLastError());
    return L3D_OK;
}

extern "C" int l3d_reconstruct_from_depth_uv64_device(const double *d_uv, int n, const float *d_depth, int width, int height,
                                                      const double *K9, double *d_xyz, int *d_keep, void *stream)
{
    L3D_REQUIRE(n >= 0 && width > 0 && height > 0, "l3d_reconstruct_from_depth_uv64_device: bad size");
    if (n == 0) return L3D_OK;
    L3D_REQUIRE(d_uv && d_depth && K9 && d_xyz && d_keep, "l3d_reconstruct_from_depth_uv64_device: null argument");
    hipLaunchKernelGGL(k_from_depth, dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, d_uv, n, d_depth, width, height, K9[0], K9[4],
                       K9[2], K9[5], d_xyz, d_keep);
    L3D_HIP_CHECK(hipGetLastError());
    return L3D_OK;
}

extern "C" int l3d_reconstruct_plane_device(const float *d_xy, const int *d_counts, int capacity, int batch, const double *Kinv9,
                                            const double *plane4, int refraction, double n_water, double *d_xyz, void *stream)
{
    L3D_REQUIRE(capacity >= 0 && batch >= 1, "l3d_reconstruct_plane_device: bad size");
    if (capacity == 0) return L3D_OK;
    L3D_REQUIRE(d_xy && d_counts && Kinv9 && plane4 && d_xyz, "l3d_reconstruct_plane_device: null argument");
    PlaneArgs a;
    for (int i = 0; i < 9; i++) a.kinv[i] = Kinv9[i];
    for (int i = 0; i < 4; i++) a.plane[i] = plane4[i];
    a.n_water = n_water;
    a.refraction = refraction;
    hipLaunchKernelGGL(k_plane_batch, dim3(cdiv(capacity, 256), batch), dim3(256), 0, (hipStream_t)stream, d_xy, d_counts, capacity, a, d_xyz);
    L3D_HIP_CHECK(hipGetLastError());
    return L3D_OK;
}

static TriArgs tri_args(const double *Q, int W, int H, int window, double min_disparity)
{
    TriArgs a;
    a.fx = Q[2 * 4 + 3]; a.baseline = 1.0 / Q[3 * 4 + 2]; a.cx = -Q[0 * 4 + 3]; a.cy = -Q[1 * 4 + 3];
    a.min_disparity = min_disparity; a.W = W; a.H = H; a.window = window;
    return a;
}

extern "C" int l3d_triangulate_disparity(const double *uv, int n, const void *disparity, int disparity_is_f64, int width, int height,
                                         const double *Q16, int window_size, double min_disparity, float *xyz, int *n_out, int device)
{
    L3D_REQUIRE(n_out, "l3d_triangulate_disparity: null argument");
    *n_out = 0;
    L3D_REQUIRE(n >= 0 && width > 0 && height > 0, "l3d_triangulate_disparity: bad size");
    L3D_REQUIRE(window_size == 0 || (window_size >= 1 && window_size <= 31), "l3d_triangulate_disparity: window_size %d outside [1, 31] (0 = nearest pixel)", window_size);
    if (n == 0) return L3D_OK;
    L3D_REQUIRE(uv && disparity && Q16 && xyz, "l3d_triangulate_disparity: null argument");
    int rc = check_device(device, "l3d_triangulate_disparity");
    if (rc) return rc;
    DeviceGuard dg(device);
    const size_t esz = disparity_is_f64 ? 8 : 4;
    DevBuf duv, ddisp, dxyz, dkeep;
    L3D_HIP_CHECK(duv.alloc((size_t)n * 16));
    L3D_HIP_CHECK(ddisp.alloc((size_t)width * height * esz));
    L3D_HIP_CHECK(dxyz.alloc((size_t)n * 12));
    L3D_HIP_CHECK(dkeep.alloc((size_t)n * 4));
    L3D_HIP_CHECK(hipMemcpy(duv.p, uv, (size_t)n * 16, hipMemcpyHostToDevice));
    L3D_HIP_CHECK(hipMemcpy(ddisp.p, disparity, (size_t)width * height * esz, hipMemcpyHostToDevice));
    const TriArgs a = tri_args(Q16, width, height, window_size, min_disparity);
    if (disparity_is_f64)
        hipLaunchKernelGGL(k_triangulate<double>, dim3(cdiv(n, 256)), dim3(256), 0, nullptr, (const double *)duv.p, n, (const double *)ddisp.p, a,
                           (float *)dxyz.p, (int *)dkeep.p);
    else
        hipLaunchKernelGGL(k_triangulate<float>, dim3(cdiv(n, 256)), dim3(256), 0, nullptr, (const double *)duv.p, n, (const float *)ddisp.p, a,
                           (float *)dxyz.p, (int *)dkeep.p);
    L3D_HIP_CHECK(hipGetLastError());
    std::vector<float> tmp((size_t)n * 3);
    std::vector<int> keep(n);
    L3D_HIP_CHECK(hipMemcpy(tmp.data(), dxyz.p, (size_t)n * 12, hipMemcpyDeviceToHost));
    L3D_HIP_CHECK(hipMemcpy(keep.data(), dkeep.p, (size_t)n * 4, hipMemcpyDeviceToHost));
    int m = 0;
    for (int i = 0; i < n; i++)
        if (keep[i]) {
            xyz[3 * m] = tmp[3 * i]; xyz[3 * m + 1] = tmp[3 * i + 1]; xyz[3 * m + 2] = tmp[3 * i + 2];
            m++;
        }
    *n_out = m;
    return L3D_OK;
}

extern "C" int l3d_triangulate_disparity_device(const double *d_uv, int n, const float *d_disparity, int width, int height, const double *Q16,
                                                int window_size, double min_disparity, float *d_xyz, int *d_keep, void *stream)
{
    L3D_REQUIRE(n >= 0 && width > 0 && height > 0, "l3d_triangulate_disparity_device: bad size");
    L3D_REQUIRE(window_size == 0 || (window_size >= 1 && window_size <= 31), "l3d_triangulate_disparity_device: window_size %d outside [1, 31]", window_size);
    if (n == 0) return L3D_OK;
    L3D_REQUIRE(d_uv && d_disparity && Q16 && d_xyz && d_keep, "l3d_triangulate_disparity_device: null argument");
    const TriArgs a = tri_args(Q16, width, height, window_size, min_disparity);
    hipLaunchKernelGGL(k_triangulate<float>, dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, d_uv, n, d_disparity, a, d_xyz, d_keep);
    L3D_HIP_CHECK(hipGetLastError());
    return L3D_OK;
}

extern "C" int l3d_laser_depth_map(const double *uv, int n, const void *disparity, int disparity_is_f64, int width, int height,
                                   const double *Q16, float *depth, int device)
{
    L3D_REQUIRE(n >= 0 && width > 0 && height > 0 && depth && Q16 && (n == 0 || (uv && disparity)), "l3d_laser_depth_map: bad argument");
    int rc = check_device(device, "l3d_laser_depth_map");
    if (rc) return rc;
    DeviceGuard dg(device);
    const size_t esz = disparity_is_f64 ? 8 : 4, px = (size_t)width * height;
    DevBuf duv, ddisp, ddep;
    L3D_HIP_CHECK(ddep.alloc(px * 4));
    L3D_HIP_CHECK(hipMemset(ddep.p, 0, px * 4));
    if (n > 0) {
        L3D_HIP_CHECK(duv.alloc((size_t)n * 16));
        L3D_HIP_CHECK(ddisp.alloc(px * esz));
        L3D_HIP_CHECK(hipMemcpy(duv.p, uv, (size_t)n * 16, hipMemcpyHostToDevice));
        L3D_HIP_CHECK(hipMemcpy(ddisp.p, disparity, px * esz, hipMemcpyHostToDevice));
        const TriArgs a = tri_args(Q16, width, height, 0, 1.0);
        if (disparity_is_f64)
            hipLaunchKernelGGL(k_laser_depth<double>, dim3(cdiv(n, 256)), dim3(256), 0, nullptr, (const double *)duv.p, n, (const double *)ddisp.p, a, (float *)ddep.p);
        else
            hipLaunchKernelGGL(k_laser_depth<float>, dim3(cdiv(n, 256)), dim3(256), 0, nullptr, (const double *)duv.p, n, (const float *)ddisp.p, a, (float *)ddep.p);
        L3D_HIP_CHECK(hipGetLastError());
    }
    L3D_HIP_CHECK(hipMemcpy(depth, ddep.p, px * 4, hipMemcpyDeviceToHost));
    return L3D_OK;
}
