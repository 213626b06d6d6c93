// Shared host-side helpers of libl3d_hip.so: error reporting and launch checks.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdio>
#include <cstdint>
#include <cstddef>
#include "../../include/l3d.h"

namespace l3d {

// thread-local error text behind l3d_last_error()
void set_error(const char *fmt, ...);

#define L3D_HIP_CHECK(expr)                                                                        \
    do {                                                                                           \
        hipError_t _e = (expr);                                                                    \
        if (_e != hipSuccess) {                                                                    \
            ::l3d::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
            return L3D_ERR_HIP;                                                                    \
        }                                                                                          \
    } while (0)

#define L3D_REQUIRE(cond, ...)                                                                     \
    do {                                                                                           \
        if (!(cond)) {                                                                             \
            ::l3d::set_error(__VA_ARGS__);                                                         \
            return L3D_ERR_INVALID_ARG;                                                            \
        }                                                                                          \
    } while (0)

// Selects `device` for the lifetime of the object and restores the previous one.
struct DeviceGuard {
    int prev = -1;
    bool ok = true;
    explicit DeviceGuard(int device)
    {
        if (hipGetDevice(&prev) != hipSuccess) { ok = false; return; }
        if (prev != device && hipSetDevice(device) != hipSuccess) ok = false;
    }
    ~DeviceGuard()
    {
        if (prev >= 0) (void)hipSetDevice(prev);
    }
};

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }

#ifdef __HIPCC__
// Workgroups are handed to the 8 XCDs round-robin by their linear id, and every XCD has its own L2.  Mapping the
// hardware id through this permutation gives each XCD a contiguous range of logical ids, so blocks that read
// the same rows or columns (neighbours in logical order, running at about the same time) share one L2.
// Ids past the last full round of 8 map to themselves.
__device__ __forceinline__ int xcd_contiguous(int id, int n)
{
    const int per = n >> 3;
    return id < per * 8 ? (id & 7) * per + (id >> 3) : id;
}
// The logical block index of a 3-D grid (x fastest) under that mapping; use it in place of blockIdx.
__device__ __forceinline__ dim3 xcd_block()
{
    const int gx = gridDim.x, gxy = gridDim.x * gridDim.y;
    const int lin = xcd_contiguous(blockIdx.x + gx * blockIdx.y + gxy * blockIdx.z, gxy * gridDim.z);
    const int z = lin / gxy, r = lin - z * gxy, y = r / gx;
    return dim3(r - y * gx, y, z);
}

// Raw buffer access: descriptor in scalar registers, per-lane byte offset, uniform byte offset.
typedef __amdgpu_buffer_rsrc_t BufRsrc;
__device__ __forceinline__ BufRsrc make_rsrc(const void *base)
{
    // raw buffer (stride 0), range 2 GiB from `base`, default data format of the gfx9 family
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), 0, 0x7fffffff, 0x00020000);
}
#endif

} // namespace l3d

// Disparity -> depth by table look-up (depth.hip); L3D_ERR_UNSUPPORTED when Z depends on the pixel position.
int depth_from_disp16_lut(const int16_t *d_disp16, const double *Q16_host, size_t npx, float *d_depth, float *d_lut, double *lut_Q,
                          hipStream_t stream);

// Colour + brightness mask after the 3x3 close and open (stages shared by SimpleLaserExtractor and
// HybridLaserExtractor); *d_mask points into the handle and stays valid until its next use.  Defined in simple.hip.
int simple_clean_mask(l3d_simple *h, const uint8_t *d_bgr, int W, int H, const int *lo, const int *hi, int thr, hipStream_t s,
                      const uint8_t **d_mask);

