// Point-cloud post-processing for gfx950: voxel-grid downsampling and the nearest-neighbour statistics
// behind the outlier filter and the spacing metric.
//
// Replaces PointCloudProcessor._simple_voxel_downsample, ._simple_outlier_removal and the k-d tree query of
// .compute_point_cloud_metrics of the reference (utils/point_cloud.py:55-79, 108-132, 270-276), i.e. the
// behaviour of PointCloudProcessor when Open3D is not installed.  Results are bit-identical to that code:
//   - voxel means are accumulated in input order and emitted in order of each voxel's first point.  Points are
//     grouped by a stable radix sort of (voxel key, input index) -- rocPRIM through hipCUB; a sort is a library
//     primitive, not a kernel worth hand-writing here -- then one thread per voxel adds its points in order;
//   - neighbour distances are brute force over LDS tiles of the cloud (N^2 pairs in double: a 100 k point cloud
//     is 10^10 pair evaluations), each thread keeping its k smallest squared distances sorted; mean and standard
//     deviation follow numpy's pairwise summation order.
#include "common.h"
#include <hipcub/hipcub.hpp>
#include <cmath>
#include <new>
#include <vector>

namespace l3d {

struct Bounds {
    long long lo[3], hi[3];
};

__global__ void __launch_bounds__(256) k_voxel_index(const double *__restrict__ pts, int n, double voxel, long long *__restrict__ vidx, Bounds *bounds)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    long long k[3] = {0, 0, 0};
    const bool on = i < n;
    if (on) {
#pragma unroll
        for (int c = 0; c < 3; c++) {
            k[c] = (long long)floor(pts[3 * (size_t)i + c] / voxel);
            vidx[3 * (size_t)i + c] = k[c];
        }
    }
    // per-wave min / max before touching the global bounds
#pragma unroll
    for (int c = 0; c < 3; c++) {
        long long lo = on ? k[c] : LLONG_MAX, hi = on ? k[c] : LLONG_MIN;
        for (int o = 32; o >= 1; o >>= 1) {
            const long long l2 = __shfl_xor(lo, o), h2 = __shfl_xor(hi, o);
            lo = l2 < lo ? l2 : lo;
            hi = h2 > hi ? h2 : hi;
        }
        if ((threadIdx.x & 63) == 0 && lo <= hi) {
            atomicMin(&bounds->lo[c], lo);
            atomicMax(&bounds->hi[c], hi);
        }
    }
}

__global__ void __launch_bounds__(256) k_voxel_key(const long long *__restrict__ vidx, int n, Bounds b, int bits_y, int bits_z,
                                                   unsigned long long *__restrict__ key, int *__restrict__ idx)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const unsigned long long x = (unsigned long long)(vidx[3 * (size_t)i] - b.lo[0]), y = (unsigned long long)(vidx[3 * (size_t)i + 1] - b.lo[1]),
                             z = (unsigned long long)(vidx[3 * (size_t)i + 2] - b.lo[2]);
    key[i] = (((x << bits_y) | y) << bits_z) | z;
    idx[i] = i;
}

// one thread per voxel (= per run of equal keys in the sorted order): mean of its points, added in input order
__global__ void __launch_bounds__(256) k_voxel_mean(const double *__restrict__ pts, int n, const unsigned long long *__restrict__ skey,
                                                    const int *__restrict__ sidx, double *__restrict__ mean_at_head, int *__restrict__ is_head)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const unsigned long long k = skey[i];
    if (i > 0 && skey[i - 1] == k) return;
    const int head = sidx[i];
    const double *p = pts + 3 * (size_t)head;
    double ax = p[0], ay = p[1], az = p[2];
    int cnt = 1;
    for (int j = i + 1; j < n && skey[j] == k; j++) {
        const double *q = pts + 3 * (size_t)sidx[j];
        ax += q[0]; ay += q[1]; az += q[2];
        cnt++;
    }
    mean_at_head[3 * (size_t)head] = ax / (double)cnt;
    mean_at_head[3 * (size_t)head + 1] = ay / (double)cnt;
    mean_at_head[3 * (size_t)head + 2] = az / (double)cnt;
    is_head[head] = 1;
}

__global__ void __launch_bounds__(256) k_compact3(const double *__restrict__ src, const int *__restrict__ flag, const int *__restrict__ pos, int n,
                                                  double *__restrict__ dst)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n || !flag[i]) return;
    const size_t o = 3 * (size_t)pos[i];
    dst[o] = src[3 * (size_t)i]; dst[o + 1] = src[3 * (size_t)i + 1]; dst[o + 2] = src[3 * (size_t)i + 2];
}

constexpr int KNN_MAX = 65; // nb_neighbors + 1

// numpy's pairwise summation (np.add.reduce of a contiguous double array) for n <= 128
__device__ double np_pairwise_sum(const double *a, int n)
{
    if (n < 8) {
        double res = 0.;
        for (int i = 0; i < n; i++) res += a[i];
        return res;
    }
    double r[8];
    int i;
    for (i = 0; i < 8; i++) r[i] = a[i];
    for (i = 8; i < n - (n % 8); i += 8)
        for (int j = 0; j < 8; j++) r[j] += a[i + j];
    double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    for (; i < n; i++) res += a[i];
    return res;
}

// k smallest distances (the point itself included) of every point, brute force over LDS tiles.
// mode 0: nn[i] = distance to the nearest other point (k = 2);
// mode 1: keep[i] = the reference's predicate on mean and standard deviation of the nb = k - 1 neighbour distances.
__global__ void __launch_bounds__(256) k_knn_stats(const double *__restrict__ pts, int n, int k, int nb, double std_ratio, int mode,
                                                   double *__restrict__ nn, int *__restrict__ keep)
{
    __shared__ double tile[256 * 3];
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const bool on = i < n;
    double px = 0, py = 0, pz = 0;
    if (on) { px = pts[3 * (size_t)i]; py = pts[3 * (size_t)i + 1]; pz = pts[3 * (size_t)i + 2]; }
    double best[KNN_MAX];
    int m = 0;
    for (int base = 0; base < n; base += 256) {
        const int cnt = min(256, n - base);
        __syncthreads();
        for (int t = threadIdx.x; t < cnt * 3; t += 256) tile[t] = pts[3 * (size_t)base + t];
        __syncthreads();
        if (!on) continue;
        for (int j = 0; j < cnt; j++) {
            double s = 0.;
            const double dx = px - tile[3 * j], dy = py - tile[3 * j + 1], dz = pz - tile[3 * j + 2];
            s += dx * dx; s += dy * dy; s += dz * dz;
            if (m < k || s < best[m - 1]) {
                int p = m < k ? m++ : k - 1;
                while (p > 0 && best[p - 1] > s) { best[p] = best[p - 1]; p--; }
                best[p] = s;
            }
        }
    }
    if (!on) return;
    for (int j = 0; j < m; j++) best[j] = sqrt(best[j]);
    if (mode == 0) { nn[i] = best[1]; return; }
    for (int j = m; j < nb + 1; j++) best[j] = INFINITY; // a k-d tree query pads missing neighbours with inf
    const double *d = best + 1;
    const double mean = np_pairwise_sum(d, nb) / (double)nb;
    double dev[KNN_MAX];
    for (int j = 0; j < nb; j++) { const double x = d[j] - mean; dev[j] = x * x; }
    const double sd = sqrt(np_pairwise_sum(dev, nb) / (double)nb);
    keep[i] = mean < (mean + std_ratio * sd);
}

struct DevMem {
    void *p = nullptr;
    ~DevMem() { if (p) (void)hipFree(p); }
    hipError_t alloc(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 1); }
    template <typename T> T *as() { return (T *)p; }
};

static int pc_check_device(int device, const char *who)
{
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        set_error("%s: no HIP device available", who);
        return L3D_ERR_NO_DEVICE;
    }
    L3D_REQUIRE(device >= 0 && device < ndev, "%s: device %d out of range (%d devices)", who, device, ndev);
    return L3D_OK;
}

static int bits_for(unsigned long long range) { int b = 1; while (b < 63 && (range >> b)) b++; return b; }

} // namespace l3d

using namespace l3d;

extern "C" int l3d_voxel_downsample(const double *points, int n, double voxel_size, double *out, int *n_out, int device)
{
    L3D_REQUIRE(n_out, "l3d_voxel_downsample: null argument");
    *n_out = 0;
    L3D_REQUIRE(n >= 0 && voxel_size > 0, "l3d_voxel_downsample: bad argument (n=%d, voxel_size=%g)", n, voxel_size);
    if (n == 0) return L3D_OK;
    L3D_REQUIRE(points && out, "l3d_voxel_downsample: null argument");
    int rc = pc_check_device(device, "l3d_voxel_downsample");
    if (rc) return rc;
    DeviceGuard dg(device);
    const dim3 grid(cdiv(n, 256)), blk(256);
    DevMem dpts, dvidx, dbounds, dkey, dskey, didx, dsidx, dmean, dflag, dpos, dout, dtmp;
    L3D_HIP_CHECK(dpts.alloc((size_t)n * 24));
    L3D_HIP_CHECK(dvidx.alloc((size_t)n * 24));
    L3D_HIP_CHECK(dbounds.alloc(sizeof(Bounds)));
    L3D_HIP_CHECK(hipMemcpy(dpts.p, points, (size_t)n * 24, hipMemcpyHostToDevice));
    Bounds b;
    for (int c = 0; c < 3; c++) { b.lo[c] = LLONG_MAX; b.hi[c] = LLONG_MIN; }
    L3D_HIP_CHECK(hipMemcpy(dbounds.p, &b, sizeof(b), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_voxel_index, grid, blk, 0, nullptr, dpts.as<double>(), n, voxel_size, dvidx.as<long long>(), dbounds.as<Bounds>());
    L3D_HIP_CHECK(hipGetLastError());
    L3D_HIP_CHECK(hipMemcpy(&b, dbounds.p, sizeof(b), hipMemcpyDeviceToHost));
    int bits[3];
    for (int c = 0; c < 3; c++) {
        if (b.hi[c] < b.lo[c] || (unsigned long long)(b.hi[c] - b.lo[c]) >= (1ull << 40)) {
            set_error("l3d_voxel_downsample: non-finite coordinates or a voxel grid of more than 2^40 cells along an axis");
            return L3D_ERR_UNSUPPORTED;
        }
        bits[c] = bits_for((unsigned long long)(b.hi[c] - b.lo[c]));
    }
    if (bits[0] + bits[1] + bits[2] > 64) {
        set_error("l3d_voxel_downsample: the voxel grid needs %d key bits (64 available); use a larger voxel_size", bits[0] + bits[1] + bits[2]);
        return L3D_ERR_UNSUPPORTED;
    }
    L3D_HIP_CHECK(dkey.alloc((size_t)n * 8));
    L3D_HIP_CHECK(dskey.alloc((size_t)n * 8));
    L3D_HIP_CHECK(didx.alloc((size_t)n * 4));
    L3D_HIP_CHECK(dsidx.alloc((size_t)n * 4));
    hipLaunchKernelGGL(k_voxel_key, grid, blk, 0, nullptr, dvidx.as<long long>(), n, b, bits[1], bits[2], dkey.as<unsigned long long>(), didx.as<int>());
    size_t tmp_bytes = 0;
    const int end_bit = bits[0] + bits[1] + bits[2];
    L3D_HIP_CHECK(hipcub::DeviceRadixSort::SortPairs(nullptr, tmp_bytes, dkey.as<unsigned long long>(), dskey.as<unsigned long long>(), didx.as<int>(),
                                                     dsidx.as<int>(), n, 0, end_bit, (hipStream_t) nullptr));
    size_t scan_bytes = 0;
    L3D_HIP_CHECK(dflag.alloc((size_t)n * 4));
    L3D_HIP_CHECK(dpos.alloc((size_t)n * 4));
    L3D_HIP_CHECK(hipcub::DeviceScan::ExclusiveSum(nullptr, scan_bytes, dflag.as<int>(), dpos.as<int>(), n, (hipStream_t) nullptr));
    L3D_HIP_CHECK(dtmp.alloc(std::max(tmp_bytes, scan_bytes)));
    L3D_HIP_CHECK(hipcub::DeviceRadixSort::SortPairs(dtmp.p, tmp_bytes, dkey.as<unsigned long long>(), dskey.as<unsigned long long>(), didx.as<int>(),
                                                     dsidx.as<int>(), n, 0, end_bit, (hipStream_t) nullptr));
    L3D_HIP_CHECK(dmean.alloc((size_t)n * 24));
    L3D_HIP_CHECK(dout.alloc((size_t)n * 24));
    L3D_HIP_CHECK(hipMemset(dflag.p, 0, (size_t)n * 4));
    hipLaunchKernelGGL(k_voxel_mean, grid, blk, 0, nullptr, dpts.as<double>(), n, dskey.as<unsigned long long>(), dsidx.as<int>(), dmean.as<double>(),
                       dflag.as<int>());
    L3D_HIP_CHECK(hipcub::DeviceScan::ExclusiveSum(dtmp.p, scan_bytes, dflag.as<int>(), dpos.as<int>(), n, (hipStream_t) nullptr));
    hipLaunchKernelGGL(k_compact3, grid, blk, 0, nullptr, dmean.as<double>(), dflag.as<int>(), dpos.as<int>(), n, dout.as<double>());
    L3D_HIP_CHECK(hipGetLastError());
    int last_pos = 0, last_flag = 0;
    L3D_HIP_CHECK(hipMemcpy(&last_pos, dpos.as<int>() + (n - 1), 4, hipMemcpyDeviceToHost));
    L3D_HIP_CHECK(hipMemcpy(&last_flag, dflag.as<int>() + (n - 1), 4, hipMemcpyDeviceToHost));
    const int m = last_pos + last_flag;
    L3D_HIP_CHECK(hipMemcpy(out, dout.p, (size_t)m * 24, hipMemcpyDeviceToHost));
    *n_out = m;
    return L3D_OK;
}

extern "C" int l3d_sor_simple(const double *points, int n, int nb_neighbors, double std_ratio, double *out, int *n_out, int device)
{
    L3D_REQUIRE(n_out, "l3d_sor_simple: null argument");
    *n_out = 0;
    L3D_REQUIRE(n >= 0 && nb_neighbors >= 1 && nb_neighbors < KNN_MAX, "l3d_sor_simple: nb_neighbors %d outside [1, %d]", nb_neighbors, KNN_MAX - 1);
    if (n == 0) return L3D_OK;
    L3D_REQUIRE(points && out, "l3d_sor_simple: null argument");
    if (n < nb_neighbors) { // the reference returns such a cloud unchanged
        std::copy(points, points + 3 * (size_t)n, out);
        *n_out = n;
        return L3D_OK;
    }
    int rc = pc_check_device(device, "l3d_sor_simple");
    if (rc) return rc;
    DeviceGuard dg(device);
    DevMem dpts, dkeep;
    L3D_HIP_CHECK(dpts.alloc((size_t)n * 24));
    L3D_HIP_CHECK(dkeep.alloc((size_t)n * 4));
    L3D_HIP_CHECK(hipMemcpy(dpts.p, points, (size_t)n * 24, hipMemcpyHostToDevice));
    const int k = std::min(nb_neighbors + 1, n);
    hipLaunchKernelGGL(k_knn_stats, dim3(cdiv(n, 256)), dim3(256), 0, nullptr, dpts.as<double>(), n, k, nb_neighbors, std_ratio, 1, (double *)nullptr,
                       dkeep.as<int>());
    L3D_HIP_CHECK(hipGetLastError());
    std::vector<int> keep(n);
    L3D_HIP_CHECK(hipMemcpy(keep.data(), dkeep.p, (size_t)n * 4, hipMemcpyDeviceToHost));
    int m = 0;
    for (int i = 0; i < n; i++)
        if (keep[i]) {
            out[3 * (size_t)m] = points[3 * (size_t)i]; out[3 * (size_t)m + 1] = points[3 * (size_t)i + 1]; out[3 * (size_t)m + 2] = points[3 * (size_t)i + 2];
            m++;
        }
    *n_out = m;
    return L3D_OK;
}

extern "C" int l3d_nn_distances(const double *points, int n, double *nn, int device)
{
    L3D_REQUIRE(n >= 2 && points && nn, "l3d_nn_distances: needs at least two points and non-null arrays");
    int rc = pc_check_device(device, "l3d_nn_distances");
    if (rc) return rc;
    DeviceGuard dg(device);
    DevMem dpts, dnn;
    L3D_HIP_CHECK(dpts.alloc((size_t)n * 24));
    L3D_HIP_CHECK(dnn.alloc((size_t)n * 8));
    L3D_HIP_CHECK(hipMemcpy(dpts.p, points, (size_t)n * 24, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_knn_stats, dim3(cdiv(n, 256)), dim3(256), 0, nullptr, dpts.as<double>(), n, 2, 1, 0.0, 0, dnn.as<double>(), (int *)nullptr);
    L3D_HIP_CHECK(hipGetLastError());
    L3D_HIP_CHECK(hipMemcpy(nn, dnn.p, (size_t)n * 8, hipMemcpyDeviceToHost));
    return L3D_OK;
}
