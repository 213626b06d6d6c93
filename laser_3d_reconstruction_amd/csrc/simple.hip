// HSV / brightness threshold laser extractor for gfx950.
//
// Replaces SimpleLaserExtractor.extract_centerline of the reference (core/laser_extractor.py:45-100):
// cv2.cvtColor(BGR2HSV) + inRange, cv2.cvtColor(BGR2GRAY) + threshold, AND, 3x3 CLOSE then OPEN,
// findContours(RETR_EXTERNAL) + contourArea > min_area + drawContours(filled), per-row
// intensity-weighted centroid.
//
// The contour stage is sequential border following in OpenCV.  Here it is restated as data-parallel
// steps that give the same filled mask:
//   1. union-find labelling of the mask (8-connected foreground, 4-connected background, one extra
//      node for "outside the image");
//   2. the shoelace area of a component's outer border = sum over all 2x2 pixel cells of the
//      directed border edge(s) the cell contains, restricted to edges whose background side
//      belongs to the outside component (3 foreground corners: the diagonal across the missing
//      corner; 2 adjacent: the axis edge; 2 diagonal: one edge per direction) -- accumulated with
//      64-bit atomics per component;
//   3. a component is kept when it touches the outside background and |area| / 2 > min_area;
//   4. drawContours(filled) of the kept components = every pixel that a 4-connected flood from
//      outside the image cannot reach without crossing a kept component: a second labelling.
#include "common.h"
#include <algorithm>
#include <new>
#include <vector>

namespace l3d {

__device__ __forceinline__ int s_uf_load(const int *L, int i) { return __hip_atomic_load(L + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ int s_uf_find(const int *L, int i)
{
    int r = i, p;
    while ((p = s_uf_load(L, r)) != r) r = p;
    return r;
}
__device__ __forceinline__ void s_uf_union(int *L, int a, int b)
{
    while (true) {
        a = s_uf_find(L, a);
        b = s_uf_find(L, b);
        if (a == b) return;
        if (a > b) { const int t = a; a = b; b = t; }
        const int old = atomicMin(L + b, a);
        if (old == b) return;
        b = old;
    }
}

struct HsvRange {
    int lo[3], hi[3];
    int thr;
};

// 8-bit BGR -> HSV (H in [0,180)) with 12-bit fixed-point division tables, and gray; writes the
// combined mask (255 / 0) and the gray image
__global__ void __launch_bounds__(256) k_simple_mask(const uint8_t *__restrict__ bgr, size_t n, HsvRange r, const int *__restrict__ sdiv,
                                                     const int *__restrict__ hdiv, uint8_t *__restrict__ gray, uint8_t *__restrict__ mask)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int b = bgr[3 * i], g = bgr[3 * i + 1], rr = bgr[3 * i + 2];
    const int v = max(b, max(g, rr)), vmin = min(b, min(g, rr)), diff = v - vmin;
    const int vr = (v == rr) ? -1 : 0, vg = (v == g) ? -1 : 0;
    const int s = ((diff * sdiv[v] + (1 << 11)) >> 12) & 255;
    int h = (vr & (g - b)) + (~vr & ((vg & (b - rr + 2 * diff)) + ((~vg) & (rr - g + 4 * diff))));
    h = (h * hdiv[diff] + (1 << 11)) >> 12;
    if (h < 0) h += 180;
    h = min(max(h, 0), 255);
    const int y = (rr * 9798 + g * 19235 + b * 3735 + (1 << 14)) >> 15;
    gray[i] = (uint8_t)y;
    const bool in = h >= r.lo[0] && h <= r.hi[0] && s >= r.lo[1] && s <= r.hi[1] && v >= r.lo[2] && v <= r.hi[2];
    mask[i] = (in && y > r.thr) ? 255 : 0;
}

// morphologyEx(CLOSE) then morphologyEx(OPEN) with the 3x3 rectangle = dilate, erode, erode, dilate, in one kernel: a
// block owns MT_W x MT_H output pixels and stages them with a margin of 4 (one per pass) in LDS as bytes; a pass is a row
// step (3 neighbours along x) into the second buffer and a column step back.  Pixels outside the image never win: they
// hold the neutral value of the pass that reads them (0 before a dilation, 255 before an erosion).  (As four launches of
// a thread-per-pixel kernel with nine byte loads the stage took 1.4 ms per 256 frames of 640x360, a third of the extractor.)
constexpr int MT_W = 64, MT_H = 16, MT_M = 4, MT_PW = MT_W + 2 * MT_M, MT_PH = MT_H + 2 * MT_M;

__global__ void __launch_bounds__(256) k_close_open(const uint8_t *__restrict__ src, int W, int H, uint8_t *__restrict__ dst)
{
    __shared__ uint8_t A[MT_PH][MT_PW + 4], Bf[MT_PH][MT_PW + 4];
    const int x0 = blockIdx.x * MT_W - MT_M, y0 = blockIdx.y * MT_H - MT_M;
    src += (size_t)blockIdx.z * W * H;      // image of a stack
    dst += (size_t)blockIdx.z * W * H;
    for (int i = threadIdx.x; i < MT_PW * MT_PH; i += 256) {
        const int ty = i / MT_PW, tx = i - ty * MT_PW, x = x0 + tx, y = y0 + ty;
        A[ty][tx] = (x >= 0 && x < W && y >= 0 && y < H) ? src[(size_t)y * W + x] : 0;
    }
    __syncthreads();
    // pass p shrinks the valid region of the tile by one on every side
#pragma unroll
    for (int p = 0; p < 4; p++) {
        const bool dilate = p == 0 || p == 3;
        const int lo = p + 1, wx = MT_PW - 2 * lo, hy = MT_PH - 2 * p;     // row step: columns lo .. , all rows still valid in y
        for (int i = threadIdx.x; i < wx * hy; i += 256) {
            const int ty = p + i / wx, tx = lo + i % wx;
            const int a = A[ty][tx - 1], b2 = A[ty][tx], c = A[ty][tx + 1];
            Bf[ty][tx] = (uint8_t)(dilate ? max(a, max(b2, c)) : min(a, min(b2, c)));
        }
        __syncthreads();
        const int hy2 = MT_PH - 2 * lo;
        const int next_neutral = (p == 0 || p == 1) ? 255 : 0;            // what the next pass wants outside the image
        for (int i = threadIdx.x; i < wx * hy2; i += 256) {
            const int ty = lo + i / wx, tx = lo + i % wx, x = x0 + tx, y = y0 + ty;
            const int a = Bf[ty - 1][tx], b2 = Bf[ty][tx], c = Bf[ty + 1][tx];
            const int v = dilate ? max(a, max(b2, c)) : min(a, min(b2, c));
            A[ty][tx] = (uint8_t)((x >= 0 && x < W && y >= 0 && y < H) ? v : next_neutral);
        }
        __syncthreads();
    }
    for (int i = threadIdx.x; i < MT_W * MT_H; i += 256) {
        const int ty = MT_M + i / MT_W, tx = MT_M + i % MT_W, x = x0 + tx, y = y0 + ty;
        if (x < W && y < H) dst[(size_t)y * W + x] = A[ty][tx];
    }
}

// Labelling, step 1: a pixel's label starts as the index of the first pixel of its row run -- the pixels next to each
// other in a row that the labelling joins anyway (background pixels, blocked[i] == 0, and with fg8 the blocked ones;
// without fg8 a blocked pixel is a run of its own).  One wave per image row finds the run starts of 64 pixels with one
// ballot.  The unions along rows, which were nearly all of them (the background of a laser frame is one component of
// a few hundred thousand pixels, every union contending for the same root: 0.5 ms per 640x360 frame), are then not needed.
// Node n = W * H, "outside the image", is initialised by the first row's wave.
__global__ void __launch_bounds__(64) k_lab_runs(const uint8_t *__restrict__ blocked, int W, int H, int fg8, int *__restrict__ lab)
{
    const int y = blockIdx.x, lane = threadIdx.x;
    blocked += (size_t)blockIdx.y * W * H;           // image of a stack; its labels (one more than pixels)
    lab += (size_t)blockIdx.y * ((size_t)W * H + 1);
    int carry_cls = -1, carry_start = 0;
    for (int x0 = 0; x0 < W; x0 += 64) {
        const int x = x0 + lane;
        const bool valid = x < W;
        const int cls = valid ? (blocked[(size_t)y * W + x] != 0 ? 1 : 0) : 2;
        int prev = __shfl_up(cls, 1);
        if (lane == 0) prev = carry_cls;
        const bool joins_left = cls == prev && (cls == 0 || fg8);
        const unsigned long long starts = __ballot(valid && !joins_left);
        const unsigned long long below = starts & ((lane == 63) ? ~0ull : ((2ull << lane) - 1));
        const int sx = below ? x0 + 63 - __clzll(below) : carry_start;
        if (valid) lab[(size_t)y * W + x] = y * W + sx;
        carry_cls = __shfl(cls, 63);
        carry_start = __shfl(sx, 63);
    }
    if (y == 0 && lane == 0) lab[(size_t)W * H] = W * H;
}

// Labelling, step 2.  Nodes: every pixel plus node n = "outside the image".  Pixels with blocked[i] != 0 are joined
// with 8-connectivity among themselves when fg8 is set (foreground components) or left isolated
// when it is not; pixels with blocked[i] == 0 are joined with 4-connectivity among themselves and
// with the outside node along the image border.  The row runs are joined already (k_lab_runs): what is left are the
// links between adjacent rows.  For the background one union per pair of overlapping runs is enough: at the first
// column of the overlap one of the two runs starts.
__global__ void __launch_bounds__(256) k_lab_merge(const uint8_t *__restrict__ blocked, int W, int H, int fg8, int *lab)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= W) return;
    const int i = y * W + x, n = W * H;
    blocked += (size_t)blockIdx.z * n;
    lab += (size_t)blockIdx.z * ((size_t)n + 1);
    const bool f = blocked[i] != 0;
    if (f) {
        if (!fg8) return;
        if (y + 1 < H) {
            if (blocked[i + W]) s_uf_union(lab, i, i + W);
            if (x + 1 < W && blocked[i + W + 1]) s_uf_union(lab, i, i + W + 1);
            if (x > 0 && blocked[i + W - 1]) s_uf_union(lab, i, i + W - 1);
        }
    } else {
        // (in the first column both pixels of a vertical pair are joined with the outside node anyway; leaving that pair out
        // keeps the big outside component a star around node n instead of a chain down the image's left edge)
        const bool starts = x == 0 || blocked[i - 1];
        if (x > 0 && y + 1 < H && !blocked[i + W] && (starts || blocked[i + W - 1])) s_uf_union(lab, i, i + W);
        if (x == 0 || x == W - 1 || ((y == 0 || y == H - 1) && starts)) s_uf_union(lab, i, n);
    }
}

// Labelling, step 3: every node's label becomes the root of its component.  First the run starts and the outside node
// walk to their roots (they hold the long chains, one link per image row at worst; a label only ever moves towards the
// root, so a walk that meets a label another thread has just shortened still ends at the root) ...
__global__ void __launch_bounds__(256) k_lab_flatten_heads(const uint8_t *__restrict__ blocked, int W, int H, int fg8, int *lab)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= W) return;
    const int i = y * W + x;
    blocked += (size_t)blockIdx.z * W * H;
    lab += (size_t)blockIdx.z * ((size_t)W * H + 1);
    if (x > 0) {
        const bool f = blocked[i] != 0, fl = blocked[i - 1] != 0;
        if (f == fl && (!f || fg8)) return;      // inside a run
    }
    int r = i, p;
    while ((p = s_uf_load(lab, r)) != r) r = p;
    __hip_atomic_store(lab + i, r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (i == 0) {                                // the outside node
        r = W * H;
        while ((p = s_uf_load(lab, r)) != r) r = p;
        __hip_atomic_store(lab + W * H, r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}
// ... then every other pixel takes the root from its run start.
__global__ void __launch_bounds__(256) k_lab_flatten(int n1, int *lab)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n1) return;
    lab += (size_t)blockIdx.y * n1;
    const int h = s_uf_load(lab, i);
    const int r = s_uf_load(lab, h);
    if (r != h) __hip_atomic_store(lab + i, r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// one thread per 2x2 cell with top-left corner (cx, cy), cx in [-1, W-1], cy in [-1, H-1]
__global__ void __launch_bounds__(256) k_contour_area(const uint8_t *__restrict__ mask, const int *__restrict__ lab, int W, int H,
                                                      long long *area2, int *touch)
{
    const int cx = (int)(blockIdx.x * blockDim.x + threadIdx.x) - 1, cy = (int)blockIdx.y - 1;
    if (cx > W - 1) return;
    const int n = W * H;
    mask += (size_t)blockIdx.z * n;
    lab += (size_t)blockIdx.z * ((size_t)n + 1);
    area2 += (size_t)blockIdx.z * ((size_t)n + 1);
    touch += (size_t)blockIdx.z * ((size_t)n + 1);
    const int outer = lab[n];
    const int px[4] = {cx, cx + 1, cx, cx + 1}, py[4] = {cy, cy, cy + 1, cy + 1}; // a b c d
    bool in[4];
    int lb[4];
    int cnt = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const bool inside = px[k] >= 0 && px[k] < W && py[k] >= 0 && py[k] < H;
        in[k] = inside && mask[inside ? py[k] * W + px[k] : 0] != 0;
        cnt += in[k];
    }
    if (cnt == 0 || cnt == 4) return;      // no border passes through the cell (nearly all cells: the labels are not read)
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const bool inside = px[k] >= 0 && px[k] < W && py[k] >= 0 && py[k] < H;
        lb[k] = inside ? lab[py[k] * W + px[k]] : outer;
    }
    // foreground pixels 4-adjacent to the outside background mark their component (pairs a-b, a-c, b-d, c-d)
    const int pa[4] = {0, 0, 1, 2}, pb[4] = {1, 2, 3, 3};
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int u = pa[k], v = pb[k];
        if (in[u] && !in[v] && lb[v] == outer) touch[lb[u]] = 1;
        if (in[v] && !in[u] && lb[u] == outer) touch[lb[v]] = 1;
    }
    auto add_edge = [&](int p, int q, int gk) {
        if (lb[gk] != outer) return;
        int x0 = px[p], y0 = py[p], x1 = px[q], y1 = py[q];
        const int cr = (x1 - x0) * (py[gk] - y0) - (y1 - y0) * (px[gk] - x0);
        if (cr < 0) { const int tx = x0, ty = y0; x0 = x1; y0 = y1; x1 = tx; y1 = ty; }
        const long long term = (long long)(x0 + 1) * (y1 + 1) - (long long)(y0 + 1) * (x1 + 1); // +1: padded coordinates
        atomicAdd(reinterpret_cast<unsigned long long *>(area2 + lb[p]), (unsigned long long)term);
    };
    if (cnt == 3) {
        if (!in[0]) add_edge(1, 2, 0);
        else if (!in[1]) add_edge(0, 3, 1);
        else if (!in[2]) add_edge(0, 3, 2);
        else add_edge(1, 2, 3);
    } else if (cnt == 2) {
        if (in[0] && in[1]) add_edge(0, 1, 2);
        else if (in[2] && in[3]) add_edge(2, 3, 0);
        else if (in[0] && in[2]) add_edge(0, 2, 1);
        else if (in[1] && in[3]) add_edge(1, 3, 0);
        else if (in[0] && in[3]) { add_edge(0, 3, 1); add_edge(0, 3, 2); }
        else { add_edge(1, 2, 0); add_edge(1, 2, 3); }
    }
}

__global__ void __launch_bounds__(256) k_keep(const uint8_t *__restrict__ mask, const int *__restrict__ lab, int n, const long long *__restrict__ area2,
                                              const int *__restrict__ touch, double min_area, uint8_t *__restrict__ kept)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    mask += (size_t)blockIdx.y * n;
    kept += (size_t)blockIdx.y * n;
    lab += (size_t)blockIdx.y * ((size_t)n + 1);
    area2 += (size_t)blockIdx.y * ((size_t)n + 1);
    touch += (size_t)blockIdx.y * ((size_t)n + 1);
    uint8_t k = 0;
    if (mask[i]) {
        const int r = lab[i];
        const long long a = area2[r];
        const double area = fabs((double)a * 0.5);
        k = (touch[r] && area > min_area) ? 255 : 0;
    }
    kept[i] = k;
}

__global__ void __launch_bounds__(256) k_final_mask(const uint8_t *__restrict__ kept, const int *__restrict__ lab2, int n, uint8_t *__restrict__ fin)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    kept += (size_t)blockIdx.y * n;
    fin += (size_t)blockIdx.y * n;
    lab2 += (size_t)blockIdx.y * ((size_t)n + 1);
    fin[i] = (kept[i] || lab2[i] != lab2[n]) ? 255 : 0;
}

// one block per row: number of mask pixels, sum of x * gray, sum of gray
__global__ void __launch_bounds__(256) k_row_centroid(const uint8_t *__restrict__ fin, const uint8_t *__restrict__ gray, int W,
                                                      unsigned long long *__restrict__ rows /* [H][3] */)
{
    __shared__ unsigned long long sh[3][256];
    const int y = blockIdx.x, tid = threadIdx.x;
    fin += (size_t)blockIdx.y * W * gridDim.x;       // image of a stack (gridDim.x = image height)
    gray += (size_t)blockIdx.y * W * gridDim.x;
    rows += (size_t)blockIdx.y * gridDim.x * 3;
    unsigned long long c = 0, sxw = 0, sw = 0;
    for (int x = tid; x < W; x += 256) {
        if (fin[(size_t)y * W + x]) {
            const unsigned g = gray[(size_t)y * W + x];
            c++; sxw += (unsigned long long)x * g; sw += g;
        }
    }
    sh[0][tid] = c; sh[1][tid] = sxw; sh[2][tid] = sw;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (tid < o) { sh[0][tid] += sh[0][tid + o]; sh[1][tid] += sh[1][tid + o]; sh[2][tid] += sh[2][tid + o]; }
        __syncthreads();
    }
    if (tid == 0) { rows[3 * y] = sh[0][0]; rows[3 * y + 1] = sh[1][0]; rows[3 * y + 2] = sh[2][0]; }
}

} // namespace l3d

using namespace l3d;

struct l3d_simple {
    int device = 0, max_w = 0, max_h = 0, max_batch = 1;
    hipStream_t stream = nullptr;
    uint8_t *bgr = nullptr, *gray = nullptr, *m0 = nullptr, *m1 = nullptr, *kept = nullptr, *fin = nullptr;
    int *lab = nullptr, *lab2 = nullptr, *touch = nullptr, *sdiv = nullptr, *hdiv = nullptr;
    long long *area2 = nullptr;
    unsigned long long *rows = nullptr;
};

extern "C" void l3d_simple_destroy(l3d_simple *h)
{
    if (!h) return;
    DeviceGuard dg(h->device);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    void *bufs[] = {h->bgr, h->gray, h->m0, h->m1, h->kept, h->fin, h->lab, h->lab2, h->touch, h->sdiv, h->hdiv, h->area2, h->rows};
    for (void *p : bufs)
        if (p) (void)hipFree(p);
    if (h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
}

extern "C" int l3d_simple_create_batch(int max_width, int max_height, int max_batch, int device, l3d_simple **out)
{
    L3D_REQUIRE(out, "l3d_simple_create_batch: null argument");
    *out = nullptr;
    L3D_REQUIRE(max_width > 0 && max_height > 0 && max_batch > 0 && (long long)max_width * max_height < (1ll << 30) && max_batch <= 65535,
                "l3d_simple_create_batch: bad size");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        set_error("l3d_simple_create_batch: no HIP device available");
        return L3D_ERR_NO_DEVICE;
    }
    L3D_REQUIRE(device >= 0 && device < ndev, "l3d_simple_create_batch: device %d out of range", device);
    l3d_simple *h = new (std::nothrow) l3d_simple();
    L3D_REQUIRE(h, "l3d_simple_create_batch: out of host memory");
    h->device = device; h->max_w = max_width; h->max_h = max_height; h->max_batch = max_batch;
    DeviceGuard dg(device);
    const size_t n = (size_t)max_width * max_height, B = (size_t)max_batch;
    hipError_t e = hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking);
    auto alloc = [&](void **ptr, size_t bytes) {
        if (e == hipSuccess) e = hipMalloc(ptr, bytes);
    };
    alloc((void **)&h->bgr, B * n * 3);
    for (uint8_t **p : {&h->gray, &h->m0, &h->m1, &h->kept, &h->fin}) alloc((void **)p, B * n);
    for (int **p : {&h->lab, &h->lab2, &h->touch}) alloc((void **)p, B * (n + 1) * sizeof(int));
    alloc((void **)&h->area2, B * (n + 1) * sizeof(long long));
    alloc((void **)&h->rows, B * (size_t)max_height * 3 * sizeof(unsigned long long));
    alloc((void **)&h->sdiv, 256 * sizeof(int));
    alloc((void **)&h->hdiv, 256 * sizeof(int));
    if (e == hipSuccess) {
        // OpenCV's 8-bit HSV tables: saturate_cast<int>((255 << 12) / (1. * i)), ((180 << 12) / (6. * i))
        int sd[256], hd[256];
        sd[0] = hd[0] = 0;
        for (int i = 1; i < 256; i++) {
            sd[i] = (int)std::nearbyint((255 << 12) / (1. * i));
            hd[i] = (int)std::nearbyint((180 << 12) / (6. * i));
        }
        e = hipMemcpy(h->sdiv, sd, sizeof(sd), hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMemcpy(h->hdiv, hd, sizeof(hd), hipMemcpyHostToDevice);
    }
    if (e != hipSuccess) {
        set_error("l3d_simple_create_batch: %s", hipGetErrorString(e));
        l3d_simple_destroy(h);
        return L3D_ERR_HIP;
    }
    *out = h;
    return L3D_OK;
}

extern "C" int l3d_simple_create(int max_width, int max_height, int device, l3d_simple **out)
{
    return l3d_simple_create_batch(max_width, max_height, 1, device, out);
}

int simple_clean_mask(l3d_simple *h, const uint8_t *d_bgr, int W, int H, const int *lo, const int *hi, int thr, hipStream_t s,
                      const uint8_t **d_mask)
{
    L3D_REQUIRE(h && W > 0 && H > 0 && W <= h->max_w && H <= h->max_h && (size_t)W * H <= (size_t)h->max_w * h->max_h,
                "clean mask: image %dx%d exceeds the handle's %dx%d", W, H, h ? h->max_w : 0, h ? h->max_h : 0);
    const int n = W * H;
    HsvRange r;
    for (int k = 0; k < 3; k++) { r.lo[k] = lo[k]; r.hi[k] = hi[k]; }
    r.thr = thr;
    const dim3 g1(cdiv(n, 256)), g2(cdiv(W, 256), H), b(256);
    hipLaunchKernelGGL(k_simple_mask, g1, b, 0, s, d_bgr, (size_t)n, r, h->sdiv, h->hdiv, h->gray, h->m0);
    hipLaunchKernelGGL(k_close_open, dim3(cdiv(W, MT_W), cdiv(H, MT_H), 1), b, 0, s, h->m0, W, H, h->m1);
    L3D_HIP_CHECK(hipGetLastError());
    *d_mask = h->m1;
    return L3D_OK;
}

// `B` images [B][H][W][3] in one sequence of launches (every kernel takes the image from its grid's last dimension);
// xy [B][H][2], count [B]; masks only for a single image.
static int simple_run(l3d_simple *h, const uint8_t *d_bgr, int W, int H, int B, const int *lo, const int *hi, int thr, double min_area, double *xy,
                      int *count, uint8_t *masks, hipStream_t s)
{
    const int n = W * H, n1 = n + 1;
    HsvRange r;
    for (int k = 0; k < 3; k++) { r.lo[k] = lo[k]; r.hi[k] = hi[k]; }
    r.thr = thr;
    const size_t nB = (size_t)n * B;
    const dim3 gall((unsigned)((nB + 255) / 256)), g1(cdiv(n, 256), B), g11(cdiv(n1, 256), B), g2(cdiv(W, 256), H, B), b(256);
    hipLaunchKernelGGL(k_simple_mask, gall, b, 0, s, d_bgr, nB, r, h->sdiv, h->hdiv, h->gray, h->m0);
    if (masks) L3D_HIP_CHECK(hipMemcpyAsync(masks, h->m0, n, hipMemcpyDeviceToHost, s));
    hipLaunchKernelGGL(k_close_open, dim3(cdiv(W, MT_W), cdiv(H, MT_H), B), b, 0, s, h->m0, W, H, h->m1);
    if (masks) L3D_HIP_CHECK(hipMemcpyAsync(masks + n, h->m1, n, hipMemcpyDeviceToHost, s));
    // components of the mask and of its background
    hipLaunchKernelGGL(k_lab_runs, dim3(H, B), dim3(64), 0, s, h->m1, W, H, 1, h->lab);
    hipLaunchKernelGGL(k_lab_merge, g2, b, 0, s, h->m1, W, H, 1, h->lab);
    hipLaunchKernelGGL(k_lab_flatten_heads, g2, b, 0, s, h->m1, W, H, 1, h->lab);
    hipLaunchKernelGGL(k_lab_flatten, g11, b, 0, s, n1, h->lab);
    L3D_HIP_CHECK(hipMemsetAsync(h->area2, 0, (size_t)n1 * B * sizeof(long long), s));
    L3D_HIP_CHECK(hipMemsetAsync(h->touch, 0, (size_t)n1 * B * sizeof(int), s));
    hipLaunchKernelGGL(k_contour_area, dim3(cdiv(W + 1, 256), H + 1, B), b, 0, s, h->m1, h->lab, W, H, h->area2, h->touch);
    hipLaunchKernelGGL(k_keep, g1, b, 0, s, h->m1, h->lab, n, h->area2, h->touch, min_area, h->kept);
    // filled contours: whatever a 4-connected flood from outside cannot reach around the kept components
    hipLaunchKernelGGL(k_lab_runs, dim3(H, B), dim3(64), 0, s, h->kept, W, H, 0, h->lab2);
    hipLaunchKernelGGL(k_lab_merge, g2, b, 0, s, h->kept, W, H, 0, h->lab2);
    hipLaunchKernelGGL(k_lab_flatten_heads, g2, b, 0, s, h->kept, W, H, 0, h->lab2);
    hipLaunchKernelGGL(k_lab_flatten, g11, b, 0, s, n1, h->lab2);
    hipLaunchKernelGGL(k_final_mask, g1, b, 0, s, h->kept, h->lab2, n, h->fin);
    if (masks) L3D_HIP_CHECK(hipMemcpyAsync(masks + 2 * (size_t)n, h->fin, n, hipMemcpyDeviceToHost, s));
    hipLaunchKernelGGL(k_row_centroid, dim3(H, B), b, 0, s, h->fin, h->gray, W, h->rows);
    L3D_HIP_CHECK(hipGetLastError());
    std::vector<unsigned long long> rows((size_t)B * H * 3);
    L3D_HIP_CHECK(hipMemcpyAsync(rows.data(), h->rows, rows.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
    L3D_HIP_CHECK(hipStreamSynchronize(s));
    for (int f = 0; f < B; f++) {
        const unsigned long long *rw = rows.data() + (size_t)f * H * 3;
        double *o = xy + (size_t)f * H * 2;
        int m = 0;
        for (int y = 0; y < H; y++) {
            if (rw[3 * y] > 0 && rw[3 * y + 2] > 0) {
                o[2 * m] = (double)rw[3 * y + 1] / (double)rw[3 * y + 2];
                o[2 * m + 1] = (double)y;
                m++;
            }
        }
        count[f] = m;
    }
    return L3D_OK;
}

static int simple_check(l3d_simple *h, int W, int H, const void *a, const void *b2, const void *c, const void *d, const void *e, const char *who)
{
    L3D_REQUIRE(h && a && b2 && c && d && e, "%s: null argument", who);
    L3D_REQUIRE(W > 0 && H > 0 && (size_t)W * H <= (size_t)h->max_w * h->max_h && H <= h->max_h, "%s: image %dx%d exceeds the handle's %dx%d", who, W, H,
                h->max_w, h->max_h);
    return L3D_OK;
}

extern "C" int l3d_simple_extract(l3d_simple *h, const uint8_t *bgr, int width, int height, const int *hsv_lower, const int *hsv_upper,
                                  int brightness_threshold, double min_area, double *xy, int *count, uint8_t *masks)
{
    int rc = simple_check(h, width, height, bgr, hsv_lower, hsv_upper, xy, count, "l3d_simple_extract");
    if (rc) return rc;
    DeviceGuard dg(h->device);
    L3D_HIP_CHECK(hipMemcpyAsync(h->bgr, bgr, (size_t)width * height * 3, hipMemcpyHostToDevice, h->stream));
    return simple_run(h, h->bgr, width, height, 1, hsv_lower, hsv_upper, brightness_threshold, min_area, xy, count, masks, h->stream);
}

extern "C" int l3d_simple_extract_device(l3d_simple *h, const uint8_t *d_bgr, int width, int height, const int *hsv_lower, const int *hsv_upper,
                                         int brightness_threshold, double min_area, double *xy, int *count, void *stream)
{
    int rc = simple_check(h, width, height, d_bgr, hsv_lower, hsv_upper, xy, count, "l3d_simple_extract_device");
    if (rc) return rc;
    DeviceGuard dg(h->device);
    return simple_run(h, d_bgr, width, height, 1, hsv_lower, hsv_upper, brightness_threshold, min_area, xy, count, nullptr, (hipStream_t)stream);
}

extern "C" int l3d_simple_extract_batch(l3d_simple *h, const uint8_t *bgr, int width, int height, int batch, const int *hsv_lower,
                                        const int *hsv_upper, int brightness_threshold, double min_area, double *xy, int *counts)
{
    int rc = simple_check(h, width, height, bgr, hsv_lower, hsv_upper, xy, counts, "l3d_simple_extract_batch");
    if (rc) return rc;
    L3D_REQUIRE(batch >= 1 && batch <= h->max_batch, "l3d_simple_extract_batch: batch %d exceeds the handle's %d", batch, h->max_batch);
    DeviceGuard dg(h->device);
    L3D_HIP_CHECK(hipMemcpyAsync(h->bgr, bgr, (size_t)width * height * 3 * batch, hipMemcpyHostToDevice, h->stream));
    return simple_run(h, h->bgr, width, height, batch, hsv_lower, hsv_upper, brightness_threshold, min_area, xy, counts, nullptr, h->stream);
}

extern "C" int l3d_simple_extract_batch_device(l3d_simple *h, const uint8_t *d_bgr, int width, int height, int batch, const int *hsv_lower,
                                               const int *hsv_upper, int brightness_threshold, double min_area, double *xy, int *counts,
                                               void *stream)
{
    int rc = simple_check(h, width, height, d_bgr, hsv_lower, hsv_upper, xy, counts, "l3d_simple_extract_batch_device");
    if (rc) return rc;
    L3D_REQUIRE(batch >= 1 && batch <= h->max_batch, "l3d_simple_extract_batch_device: batch %d exceeds the handle's %d", batch, h->max_batch);
    DeviceGuard dg(h->device);
    return simple_run(h, d_bgr, width, height, batch, hsv_lower, hsv_upper, brightness_threshold, min_area, xy, counts, nullptr,
                      (hipStream_t)stream);
}
