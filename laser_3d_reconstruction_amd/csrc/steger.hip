// Steger ridge-centre extraction for gfx950.
//
// Replaces FastStegerExtractor.extract_centerline / extract_batch of the reference
// (core/laser_extractor.py:160-261, 279-285): BGR->gray, float GaussianBlur(sigma), five
// filter2D difference stencils, and the per-pixel Python loop that calls np.linalg.eig on the
// 2x2 Hessian of every pixel brighter than the threshold.
//
// A streaming pass flags the 64x16 pixel tiles that hold a pixel above the threshold (the laser covers a few
// percent of a frame); one fused kernel, a fixed grid whose blocks share the tiles out by stride, then works per flagged tile:
//   gray (u8) of the tile -> exact per-pixel brightness test (exit when none passes) -> gray tile with
//   halo staged in LDS as float -> separable Gaussian through LDS -> stencils, closed-form 2x2
//   symmetric eigen-decomposition in double, acceptance test -> per-row ballot compaction into
//   fixed per-(row, tile) cells.
// A second kernel turns the cell counts into raster-order offsets (one block per image) and
// copies the points into the dense output, so the result has the order of the reference's
// `for j: for i:` loop.
//
// The same tile kernel, as template variants, serves ImprovedStegerExtractor.extract_centerline,
// .extract_centerline_optimized and HybridLaserExtractor.extract_centerline (improved_steger.py:39-126,
// 128-223, 250-344): Sobel-of-Sobel derivatives (blurred halo of 2), a negative-eigenvalue test, and for
// the last two one point per image row, the accepted pixel with the largest |eigenvalue| (wave arg-max in
// the tile, then a per-row pass over the row's tiles).
#include "common.h"
#include <cmath>
#include <cstdlib>
#include <new>
#include <vector>

namespace l3d {

constexpr int TW = 64, TH = 16;
constexpr int MAX_TAPS = 65;
typedef float f32x2 __attribute__((ext_vector_type(2)));

struct StegerArgs {
    int W, H;             // full image size
    int channels;         // 1 or 3
    int x0, y0, rw, rh;   // region processed (ROI or whole image)
    int R;                // kernel radius (ksize = 2R+1)
    int thr;
    int ntx, nty;         // tiles per region
    int ntiles;           // ntx * nty * batch
    double resp_thr;      // largest accepted |offset| of the improved variants
};

// kernel variants
constexpr int VAR_FAST = 0;      // FastStegerExtractor: every accepted pixel
constexpr int VAR_IMPROVED = 1;  // ImprovedStegerExtractor.extract_centerline: every accepted pixel
constexpr int VAR_PER_ROW = 2;   // ... .extract_centerline_optimized: best pixel per row
constexpr int VAR_HYBRID = 3;    // HybridLaserExtractor: best pixel per row among the pixels of a mask

__device__ __forceinline__ int reflect101(int p, int n)
{
    if (n == 1) return 0;
    while (p < 0 || p >= n) p = p < 0 ? -p : 2 * (n - 1) - p;
    return p;
}

__device__ __forceinline__ int gray_at(const uint8_t *img, const StegerArgs &a, int x, int y)
{
    // (x, y) relative to the region
    const uint8_t *p = img + ((size_t)(y + a.y0) * a.W + (x + a.x0)) * a.channels;
    if (a.channels == 3) return (int)(((uint32_t)p[2] * 9798u + (uint32_t)p[1] * 19235u + (uint32_t)p[0] * 3735u + (1u << 14)) >> 15);
    return p[0];
}

// Steger decision from the five derivative values (all float, as the reference's arrays);
// the 2x2 symmetric eigenproblem and the offset are evaluated in double.
__device__ __forceinline__ bool steger_point(float dx, float dy, float dxx, float dyy, float dxy, double &ox, double &oy)
{
    const double a = dxx, b = dxy, c = dyy;
    double nx, ny;
    if (b == 0.0) {
        if (fabs(a) >= fabs(c)) { nx = 1.0; ny = 0.0; }
        else { nx = 0.0; ny = 1.0; }
    } else {
        const double m = 0.5 * (a + c);
        const double hd = 0.5 * (a - c);
        const double rr = sqrt(hd * hd + b * b);
        const double lam = (m >= 0.0) ? m + rr : m - rr;
        const double v1x = b, v1y = lam - a;
        const double v2x = lam - c, v2y = b;
        const double n1 = v1x * v1x + v1y * v1y, n2 = v2x * v2x + v2y * v2y;
        // the eigenvector with the larger norm, chosen before the square root and the division so that a wave
        // whose lanes disagree runs them once
        const bool first = n1 >= n2;
        const double s = 1.0 / sqrt(first ? n1 : n2);
        nx = (first ? v1x : v2x) * s;
        ny = (first ? v1y : v2y) * s;
    }
    const double den = nx * nx * a + 2.0 * nx * ny * b + ny * ny * c;
    if (!(fabs(den) > 1e-10)) return false;
    const double t = -(nx * (double)dx + ny * (double)dy) / den;
    const double tx = t * nx, ty = t * ny;
    if (fabs(tx) <= 0.5 && fabs(ty) <= 0.5) { ox = tx; oy = ty; return true; }
    return false;
}

// Decision of the improved variants.  The Hessian entries are float as in the reference; its eigenpair comes
// from the closed form in double and is rounded to float (np.linalg.eig of a float32 matrix returns float32);
// the remaining scalar arithmetic follows NumPy < 2.0: float32 * float32 stays float32, a Python number times
// a float32 scalar is float64.  Returns false when the reference `continue`s; lam = the eigenvalue.
__device__ __forceinline__ bool steger2_point(float dx, float dy, float dxx, float dyy, float dxy, float &lamf, double &ox, double &oy)
{
    const double a = dxx, b = dxy, c = dyy;
    double lam, nx, ny;
    if (b == 0.0) {
        if (fabs(a) >= fabs(c)) { lam = a; nx = 1.0; ny = 0.0; }
        else { lam = c; nx = 0.0; ny = 1.0; }
    } else {
        const double m = 0.5 * (a + c), hd = 0.5 * (a - c);
        const double rr = sqrt(hd * hd + b * b);
        lam = (m >= 0.0) ? m + rr : m - rr;
        const double v1x = b, v1y = lam - a, v2x = lam - c, v2y = b;
        const double n1 = v1x * v1x + v1y * v1y, n2 = v2x * v2x + v2y * v2y;
        const bool first = n1 >= n2;
        const double s = 1.0 / sqrt(first ? n1 : n2);
        nx = (first ? v1x : v2x) * s;
        ny = (first ? v1y : v2y) * s;
    }
    lamf = (float)lam;
    const float nxf = (float)nx, nyf = (float)ny;
    if (lamf >= 0.0f) return false;
    const float t1 = (nxf * nxf) * dxx, t3 = (nyf * nyf) * dyy;
    const double t2 = ((2.0 * (double)nxf) * (double)nyf) * (double)dxy;
    const double den = ((double)t1 + t2) + (double)t3;
    if (fabs(den) < 1e-6) return false;
    const float num = -(nxf * dx + nyf * dy);
    const double t = (double)num / den;
    ox = t * (double)nxf;
    oy = t * (double)nyf;
    return true;
}

// grid: blocks that walk the list of marked tiles (or, without a list, (ntx, nty, batch): one block per tile);
// block 256 = 4 waves; wave w handles tile rows w, w+4, w+8, w+12
template <int VAR>
__global__ void __launch_bounds__(256) k_steger_tile(StegerArgs a, const uint8_t *__restrict__ img, const uint8_t *__restrict__ mask,
                                                     const float *__restrict__ taps, float2 *__restrict__ cells, int *__restrict__ cell_counts,
                                                     int *__restrict__ tiles)
{
    extern __shared__ float sm[];
    __shared__ int s_list[256];
    __shared__ int s_n, s_wcnt[4];
    constexpr int HB = VAR == VAR_FAST ? 1 : 2;         // halo of the blurred tile that the derivative stencils reach
    const int R = a.R, ks = 2 * R + 1;
    const int GW = TW + 2 * HB + 2 * R, GH = TH + 2 * HB + 2 * R; // gray tile incl. halo
    const int BW = TW + 2 * HB, BH = TH + 2 * HB;                 // blurred tile
    float *hrow = sm;               // [GH][BW] row-filtered
    float *bl = hrow + GH * BW;     // [BH][BW] blurred
    float *tp = bl + BH * BW;       // [ks]
    volatile int &any_bright = *reinterpret_cast<volatile int *>(tp + ks);
    int *xrange = reinterpret_cast<int *>(tp + ks) + 1;   // [2]: first and last tile column with a bright pixel
    uint8_t *gt = reinterpret_cast<uint8_t *>(xrange + 2);   // [GH][GW] gray values, kept as bytes (a quarter of the LDS: 8 blocks per CU instead of 5)

    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    // With tile flags (tiles[1 + t] != 0: tile t holds a bright pixel) the grid is G blocks per image: every block of an
    // image reads that image's flags, 256 at a time, ranks the flagged tiles by ballots and prefix sums, and works on those
    // whose rank is its own number modulo G -- the flagged tiles of an image are shared out evenly whatever their pattern,
    // without a list in memory, a kernel to build it, or additions to a shared counter.  The compaction kernel, which runs
    // after this one, puts the flags back to zero.  Without flags the grid is (ntx, nty, batch) and the block works on
    // its own tile.
    const int nt = a.ntx * a.nty;                                   // tiles of an image
    const int n_img = a.ntiles / nt;
    const int G = tiles ? (int)gridDim.x / n_img : 1;
    // Block numbers run over the images first, then over g: blocks start in the order of their numbers, and the blocks with
    // a low g never have the smaller share of their image's tiles, so the longer blocks start first and the launch does not
    // end in a tail of them (140 -> 122 us per 256 frames against g running first).
    const int img_b = tiles ? (int)blockIdx.x % n_img : 0, g = tiles ? (int)blockIdx.x / n_img : 0;
    int rank_base = 0;
    for (int c0 = 0; c0 < (tiles ? nt : 1); c0 += 256) {
    if (tiles) {
        __syncthreads();       // the previous chunk's list is used up
        if (tid == 0) s_n = 0;
        const int i = c0 + tid;
        const bool on = i < nt && tiles[1 + (size_t)img_b * nt + i] != 0;
        const unsigned long long m = __ballot(on);
        if (lane == 0) s_wcnt[wv] = __popcll(m);
        __syncthreads();
        int pre = rank_base;
        for (int w = 0; w < wv; w++) pre += s_wcnt[w];
        const int r = pre + __popcll(m & ((1ull << lane) - 1ull));
        if (on && r % G == g) s_list[atomicAdd(&s_n, 1)] = img_b * nt + i;
        rank_base += s_wcnt[0] + s_wcnt[1] + s_wcnt[2] + s_wcnt[3];
        __syncthreads();
    }
    const int n_here = tiles ? s_n : 1;
    for (int ti = 0; ti < n_here; ti++) {
    int txi = blockIdx.x, tyi = blockIdx.y, b = blockIdx.z;
    if (tiles) {
        const int t = s_list[ti];
        txi = t % a.ntx;
        tyi = (t / a.ntx) % a.nty;
        b = t / (a.ntx * a.nty);
        __syncthreads();   // the previous tile's LDS is free
    }
    const int tx0 = txi * TW, ty0 = tyi * TH;
    const uint8_t *im = img + (size_t)b * a.W * a.H * a.channels;
    const uint8_t *mk = VAR == VAR_HYBRID ? mask + (size_t)b * a.W * a.H : nullptr;
    const size_t cell_base = ((size_t)b * a.rh) * a.ntx;
    // value that is blurred: the gray image; for the hybrid variant the gray image zeroed outside the mask
    auto src_at = [&](int x, int y) -> int {
        if (VAR == VAR_HYBRID && !mk[(size_t)y * a.W + x]) return 0;
        return gray_at(im, a, x, y);
    };

    if (tid == 0) { any_bright = 0; xrange[0] = TW; xrange[1] = -1; }
    __syncthreads();
    // own pixels: brightness test on the unblurred gray value
    int own[4];
    bool bright = false;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int x = tx0 + lane, y = ty0 + wv + 4 * k;
        const bool inside = VAR == VAR_FAST ? (x < a.rw && y < a.rh) : (x >= 1 && x < a.rw - 1 && y >= 1 && y < a.rh - 1);
        if (VAR == VAR_HYBRID) own[k] = (inside && mk[(size_t)y * a.W + x]) ? a.thr + 1 : -1;
        else own[k] = inside ? gray_at(im, a, x, y) : -1;
        bright |= own[k] > a.thr;
    }
    {
        const unsigned long long bm = __ballot(bright);
        if (bm != 0ull && lane == 0) {
            any_bright = 1;
            atomicMin(&xrange[0], __ffsll((long long)bm) - 1);
            atomicMax(&xrange[1], 63 - __clzll((long long)bm));
        }
    }
    __syncthreads();
    if (!any_bright) {
        if (lane == 0) {
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int y = ty0 + wv + 4 * k;
                if (y < a.rh) cell_counts[cell_base + (size_t)y * a.ntx + txi] = 0;
            }
        }
        continue;
    }

    for (int i = tid; i < ks; i += 256) tp[i] = taps[i];
    // Only the columns around the bright pixels are blurred (a laser line crosses a tile in a few columns): blurred
    // columns b0 .. b0+nb-1 of the tile's BW, from gray columns b0 .. b0+nb+2R-1 of its GW.  Every value is
    // computed as it would be for the whole tile.
    const int b0 = xrange[0], nb = xrange[1] - xrange[0] + 1 + 2 * HB, ng = nb + 2 * R;
    // gray tile with halo; coordinates outside the region are mirrored (BORDER_REFLECT_101).  A lane takes a column, a
    // wave every fourth row: the mirrored column is found once per lane, the mirrored row is wave-uniform.
    for (int gxl = lane; gxl < ng; gxl += 64) {
        const int gx = b0 + gxl;
        const int x = reflect101(tx0 - HB - R + gx, a.rw);
#pragma unroll 4
        for (int gy = wv; gy < GH; gy += 4) {
            const int y = reflect101(ty0 - HB - R + gy, a.rh);
            gt[gy * GW + gx] = (uint8_t)src_at(x, y);
        }
    }
    __syncthreads();
    // row pass: taps accumulated left to right.  Outputs are dealt to the threads in raster order of the nb columns
    // (all lanes busy whatever nb is); a thread steps from one of its outputs to the next without dividing again.
    // A thread computes the same column of two adjacent rows (GH is even) as the two halves of packed single-precision
    // operations: one multiplication and one addition per tap for both, each half rounded as the scalar operation is.
    {
        const int q256 = 256 / nb, r256 = 256 - q256 * nb;
        int gp = tid / nb, bxl = tid - gp * nb;
        for (; gp < GH / 2; ) {
            const uint8_t *src0 = gt + 2 * gp * GW + b0 + bxl, *src1 = src0 + GW;
            f32x2 s = {0.0f, 0.0f};
#pragma unroll 5
            for (int k = 0; k < ks; k++) {   // (unrolled: the LDS reads of five taps go out together)
                const f32x2 v = {(float)src0[k], (float)src1[k]};
                s = s + tp[k] * v;
            }
            hrow[2 * gp * BW + b0 + bxl] = s.x;
            hrow[(2 * gp + 1) * BW + b0 + bxl] = s.y;
            gp += q256; bxl += r256;
            if (bxl >= nb) { bxl -= nb; gp++; }
        }
    }
    __syncthreads();
    // column pass: centre tap, then symmetric pairs
    {
        const int q256 = 256 / nb, r256 = 256 - q256 * nb;
        int by = tid / nb, bxl = tid - by * nb;
        for (; by < BH; ) {
            const float *src = hrow + (by + R) * BW + b0 + bxl;
            float s = tp[R] * src[0];
#pragma unroll 4
            for (int k = 1; k <= R; k++) s = s + tp[R + k] * (src[-k * BW] + src[k * BW]);
            bl[by * BW + b0 + bxl] = s;
            by += q256; bxl += r256;
            if (bxl >= nb) { bxl -= nb; by++; }
        }
    }
    __syncthreads();

    // blurred value at region coordinates (x, y), mirrored at the region border
    auto B = [&](int x, int y) -> float {
        const int xr = reflect101(x, a.rw), yr = reflect101(y, a.rh);
        return bl[(yr - ty0 + HB) * BW + (xr - tx0 + HB)];
    };
    // the decision for one pixel whose brightness test passed; `val` is what the pixel's cell gets
    auto decide = [&](int x, int y, float2 &val, float &resp) -> bool {
        bool acc = false;
        double ox = 0.0, oy = 0.0;
        resp = 0.0f;
        if (VAR == VAR_FAST) {
            const float c = B(x, y), l = B(x - 1, y), r = B(x + 1, y), u = B(x, y - 1), d = B(x, y + 1), ul = B(x - 1, y - 1);
            const float dx = l - c;
            const float dy = u - c;
            const float dxx = (l + (-2.0f * c)) + r;
            const float dyy = (u + (-2.0f * c)) + d;
            const float dxy = ((ul - u) - l) + c;
            acc = steger_point(dx, dy, dxx, dyy, dxy, ox, oy);
        } else {
            // Sobel (ksize 3) of the blurred image at the 3x3 neighbourhood, then Sobel of those: 3-tap
            // passes along x first, then along y, summed as (a + c) + 2b
            float rd[5][3], rs[5][3]; // x-difference and x-smoothing of the blurred rows y-2..y+2 at columns x-1..x+1
#pragma unroll
            for (int j = 0; j < 5; j++)
#pragma unroll
                for (int i = 0; i < 3; i++) {
                    const float l = B(x + i - 2, y + j - 2), c = B(x + i - 1, y + j - 2), r = B(x + i, y + j - 2);
                    rd[j][i] = r - l;
                    rs[j][i] = (l + r) + c * 2.0f;
                }
            float gx[3][3], gy[3][3]; // dx and dy images at rows y-1..y+1, columns x-1..x+1
#pragma unroll
            for (int j = 0; j < 3; j++)
#pragma unroll
                for (int i = 0; i < 3; i++) {
                    gx[j][i] = (rd[j][i] + rd[j + 2][i]) + rd[j + 1][i] * 2.0f;
                    gy[j][i] = rs[j + 2][i] - rs[j][i];
                }
            const float dx = gx[1][1], dy = gy[1][1];
            const float dxx = ((gx[0][2] - gx[0][0]) + (gx[2][2] - gx[2][0])) + (gx[1][2] - gx[1][0]) * 2.0f;
            const float dyy = ((gy[2][0] + gy[2][2]) + gy[2][1] * 2.0f) - ((gy[0][0] + gy[0][2]) + gy[0][1] * 2.0f);
            const float dxy = ((gx[2][0] + gx[2][2]) + gx[2][1] * 2.0f) - ((gx[0][0] + gx[0][2]) + gx[0][1] * 2.0f);
            float lam;
            if (steger2_point(dx, dy, dxx, dyy, dxy, lam, ox, oy)) {
                resp = fabsf(lam);
                if (VAR == VAR_HYBRID) acc = fabs(ox) <= 0.5;
                else acc = fabs(ox) <= a.resp_thr && fabs(oy) <= a.resp_thr;
                if (VAR == VAR_IMPROVED && acc) {
                    const double cx = (double)x + ox, cy = (double)y + oy;
                    acc = 0 <= cx && cx < a.W && 0 <= cy && cy < a.H;
                }
            }
        }
        if (VAR == VAR_FAST || VAR == VAR_IMPROVED)
            val = make_float2((float)((double)x + ox + (double)a.x0), (float)((double)y + oy + (double)a.y0));
        else {
            // the reference drops a row whose best point leaves the image: flagged by a NaN x
            const double bx = (double)x + ox;
            val = make_float2((0 <= bx && bx < a.W) ? (float)bx : __builtin_nanf(""), resp);
        }
        return acc;
    };
    if (VAR == VAR_FAST || VAR == VAR_IMPROVED) {
        // A laser line crosses a tile row in a few of its 64 columns, so deciding row by row would run the
        // double-precision eigen step four times per wave with a tenth of the lanes at work.  The wave first gathers
        // the bright pixels of its four rows, in (row, column) order, into a list (in the row-filtered tile's LDS, free by now)
        // and then takes them 64 at a time, one per lane; a pixel's place in its cell is the number of accepted
        // pixels of its row before it: those of earlier rounds (carry) plus those on lower lanes of this round.
        uint16_t *wlist = reinterpret_cast<uint16_t *>(hrow) + wv * 256;
        int rstart[5];
        rstart[0] = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const bool br = own[k] > a.thr && ty0 + wv + 4 * k < a.rh;
            const unsigned long long bm = __ballot(br);
            if (br) wlist[rstart[k] + __popcll(bm & ((1ull << lane) - 1ull))] = (uint16_t)((k << 6) | lane);
            rstart[k + 1] = rstart[k] + __popcll(bm);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the wave reads back what its own lanes wrote
        int carry[4] = {0, 0, 0, 0};
        const int total = rstart[4];
        for (int c0 = 0; c0 < total; c0 += 64) {
            const int e = c0 + lane;
            const bool valid = e < total;
            const int code = wlist[valid ? e : total - 1];
            const int k = code >> 6, x = tx0 + (code & 63), y = ty0 + wv + 4 * k;
            float2 val = make_float2(0.0f, 0.0f);
            float resp;
            const bool acc = valid && decide(x, y, val, resp);
            const unsigned long long m = __ballot(acc);
            int rank = 0;
#pragma unroll
            for (int kk = 0; kk < 4; kk++) {
                const int lo = max(rstart[kk] - c0, 0), hi = min(rstart[kk + 1] - c0, 64);   // lanes that hold row kk in this round
                unsigned long long rm = 0ull;
                if (hi > lo) rm = (hi >= 64 ? ~0ull : ((1ull << hi) - 1ull)) & ~((1ull << lo) - 1ull);
                if (k == kk) rank = carry[kk] + __popcll(m & rm & ((1ull << lane) - 1ull));
                carry[kk] += __popcll(m & rm);
            }
            if (acc) cells[(cell_base + (size_t)y * a.ntx + txi) * TW + rank] = val;
        }
        if (lane == 0) {
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int y = ty0 + wv + 4 * k;
                if (y < a.rh) cell_counts[cell_base + (size_t)y * a.ntx + txi] = carry[k];
            }
        }
    } else {
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int x = tx0 + lane, y = ty0 + wv + 4 * k;
            if (y >= a.rh) continue; // wave-uniform
            float2 val = make_float2(0.0f, 0.0f);
            float resp = 0.0f;
            const bool acc = own[k] > a.thr && decide(x, y, val, resp);
            const unsigned long long m = __ballot(acc);
            const size_t cell = cell_base + (size_t)y * a.ntx + txi;
            // largest response of the tile row, leftmost pixel on ties: max over (response bits, 63 - lane)
            unsigned long long key = acc ? (((unsigned long long)__float_as_uint(resp) << 8) | (unsigned long long)(63 - lane)) : 0ull;
            unsigned long long best = key;
#pragma unroll
            for (int o = 32; o >= 1; o >>= 1) {
                const unsigned long long other = __shfl_xor(best, o);
                best = other > best ? other : best;
            }
            if (acc && key == best) cells[cell * TW] = val;
            if (lane == 0) cell_counts[cell] = m != 0ull;
        }
    }
    } // tiles
    } // chunks of flags
}

// Most tiles hold no bright pixel, and starting a 256-thread block per tile only to find that out took more than
// half of the extraction time.  A streaming pass flags the tiles that have one (a thread tests sixteen pixels of a row:
// for the hybrid variant the mask, otherwise gray > threshold; the tile kernel repeats the exact test per pixel), and
// the blocks of the tile kernel share the flagged tiles of an image out among themselves.  (A list of the flagged tiles,
// built by a kernel in between, cost a launch and 12 us of additions to one counter; a call is a chain of dependent
// launches of a few tens of microseconds each, and every launch fewer is worth about 8 us.)
// MARK_PX pixels of a row per thread (a divisor of the tile width, so they share a tile); the threads are numbered over
// (image, row, group of MARK_PX) without gaps, whatever the row length.
constexpr int MARK_PX = 16;
__global__ void __launch_bounds__(256) k_steger_mark(StegerArgs a, const uint8_t *__restrict__ img, const uint8_t *__restrict__ mask,
                                                     int *__restrict__ tiles, int groups_per_row, int total)
{
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= total) return;
    const int row = t / groups_per_row, x = (t - row * groups_per_row) * MARK_PX;
    const int b = row / a.rh, y = row - b * a.rh;
    const uint8_t *im = img + (size_t)b * a.W * a.H * a.channels;
    bool bright = false;
    if (!mask && a.channels == 3 && x + MARK_PX <= a.rw) {
        // the pixels as (unaligned) words instead of single bytes: 48 bytes in flight per thread
        uint32_t w[3 * MARK_PX / 4];
        __builtin_memcpy(w, im + ((size_t)(y + a.y0) * a.W + (x + a.x0)) * 3, 3 * MARK_PX);
        auto byte = [&](int i) { return (w[i >> 2] >> (8 * (i & 3))) & 0xffu; };
#pragma unroll
        for (int k = 0; k < MARK_PX; k++)
            bright |= (int)((byte(3 * k + 2) * 9798u + byte(3 * k + 1) * 19235u + byte(3 * k) * 3735u + (1u << 14)) >> 15) > a.thr;
    } else if (!mask && a.channels == 1 && x + MARK_PX <= a.rw) {
        uint32_t w[MARK_PX / 4];
        __builtin_memcpy(w, im + (size_t)(y + a.y0) * a.W + (x + a.x0), MARK_PX);
#pragma unroll
        for (int k = 0; k < MARK_PX; k++) bright |= (int)((w[k >> 2] >> (8 * (k & 3))) & 0xffu) > a.thr;
    } else {
        for (int k = 0; k < MARK_PX; k++) {
            if (x + k < a.rw) {
                if (mask) bright |= mask[(size_t)b * a.W * a.H + (size_t)y * a.W + x + k] != 0;
                else bright |= gray_at(im, a, x + k, y) > a.thr;
            }
        }
    }
    if (bright) tiles[1 + ((size_t)b * a.nty + y / TH) * a.ntx + x / TW] = 1;   // MARK_PX | TW: the pixels share a tile
}

// Clears the tile flags and the per-cell counts: what a call expects to find at zero.  The kernels of a call leave them at
// zero again (the compaction kernel clears both), so this runs only before a handle's first call, after a call that failed
// half way, and once per device for the static work array.
__global__ void __launch_bounds__(256) k_steger_clear(int *__restrict__ tiles, size_t n_tiles, int *__restrict__ cell_counts, size_t n_cells)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x, stride = (size_t)gridDim.x * 256;
    for (size_t k = i; k < n_tiles; k += stride) tiles[k] = 0;
    for (size_t k = i; k < n_cells; k += stride) cell_counts[k] = 0;
}

// per-row variants: the best candidate over the tiles of a row (first tile on ties) becomes the row's only point
__global__ void __launch_bounds__(256) k_steger_rowbest(int nrows_total, int rh, int ntx, float2 *__restrict__ cells, int *__restrict__ cell_counts)
{
    const int row = blockIdx.x * blockDim.x + threadIdx.x; // over batch * rh
    if (row >= nrows_total) return;
    const size_t c0 = (size_t)row * ntx;
    float best_resp = -1.0f, best_x = 0.0f;
    bool have = false;
    for (int t = 0; t < ntx; t++) {
        if (cell_counts[c0 + t]) {
            const float2 v = cells[(c0 + t) * TW];
            if (v.y > best_resp) { best_resp = v.y; best_x = v.x; have = true; }
        }
        cell_counts[c0 + t] = 0;
    }
    if (have && best_x == best_x) {
        cells[c0 * TW] = make_float2(best_x, (float)(row % rh));
        cell_counts[c0] = 1;
    }
}

// one block per image: exclusive scan of the cell counts in raster order; every thread then copies the few
// points of its own cell (a cell holds the accepted pixels of 64 columns of one row: a handful at most on a
// laser line) to where the scan says they go
// (the counts it has read are left at zero for the next call)
__global__ void __launch_bounds__(1024) k_steger_compact(int ncells, const float2 *__restrict__ cells, int *__restrict__ cell_counts,
                                                         float2 *__restrict__ out, int capacity, int *__restrict__ counts,
                                                         int *__restrict__ tiles, int tiles_per_image)
{
    for (int i = threadIdx.x; i < tiles_per_image; i += 1024) tiles[1 + (size_t)blockIdx.x * tiles_per_image + i] = 0;   // the image's tile flags
    __shared__ int wsum[2][16];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    int *cc = cell_counts + (size_t)b * ncells;
    const float2 *cs = cells + (size_t)b * ncells * TW;
    float2 *o = out + (size_t)b * capacity;
    int carry = 0; // points before this chunk: the same in every thread
    for (int base = 0, it = 0; base < ncells; base += 1024, it ^= 1) {
        const int i = base + tid;
        const int v = i < ncells ? cc[i] : 0;
        if (v) cc[i] = 0;
        int s = v; // inclusive scan inside the wave
#pragma unroll
        for (int o2 = 1; o2 < 64; o2 <<= 1) {
            const int t = __shfl_up(s, o2);
            if (lane >= o2) s += t;
        }
        if (lane == 63) wsum[it][wv] = s;
        __syncthreads(); // wsum is double-buffered: one barrier per chunk is enough
        int pre = carry, total = carry;
#pragma unroll
        for (int w = 0; w < 16; w++) {
            const int t = wsum[it][w];
            if (w < wv) pre += t;
            total += t;
        }
        carry = total;
        const int off = pre + s - v;
        const float2 *src = cs + (size_t)i * TW;
        for (int k = 0; k < v && off + k < capacity; k++) o[off + k] = src[k];
    }
    if (tid == 0) counts[b] = carry;
}

} // namespace l3d

using namespace l3d;

struct l3d_steger {
    int device = 0, max_w = 0, max_h = 0, max_batch = 0;
    double sigma = 3.0, resp_thr = 0.5;
    int thr = 200, R = 12;
    bool improved = false; // built by l3d_steger_create_improved: serves the improved / hybrid variants
    hipStream_t stream = nullptr;
    float *taps = nullptr;
    float2 *cells = nullptr;
    int *cell_counts = nullptr;
    int *tile_list = nullptr;   // [1 + t]: tile t holds a bright pixel ([0] unused)
    // The kernels of a call leave the tile flags and the per-cell counts at zero (the compaction kernel clears both),
    // which is what the next call expects.
    // `dirty` says that this may not hold (before the first call, and after a call that failed half way): the next
    // call then clears both arrays first.  The static work array has one such flag per device instead.
    // All calls on one handle must be ordered on the device (one stream, or streams the caller orders with events): the
    // arrays are shared.  A call on another stream than the previous one cannot know that the previous call has run,
    // so it clears first as well (`last`: the previous call's stream; the static array keeps one per device).
    struct StreamNote { hipStream_t stream = nullptr; bool any = false; };
    bool dirty_own = true;
    bool *dirty = nullptr;
    StreamNote last_own;
    StreamNote *last = nullptr;
    size_t max_tiles = 0;
    // staging for the host entry point
    uint8_t *in = nullptr;
    float2 *out = nullptr;
    int *counts = nullptr;
    size_t max_cells = 0;        // cells of one image of the largest shape
    size_t cells_capacity = 0;   // cells the `cells` / `cell_counts` arrays hold in all
};

// cv::GaussianBlur on a float image with ksize (0,0): ksize = round(sigma*4*2 + 1) | 1 taps of
// cv::getGaussianKernel(ks, sigma, CV_32F): exp(-x^2 / 2 sigma^2) normalised in double.  Returns ks, or -ks when
// it exceeds MAX_TAPS.
static int steger_taps(double sigma, float *tf)
{
    const int ks = (int)std::nearbyint(sigma * 4 * 2 + 1) | 1;
    if (ks > MAX_TAPS) return -ks;
    double t[MAX_TAPS];
    double sum = 0;
    const double scale2x = -0.5 / (sigma * sigma);
    for (int i = 0; i < ks; i++) {
        const double x = i - (ks - 1) * 0.5;
        t[i] = std::exp(scale2x * x * x);
        sum += t[i];
    }
    for (int i = 0; i < ks; i++) tf[i] = (float)(t[i] * (1.0 / sum));
    return ks;
}

extern "C" void l3d_steger_destroy(l3d_steger *h)
{
    if (!h) return;
    DeviceGuard dg(h->device);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    void *bufs[] = {h->taps, h->cells, h->cell_counts, h->tile_list, h->in, h->out, h->counts};
    for (void *p : bufs)
        if (p) (void)hipFree(p);
    if (h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
}

static int steger_create(double sigma, int brightness_threshold, int max_width, int max_height, int max_batch, int device, l3d_steger **out);

extern "C" int l3d_steger_create(double sigma, int brightness_threshold, int max_width, int max_height, int max_batch, int device,
                                 l3d_steger **out)
{
    return steger_create(sigma, brightness_threshold, max_width, max_height, max_batch, device, out);
}

extern "C" int l3d_steger_create_improved(double sigma, int brightness_threshold, double response_threshold, int max_width, int max_height,
                                          int max_batch, int device, l3d_steger **out)
{
    const int rc = steger_create(sigma, brightness_threshold, max_width, max_height, max_batch, device, out);
    if (rc) return rc;
    (*out)->improved = true;
    (*out)->resp_thr = response_threshold;
    return L3D_OK;
}

static int steger_create(double sigma, int brightness_threshold, int max_width, int max_height, int max_batch, int device, l3d_steger **out)
{
    L3D_REQUIRE(out, "l3d_steger_create: null argument");
    *out = nullptr;
    L3D_REQUIRE(sigma > 0 && max_width > 0 && max_height > 0 && max_batch > 0, "l3d_steger_create: bad argument");
    float tf[MAX_TAPS];
    const int ks = steger_taps(sigma, tf);
    if (ks < 0) {
        set_error("l3d_steger_create: sigma %.3f needs a %d-tap kernel (max %d)", sigma, -ks, MAX_TAPS);
        return L3D_ERR_UNSUPPORTED;
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        set_error("l3d_steger_create: no HIP device available");
        return L3D_ERR_NO_DEVICE;
    }
    L3D_REQUIRE(device >= 0 && device < ndev, "l3d_steger_create: device %d out of range", device);
    l3d_steger *h = new (std::nothrow) l3d_steger();
    L3D_REQUIRE(h, "l3d_steger_create: out of host memory");
    h->device = device; h->max_w = max_width; h->max_h = max_height; h->max_batch = max_batch;
    h->sigma = sigma; h->thr = brightness_threshold; h->R = ks / 2;
    DeviceGuard dg(device);
    const size_t B = (size_t)max_batch;
    h->max_cells = (size_t)max_height * cdiv(max_width, TW);
    h->cells_capacity = B * h->max_cells;
    hipError_t e = hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking);
    auto alloc = [&](void **ptr, size_t bytes) {
        if (e == hipSuccess) e = hipMalloc(ptr, bytes);
    };
    alloc((void **)&h->taps, MAX_TAPS * sizeof(float));
    alloc((void **)&h->cells, B * h->max_cells * TW * sizeof(float2));
    alloc((void **)&h->cell_counts, B * h->max_cells * sizeof(int));
    h->max_tiles = B * cdiv(max_width, TW) * cdiv(max_height, TH);
    alloc((void **)&h->tile_list, (1 + 2 * h->max_tiles) * sizeof(int));
    alloc((void **)&h->in, B * (size_t)max_width * max_height * 3);
    alloc((void **)&h->out, B * (size_t)max_width * max_height * sizeof(float2));
    alloc((void **)&h->counts, B * sizeof(int));
    if (e == hipSuccess) e = hipMemcpy(h->taps, tf, ks * sizeof(float), hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        set_error("l3d_steger_create: %s", hipGetErrorString(e));
        l3d_steger_destroy(h);
        return L3D_ERR_HIP;
    }
    *out = h;
    return L3D_OK;
}

static int steger_run(l3d_steger *h, const uint8_t *d_img, int channels, int W, int H, int B, const int *roi, float *d_xy, int capacity,
                      int *d_counts, hipStream_t s, int variant = VAR_FAST, const uint8_t *d_mask = nullptr)
{
    StegerArgs a;
    a.W = W; a.H = H; a.channels = channels;
    a.x0 = 0; a.y0 = 0; a.rw = W; a.rh = H;
    if (roi) {
        // numpy slicing gray[y:y+h, x:x+w] for non-negative x, y
        L3D_REQUIRE(roi[0] >= 0 && roi[1] >= 0, "steger: negative roi origin is not supported");
        a.x0 = std::min(roi[0], W); a.y0 = std::min(roi[1], H);
        a.rw = std::max(0, std::min(W, roi[0] + roi[2]) - a.x0);
        a.rh = std::max(0, std::min(H, roi[1] + roi[3]) - a.y0);
    }
    a.R = h->R; a.thr = h->thr; a.resp_thr = h->resp_thr;
    a.ntx = cdiv(std::max(a.rw, 1), TW);
    a.nty = cdiv(std::max(a.rh, 1), TH);
    if (a.rw <= 0 || a.rh <= 0) {
        L3D_HIP_CHECK(hipMemsetAsync(d_counts, 0, (size_t)B * sizeof(int), s));
        return L3D_OK;
    }
    const int R = a.R, HB = variant == VAR_FAST ? 1 : 2;
    const size_t lds = ((size_t)(TH + 2 * HB + 2 * R) * (TW + 2 * HB) + (size_t)(TH + 2 * HB) * (TW + 2 * HB) + (2 * R + 1) + 3) * sizeof(float) +
                       (size_t)(TH + 2 * HB + 2 * R) * (TW + 2 * HB + 2 * R);   // float tiles, taps and flags, then the gray tile in bytes
    a.ntiles = a.ntx * a.nty * B;
    // tiles with bright pixels get a flag; cells of all other tiles are empty
    L3D_REQUIRE((size_t)a.ntiles <= h->max_tiles, "steger: %d tiles exceed the %zu the work buffers hold", a.ntiles, h->max_tiles);
    L3D_REQUIRE((size_t)B * a.rh * a.ntx <= h->cells_capacity, "steger: %zu cells exceed the %zu the work buffers hold", (size_t)B * a.rh * a.ntx,
                h->cells_capacity);
    bool &dirty = h->dirty ? *h->dirty : h->dirty_own;
    l3d_steger::StreamNote &last = h->last ? *h->last : h->last_own;
    if (last.any && last.stream != s) dirty = true;
    last.stream = s; last.any = true;
    if (dirty) {
        const size_t n_tiles = 1 + h->max_tiles, n_cells = h->cells_capacity;
        hipLaunchKernelGGL(k_steger_clear, dim3((unsigned)std::min<size_t>((std::max(n_tiles, n_cells) + 1023) / 1024, 2048)), dim3(256), 0, s,
                           h->tile_list, n_tiles, h->cell_counts, n_cells);
    }
    dirty = true;
    {
        const int groups = cdiv(a.rw, MARK_PX);
        const long long total = (long long)groups * a.rh * B;
        L3D_REQUIRE(total < (1ll << 31), "steger: %lld pixel groups exceed the marking pass's index range", total);
        hipLaunchKernelGGL(k_steger_mark, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, a, d_img, variant == VAR_HYBRID ? d_mask : nullptr,
                           h->tile_list, groups, (int)total);
    }
    // G blocks per image share its flagged tiles out among themselves: about two resident rounds of blocks in all, and
    // no more blocks per image than it has tiles
    // (measured at 256 frames of 640x360 with a laser line, 10,000 of 58,880 tiles flagged: 2048 blocks 151 us per call,
    // 3072 143, 4096 140, 6144 151, 8192 150; without a laser 38 ... 49 us)
    const int G = std::max(1, std::min(std::min(a.ntx * a.nty, 64), 4096 / B));
    const dim3 grid((unsigned)(B * G));
    switch (variant) {
    case VAR_FAST: hipLaunchKernelGGL(k_steger_tile<VAR_FAST>, grid, dim3(256), lds, s, a, d_img, d_mask, h->taps, h->cells, h->cell_counts, h->tile_list); break;
    case VAR_IMPROVED: hipLaunchKernelGGL(k_steger_tile<VAR_IMPROVED>, grid, dim3(256), lds, s, a, d_img, d_mask, h->taps, h->cells, h->cell_counts, h->tile_list); break;
    case VAR_PER_ROW: hipLaunchKernelGGL(k_steger_tile<VAR_PER_ROW>, grid, dim3(256), lds, s, a, d_img, d_mask, h->taps, h->cells, h->cell_counts, h->tile_list); break;
    default: hipLaunchKernelGGL(k_steger_tile<VAR_HYBRID>, grid, dim3(256), lds, s, a, d_img, d_mask, h->taps, h->cells, h->cell_counts, h->tile_list); break;
    }
    if (variant == VAR_PER_ROW || variant == VAR_HYBRID)
        hipLaunchKernelGGL(k_steger_rowbest, dim3(cdiv(B * a.rh, 256)), dim3(256), 0, s, B * a.rh, a.rh, a.ntx, h->cells, h->cell_counts);
    const int ncells = a.rh * a.ntx;
    hipLaunchKernelGGL(k_steger_compact, dim3(B), dim3(1024), 0, s, ncells, h->cells, h->cell_counts, (float2 *)d_xy,
                       capacity, d_counts, h->tile_list, a.ntx * a.nty);
    L3D_HIP_CHECK(hipGetLastError());
    dirty = false;   // every kernel of the call is enqueued: flags and counts are at zero again when they have run
    return L3D_OK;
}

static int steger_check(l3d_steger *h, int channels, int W, int H, int B, int capacity, const char *who)
{
    L3D_REQUIRE(h, "%s: null handle", who);
    L3D_REQUIRE(channels == 1 || channels == 3, "%s: channels must be 1 or 3, got %d", who, channels);
    L3D_REQUIRE(W > 0 && H > 0 && W <= h->max_w && H <= h->max_h, "%s: image %dx%d exceeds the handle's %dx%d", who, W, H, h->max_w, h->max_h);
    L3D_REQUIRE(B >= 1 && B <= h->max_batch, "%s: batch %d outside [1, %d]", who, B, h->max_batch);
    L3D_REQUIRE(capacity >= 0, "%s: negative capacity", who);
    L3D_REQUIRE((size_t)H * cdiv(W, TW) <= h->max_cells, "%s: image shape needs more cells than the handle holds", who);
    return L3D_OK;
}

extern "C" int l3d_steger_extract_device(l3d_steger *h, const uint8_t *d_img, int channels, int width, int height, int batch,
                                         const int *roi, float *d_xy, int capacity, int *d_counts, void *stream)
{
    int rc = steger_check(h, channels, width, height, batch, capacity, "l3d_steger_extract_device");
    if (rc) return rc;
    L3D_REQUIRE(d_img && d_xy && d_counts, "l3d_steger_extract_device: null argument");
    DeviceGuard dg(h->device);
    return steger_run(h, d_img, channels, width, height, batch, roi, d_xy, capacity, d_counts, (hipStream_t)stream);
}

// ---------------------------------------------------------------------------------------------
// l3d_steger_extract_device without any allocation: the work buffers live in a device array of the library
// ---------------------------------------------------------------------------------------------
namespace l3d {
constexpr size_t kStaticCells = 16384;                                     // cells (one image row x 64 columns each)
constexpr size_t kStaticTapsBytes = 512;                                   // MAX_TAPS floats, padded
constexpr size_t kStaticListBytes = (1 + 2 * kStaticCells) * sizeof(int) + 252;   // tile flags + list (tiles <= cells), padded to 256
constexpr size_t kStaticCountBytes = kStaticCells * sizeof(int);
constexpr size_t kStaticCellBytes = kStaticCells * TW * sizeof(float2);
constexpr size_t kStaticBytes = kStaticTapsBytes + kStaticListBytes + kStaticCountBytes + kStaticCellBytes;
static_assert(kStaticListBytes % 256 == 0 && kStaticBytes % 256 == 0, "sub-buffers stay 256-byte aligned");
// Every GPU the library runs on holds its own copy of this array (8.6 MB) until the process ends.  In host code the
// name is only a handle for hipGetSymbolAddress: the per-device address it returns is what kernels and memsets get.
static __device__ __align__(256) __attribute__((used)) unsigned char scratch[kStaticBytes];

struct TapsArg {
    float t[MAX_TAPS];
};
__global__ void k_steger_put_taps(TapsArg v, int n, float *__restrict__ dst)
{
    const int i = threadIdx.x;
    if (i < n) dst[i] = v.t[i];
}

constexpr int kMaxDevices = 64;
static unsigned char *g_scratch_ptr[kMaxDevices];   // per device, set once (racing first calls store the same value)

// address of this device's copy of `scratch`; `device` must be the current device
static int static_scratch(int device, unsigned char **out)
{
    L3D_REQUIRE(device >= 0 && device < kMaxDevices, "static scratch: device %d out of range", device);
    unsigned char *p = __atomic_load_n(&g_scratch_ptr[device], __ATOMIC_ACQUIRE);
    if (!p) {
        void *sym = nullptr;
        L3D_HIP_CHECK(hipGetSymbolAddress(&sym, HIP_SYMBOL(scratch)));
        L3D_REQUIRE(sym && ((uintptr_t)sym & 255) == 0, "static scratch: hipGetSymbolAddress returned %p", sym);
        p = (unsigned char *)sym;
        __atomic_store_n(&g_scratch_ptr[device], p, __ATOMIC_RELEASE);
    }
    *out = p;
    return L3D_OK;
}
} // namespace l3d

extern "C" int l3d_steger_extract_device_static(double sigma, int brightness_threshold, const uint8_t *d_img, int channels, int width,
                                                int height, int batch, const int *roi, float *d_xy, int capacity, int *d_counts, int device,
                                                void *stream)
{
    const char *who = "l3d_steger_extract_device_static";
    L3D_REQUIRE(d_img && d_xy && d_counts, "%s: null argument", who);
    L3D_REQUIRE(channels == 1 || channels == 3, "%s: channels must be 1 or 3, got %d", who, channels);
    L3D_REQUIRE(sigma > 0 && width > 0 && height > 0 && batch >= 1 && capacity >= 0, "%s: bad argument", who);
    // the whole image must fit even when a region of interest makes the job smaller: one rule, easy to state
    const size_t cells = (size_t)batch * height * cdiv(width, TW);
    if (cells > kStaticCells) {
        set_error("%s: %d image(s) of %dx%d need %zu cells (rows x 64-column tiles); the static work buffer holds %zu "
                  "(e.g. one 1280x720 frame or four 640x360 frames) -- use l3d_steger_extract_device with a handle for more",
                  who, batch, width, height, cells, kStaticCells);
        return L3D_ERR_CAPACITY;
    }
    TapsArg taps;
    const int ks = steger_taps(sigma, taps.t);
    if (ks < 0) {
        set_error("%s: sigma %.3f needs a %d-tap kernel (max %d)", who, sigma, -ks, MAX_TAPS);
        return L3D_ERR_UNSUPPORTED;
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        set_error("%s: no HIP device available", who);
        return L3D_ERR_NO_DEVICE;
    }
    L3D_REQUIRE(device >= 0 && device < ndev, "%s: device %d out of range (%d devices)", who, device, ndev);
    DeviceGuard dg(device);
    unsigned char *base = nullptr;
    int rc = static_scratch(device, &base);
    if (rc) return rc;
    // a handle-shaped view of the static array (nothing in it is owned)
    l3d_steger v;
    v.device = device; v.max_w = width; v.max_h = height; v.max_batch = batch;
    v.sigma = sigma; v.thr = brightness_threshold; v.R = ks / 2;
    v.taps = reinterpret_cast<float *>(base);
    v.tile_list = reinterpret_cast<int *>(base + kStaticTapsBytes);
    v.cell_counts = reinterpret_cast<int *>(base + kStaticTapsBytes + kStaticListBytes);
    v.cells = reinterpret_cast<float2 *>(base + kStaticTapsBytes + kStaticListBytes + kStaticCountBytes);
    v.max_tiles = kStaticCells;
    // One flag and one stream note per device, constant-initialised (no set-up step that first calls on different
    // GPUs could race on): the device array is zero when the library is loaded, which is the state a call expects, so
    // a device starts clean; it is marked again by a call that fails half way (steger_run).  Threads that call for
    // different GPUs touch different elements; calls for one GPU must not overlap anyway.
    static bool static_dirty[kMaxDevices] = {};
    static l3d_steger::StreamNote static_last[kMaxDevices] = {};
    v.dirty = &static_dirty[device];
    v.last = &static_last[device];
    v.max_cells = (size_t)height * cdiv(width, TW);
    v.cells_capacity = kStaticCells;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(k_steger_put_taps, dim3(1), dim3(MAX_TAPS < 64 ? 64 : 128), 0, s, taps, ks, v.taps);
    return steger_run(&v, d_img, channels, width, height, batch, roi, d_xy, capacity, d_counts, s);
}

// host images in, points out, through the handle's staging buffers
static int steger_host(l3d_steger *h, const uint8_t *img, int channels, int width, int height, int batch, const int *roi, float *xy, int capacity,
                       int *counts, int variant, const char *who)
{
    int rc = steger_check(h, channels, width, height, batch, capacity, who);
    if (rc) return rc;
    L3D_REQUIRE(img && xy && counts, "%s: null argument", who);
    DeviceGuard dg(h->device);
    hipStream_t s = h->stream;
    const size_t px = (size_t)width * height;
    const int cap_dev = (int)std::min<size_t>((size_t)capacity, px);
    L3D_HIP_CHECK(hipMemcpyAsync(h->in, img, px * channels * batch, hipMemcpyHostToDevice, s));
    rc = steger_run(h, h->in, channels, width, height, batch, roi, (float *)h->out, cap_dev, h->counts, s, variant);
    if (rc) return rc;
    L3D_HIP_CHECK(hipMemcpyAsync(counts, h->counts, (size_t)batch * sizeof(int), hipMemcpyDeviceToHost, s));
    L3D_HIP_CHECK(hipStreamSynchronize(s));
    bool overflow = false;
    for (int b = 0; b < batch; b++) {
        const int n = std::min(counts[b], cap_dev);
        if (counts[b] > capacity) overflow = true;
        if (n > 0) L3D_HIP_CHECK(hipMemcpy(xy + (size_t)b * capacity * 2, h->out + (size_t)b * cap_dev, (size_t)n * sizeof(float2), hipMemcpyDeviceToHost));
    }
    if (overflow) {
        set_error("%s: more points than capacity %d", who, capacity);
        return L3D_ERR_CAPACITY;
    }
    return L3D_OK;
}

extern "C" int l3d_steger_extract(l3d_steger *h, const uint8_t *img, int channels, int width, int height, int batch, const int *roi,
                                  float *xy, int capacity, int *counts)
{
    return steger_host(h, img, channels, width, height, batch, roi, xy, capacity, counts, VAR_FAST, "l3d_steger_extract");
}

extern "C" int l3d_steger_extract_improved(l3d_steger *h, const uint8_t *img, int channels, int width, int height, int batch, int per_row,
                                           float *xy, int capacity, int *counts)
{
    L3D_REQUIRE(h && h->improved, "l3d_steger_extract_improved: the handle was not made by l3d_steger_create_improved");
    return steger_host(h, img, channels, width, height, batch, nullptr, xy, capacity, counts, per_row ? VAR_PER_ROW : VAR_IMPROVED,
                       "l3d_steger_extract_improved");
}

extern "C" int l3d_steger_extract_improved_device(l3d_steger *h, const uint8_t *d_img, int channels, int width, int height, int batch, int per_row,
                                                  float *d_xy, int capacity, int *d_counts, void *stream)
{
    L3D_REQUIRE(h && h->improved, "l3d_steger_extract_improved_device: the handle was not made by l3d_steger_create_improved");
    int rc = steger_check(h, channels, width, height, batch, capacity, "l3d_steger_extract_improved_device");
    if (rc) return rc;
    L3D_REQUIRE(d_img && d_xy && d_counts, "l3d_steger_extract_improved_device: null argument");
    DeviceGuard dg(h->device);
    return steger_run(h, d_img, channels, width, height, batch, nullptr, d_xy, capacity, d_counts, (hipStream_t)stream,
                      per_row ? VAR_PER_ROW : VAR_IMPROVED);
}

extern "C" int l3d_hybrid_extract_device(l3d_simple *mask_maker, l3d_steger *h, const uint8_t *d_bgr, int width, int height, const int *hsv_lower,
                                         const int *hsv_upper, int brightness_threshold, float *d_xy, int capacity, int *d_count, void *stream)
{
    L3D_REQUIRE(mask_maker && h && h->improved, "l3d_hybrid_extract_device: needs a l3d_simple handle and a handle made by l3d_steger_create_improved");
    int rc = steger_check(h, 3, width, height, 1, capacity, "l3d_hybrid_extract_device");
    if (rc) return rc;
    L3D_REQUIRE(d_bgr && hsv_lower && hsv_upper && d_xy && d_count, "l3d_hybrid_extract_device: null argument");
    DeviceGuard dg(h->device);
    const uint8_t *d_mask = nullptr;
    rc = simple_clean_mask(mask_maker, d_bgr, width, height, hsv_lower, hsv_upper, brightness_threshold, (hipStream_t)stream, &d_mask);
    if (rc) return rc;
    return steger_run(h, d_bgr, 3, width, height, 1, nullptr, d_xy, capacity, d_count, (hipStream_t)stream, VAR_HYBRID, d_mask);
}

extern "C" int l3d_hybrid_extract(l3d_simple *mask_maker, l3d_steger *h, const uint8_t *bgr, int width, int height, const int *hsv_lower,
                                  const int *hsv_upper, int brightness_threshold, float *xy, int capacity, int *count, uint8_t *mask)
{
    L3D_REQUIRE(mask_maker && h && h->improved, "l3d_hybrid_extract: needs a l3d_simple handle and a handle made by l3d_steger_create_improved");
    int rc = steger_check(h, 3, width, height, 1, capacity, "l3d_hybrid_extract");
    if (rc) return rc;
    L3D_REQUIRE(bgr && hsv_lower && hsv_upper && xy && count, "l3d_hybrid_extract: null argument");
    DeviceGuard dg(h->device);
    hipStream_t s = h->stream;
    const size_t px = (size_t)width * height;
    const int cap_dev = (int)std::min<size_t>((size_t)capacity, px);
    L3D_HIP_CHECK(hipMemcpyAsync(h->in, bgr, px * 3, hipMemcpyHostToDevice, s));
    const uint8_t *d_mask = nullptr;
    rc = simple_clean_mask(mask_maker, h->in, width, height, hsv_lower, hsv_upper, brightness_threshold, s, &d_mask);
    if (rc) return rc;
    rc = steger_run(h, h->in, 3, width, height, 1, nullptr, (float *)h->out, cap_dev, h->counts, s, VAR_HYBRID, d_mask);
    if (rc) return rc;
    L3D_HIP_CHECK(hipMemcpyAsync(count, h->counts, sizeof(int), hipMemcpyDeviceToHost, s));
    if (mask) L3D_HIP_CHECK(hipMemcpyAsync(mask, d_mask, px, hipMemcpyDeviceToHost, s));
    L3D_HIP_CHECK(hipStreamSynchronize(s));
    const int n = std::min(*count, cap_dev);
    if (n > 0) L3D_HIP_CHECK(hipMemcpy(xy, h->out, (size_t)n * sizeof(float2), hipMemcpyDeviceToHost));
    if (*count > capacity) {
        set_error("l3d_hybrid_extract: more points than capacity %d", capacity);
        return L3D_ERR_CAPACITY;
    }
    return L3D_OK;
}
