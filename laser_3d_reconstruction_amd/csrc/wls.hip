// Confidence-weighted disparity refinement with the Fast Global Smoother, for gfx950.
//
// Replaces cv2.ximgproc.createDisparityWLSFilter(left_matcher).filter(dL, left_gray,
// disparity_map_right=dR) of the reference (camera/single_usb_stereo_camera.py:280-282, 328-332).
//
// All work happens on the valid-disparity sub-image (the columns x >= minD + D of the left view,
// x < W - (minD + D) of the right view), width rw.  Planes in HBM, per handle:
//   discL, discR [B][H][rw] f32   depth-discontinuity confidence of each view
//   u1, u2       [B][H][rw] f32   confidence * disparity and confidence (smoothed in place)
//   cv           [B][H][rw] f32   vertical edge weights  -exp(-|I(y,x) - I(y+1,x)| / sigma)
//   ch           [B][H][rw] f32   horizontal edge weights
//   inter        [B][max(H,rw)][min] f32  normalised upper diagonal of the forward sweep
//   conf         [B][H][W]  f32   confidence map (x255) in full-image coordinates
//   (t1, t2, chT [B][rw][H] f32: u1, u2 and ch transposed, only when rw is not a multiple of 4)
// The smoother's tridiagonal solves are sequential along a line and independent across lines, so
// both passes run as one thread per line.  In the vertical pass adjacent lines are adjacent in
// memory and every access of a wave is coalesced as it is (k_fgs_sweep).  In the horizontal pass a
// line is a row of the plane: a wave moves 64 x 16 tiles of 64 rows as 16-byte pieces and turns
// them through LDS (k_fgs_sweep_rows) -- no transposed copies of the planes, which cost two more
// passes over both planes per iteration.  Sub-images whose width is not a multiple of 4 take the
// older route: transpose, k_fgs_sweep, transpose back.
#include "common.h"
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <new>
#include <vector>

namespace l3d {

#ifndef L3D_DT_H
#define L3D_DT_H 32
#endif
constexpr int DT_W = 64, DT_H = L3D_DT_H; // discontinuity-map tile
constexpr int DT_XS = 16, DT_YS = 8; // outputs a thread produces by sliding its window along a row / down a column
#ifndef L3D_DT_LD
#define L3D_DT_LD 13
#endif
constexpr int DT_LD = L3D_DT_LD;     // tile values a thread requests together (13: the whole tile of the common radius 5 in one go)

__device__ __forceinline__ int wls_reflect101(int p, int n)
{
    if (n == 1) return 0;
    while (p < 0 || p >= n) p = p < 0 ? -p : 2 * (n - 1) - p;
    return p;
}

// LDS of k_wls_disc in bytes: the row sums of d^2 (64-bit) and of d (int), pitch 65, and the disparity tile
__host__ __device__ inline size_t wls_disc_lds(int r)
{
    const int GH = DT_H + 2 * r, GWp = ((DT_W + 2 * r) / 2 | 1) * 2;      // 16-bit entries, an odd number of words a row
    return (size_t)GH * GWp * sizeof(int16_t) + 8 + (size_t)GH * (DT_W + 1) * (sizeof(int) + sizeof(long long));
}

// var = box(d^2) - box(d)^2 over (2r+1)^2, the sub-image extended by BORDER_REFLECT_101 (cv::boxFilter's default on the
// float copy of the sub-image), conf = max(0, 1 - rolloff * var); box sums are exact integers, scaled by the double
// 1 / (window area) and rounded to float once, as boxFilter's double accumulator does.
// Both views in one launch (blockIdx.z = 2 * frame + view).  The box is two passes of running sums over a 64 x 16 tile in
// LDS: along the rows a thread owns 16 outputs of one tile row (window start-up, then one value in and one out per
// step), down the columns 8 outputs of one column -- about 3 LDS reads per output and pass instead of 2r + 1.  Threads
// of a wave walk adjacent rows in the first pass (the tile's pitch is odd, so they sit on different banks) and
// adjacent columns in the second.
__global__ void __launch_bounds__(256) k_wls_disc(const int16_t *__restrict__ dispL, const int16_t *__restrict__ dispR, int W, int H, int xL,
                                                  int xR, int rw, int r, float rolloff, float *__restrict__ outL, float *__restrict__ outR)
{
    extern __shared__ int smi[];
    const int GW = DT_W + 2 * r, GH = DT_H + 2 * r, GWp = (GW / 2 | 1) * 2, HP = DT_W + 1;
    long long *h2 = reinterpret_cast<long long *>(smi);   // [GH][HP] row sums of d^2
    int *h1 = reinterpret_cast<int *>(h2 + GH * HP);      // [GH][HP] row sums of d
    int16_t *tile = reinterpret_cast<int16_t *>(h1 + GH * HP);   // [GH][GWp] disparity
    const int tid = threadIdx.x;
    const int tx0 = blockIdx.x * DT_W, ty0 = blockIdx.y * DT_H, b = blockIdx.z >> 1, side = blockIdx.z & 1;
    const int16_t *d = (side ? dispR : dispL) + (size_t)b * W * H;
    const int x0 = side ? xR : xL;
    float *out = side ? outR : outL;
    // (DT_LD values a thread are requested before the first is stored: a load per turn of the loop left the block waiting
    // for memory a dozen times in a row)
    for (int i0 = tid; i0 < GH * GW; i0 += DT_LD * 256) {
        int v[DT_LD], at[DT_LD];
#pragma unroll
        for (int k = 0; k < DT_LD; k++) {
            const int i = min(i0 + k * 256, GH * GW - 1);
            const int gy = i / GW, gx = i - gy * GW;
            const int x = wls_reflect101(tx0 - r + gx, rw), y = wls_reflect101(ty0 - r + gy, H);
            at[k] = gy * GWp + gx;
            v[k] = d[(size_t)y * W + x0 + x];
        }
#pragma unroll
        for (int k = 0; k < DT_LD; k++)
            if (i0 + k * 256 < GH * GW) tile[at[k]] = (int16_t)v[k];
    }
    __syncthreads();
    const int win = 2 * r + 1;
    for (int i = tid; i < GH * (DT_W / DT_XS); i += 256) {
        const int seg = i / GH, gy = i - seg * GH;
        const int16_t *row = tile + gy * GWp + seg * DT_XS;
        int a = 0;
        long long q = 0;
#pragma unroll 4
        for (int j = 0; j < win; j++) {
            const int v = row[j];
            a += v;
            q += (long long)(v * v);        // |v| < 2^15: the square fits 32 bits
        }
        int *o1 = h1 + gy * HP + seg * DT_XS;
        long long *o2 = h2 + gy * HP + seg * DT_XS;
        o1[0] = a;
        o2[0] = q;
#pragma unroll 5
        for (int x = 1; x < DT_XS; x++) {
            const int vin = row[x + win - 1], vout = row[x - 1];
            a += vin - vout;
            q += (long long)(vin * vin) - (long long)(vout * vout);
            o1[x] = a;
            o2[x] = q;
        }
    }
    __syncthreads();
    const double scale = 1.0 / (win * win);
    for (int i = tid; i < DT_W * (DT_H / DT_YS); i += 256) {
        const int vs = i / DT_W, x = i - vs * DT_W;
        const int *c1 = h1 + vs * DT_YS * HP + x;
        const long long *c2 = h2 + vs * DT_YS * HP + x;
        long long a = 0, q = 0;
#pragma unroll 4
        for (int j = 0; j < win; j++) {
            a += c1[j * HP];
            q += c2[j * HP];
        }
        const bool col_ok = tx0 + x < rw;
        float *o = out + ((size_t)b * H + ty0 + vs * DT_YS) * rw + tx0 + x;
#pragma unroll 4
        for (int k = 0; k < DT_YS; k++) {
            if (k > 0) {
                a += c1[(k + win - 1) * HP] - c1[(k - 1) * HP];
                q += c2[(k + win - 1) * HP] - c2[(k - 1) * HP];
            }
            const float m1 = (float)((double)a * scale);
            const float m2 = (float)((double)q * scale);
            const float var = m2 - m1 * m1;
            const float c = 1.0f - rolloff * var;
            if (col_ok && ty0 + vs * DT_YS + k < H) o[(size_t)k * rw] = c > 0.0f ? c : 0.0f;
        }
    }
}

// left-right consistency + discontinuity confidence; fills u1 = conf*dL, u2 = conf, and the
// vertical / transposed-horizontal edge weights of the guide image
__global__ void __launch_bounds__(256) k_wls_conf(const int16_t *__restrict__ dL, const int16_t *__restrict__ dR,
                                                  const uint8_t *__restrict__ guide, const float *__restrict__ lut, int W, int H, int lx0, int rx0,
                                                  int rw, int thresh, const float *__restrict__ discL, const float *__restrict__ discR,
                                                  float *__restrict__ u1, float *__restrict__ u2, float *__restrict__ cv, float *__restrict__ ch,
                                                  float *__restrict__ chT, float *__restrict__ conf_full)
{
    // one thread per image column: the columns outside the sub-image only get their zero in the confidence map
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y, b = blockIdx.z;
    if (j >= W) return;
    const size_t img = ((size_t)b * H + y) * W;
    const int x = j - lx0;
    if (x < 0 || x >= rw) {
        conf_full[img + j] = 0.0f;
        return;
    }
    const size_t sub = ((size_t)b * H + y) * rw;
    const int dl = dL[img + j];
    const int ridx = j - (dl >> 4);
    float c = 0.0f;
    if (ridx >= rx0 && ridx < rx0 + rw) {
        if (abs(dl + (int)dR[img + ridx]) < thresh) {
            const float a = discL[sub + x], q = discR[sub + (ridx - rx0)];
            c = a < q ? a : q;
        }
    }
    c = 255.0f * c;
    u2[sub + x] = c;
    u1[sub + x] = c * (float)dl;
    conf_full[img + j] = c;
    const int g0 = guide[img + j];
    cv[sub + x] = (y < H - 1) ? lut[abs(g0 - (int)guide[img + W + j])] : 0.0f;
    const float wh = (x < rw - 1) ? lut[abs(g0 - (int)guide[img + j + 1])] : 0.0f;
    if (ch) ch[sub + x] = wh;                             // row sweeps on the planes as they lie
    else chT[((size_t)b * rw + x) * H + y] = wh;          // sweeps on transposed planes
}

// k_wls_conf for shapes whose rows split into whole groups of four pixels (W, lx0 and rw multiples of 4): a thread takes
// four adjacent pixels, reads its disparities and guide values as one 8- and two 4-byte words and stores every plane as
// 16-byte pieces.  Same operations per pixel as k_wls_conf.
__global__ void __launch_bounds__(256) k_wls_conf4(const int16_t *__restrict__ dL, const int16_t *__restrict__ dR,
                                                   const uint8_t *__restrict__ guide, const float *__restrict__ lut, int W, int H, int lx0, int rx0,
                                                   int rw, int thresh, const float *__restrict__ discL, const float *__restrict__ discR,
                                                   float *__restrict__ u1, float *__restrict__ u2, float *__restrict__ cv, float *__restrict__ ch,
                                                   float *__restrict__ conf_full)
{
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    typedef short s16x4 __attribute__((ext_vector_type(4)));
    const int j0 = 4 * (blockIdx.x * blockDim.x + threadIdx.x);
    const int y = blockIdx.y, b = blockIdx.z;
    if (j0 >= W) return;
    const size_t img = ((size_t)b * H + y) * W;
    const int x0 = j0 - lx0;
    if (x0 < 0 || x0 >= rw) {
        *reinterpret_cast<f32x4 *>(conf_full + img + j0) = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        return;
    }
    const size_t sub = ((size_t)b * H + y) * rw;
    const s16x4 dl4 = *reinterpret_cast<const s16x4 *>(dL + img + j0);
    const uint32_t g4 = *reinterpret_cast<const uint32_t *>(guide + img + j0);
    const uint32_t gd4 = (y < H - 1) ? *reinterpret_cast<const uint32_t *>(guide + img + W + j0) : 0u;
    const int gnext = (x0 + 4 < rw) ? guide[img + j0 + 4] : 0;
    const f32x4 a4 = *reinterpret_cast<const f32x4 *>(discL + sub + x0);
    int ridx[4];
    int dr[4];
    float q[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        ridx[k] = j0 + k - (dl4[k] >> 4);
        const bool in = ridx[k] >= rx0 && ridx[k] < rx0 + rw;
        dr[k] = in ? (int)dR[img + ridx[k]] : 0;
        q[k] = in ? discR[sub + (ridx[k] - rx0)] : 0.0f;
    }
    f32x4 c4, p4, v4, h4;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int dl = dl4[k];
        float c = 0.0f;
        if (ridx[k] >= rx0 && ridx[k] < rx0 + rw) {
            if (abs(dl + dr[k]) < thresh) c = a4[k] < q[k] ? a4[k] : q[k];
        }
        c = 255.0f * c;
        c4[k] = c;
        p4[k] = c * (float)dl;
        const int g0 = (int)((g4 >> (8 * k)) & 0xffu);
        v4[k] = (y < H - 1) ? lut[abs(g0 - (int)((gd4 >> (8 * k)) & 0xffu))] : 0.0f;
        const int gr = k < 3 ? (int)((g4 >> (8 * k + 8)) & 0xffu) : gnext;
        h4[k] = (x0 + k < rw - 1) ? lut[abs(g0 - gr)] : 0.0f;
    }
    *reinterpret_cast<f32x4 *>(u2 + sub + x0) = c4;
    *reinterpret_cast<f32x4 *>(u1 + sub + x0) = p4;
    *reinterpret_cast<f32x4 *>(conf_full + img + j0) = c4;
    *reinterpret_cast<f32x4 *>(cv + sub + x0) = v4;
    *reinterpret_cast<f32x4 *>(ch + sub + x0) = h4;
}

// out[b][c][r] = in[b][r][c] for two planes at once; 32x32 tiles through LDS
__global__ void __launch_bounds__(256) k_transpose2(const float *__restrict__ in1, const float *__restrict__ in2, int rows, int cols,
                                                    float *__restrict__ out1, float *__restrict__ out2)
{
    __shared__ float t1[32][33], t2[32][33];
    const int b = blockIdx.z;
    const size_t plane = (size_t)rows * cols;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5; // 32 x 8
    const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
    for (int k = ty; k < 32; k += 8) {
        const int r = r0 + k, c = c0 + tx;
        if (r < rows && c < cols) {
            t1[k][tx] = in1[b * plane + (size_t)r * cols + c];
            t2[k][tx] = in2[b * plane + (size_t)r * cols + c];
        }
    }
    __syncthreads();
    for (int k = ty; k < 32; k += 8) {
        const int c = c0 + k, r = r0 + tx;
        if (r < rows && c < cols) {
            out1[b * plane + (size_t)c * rows + r] = t1[tx][k];
            out2[b * plane + (size_t)c * rows + r] = t2[tx][k];
        }
    }
}

// Thomas algorithm down every column of a [B][n][m] plane pair (m lines of length n, adjacent
// lines adjacent in memory).  w[i] is the (negative) weight between element i and i+1.
// One thread solves one line, in the operation order of the sequential definition.  The recurrences are short
// dependent chains, so what a thread would wait for is memory: the operands of FGS_CHUNK elements are requested
// together, one chunk ahead of their use, instead of one element at a time.
constexpr int FGS_CHUNK = 16;

// filtered disparity of one pixel from the two smoothed planes: u1 / u2 rounded to int16 (cv::Mat::convertTo), NaN -> INT16_MIN
__device__ __forceinline__ int wls_final_value(float a1, float a2)
{
    const float t = 1.0f / (a2 + 1e-43f);
    const float v = a1 * t;
    if (!(v == v)) return -32768;
    if (v >= 32767.0f) return 32767;
    if (v <= -32768.0f) return -32768;
    return (int)rintf(v);
}

// FINAL: the last sweep of the filter (a vertical pass: line = column of the sub-image, element = image row).  Its back
// substitution yields the smoothed planes' final values, which nothing reads but the division that makes the output:
// they go straight into `out` (int16, full-image coordinates) and are not stored.
template <bool FINAL>
__global__ void __launch_bounds__(64) k_fgs_sweep(float *__restrict__ u1, float *__restrict__ u2, const float *__restrict__ w,
                                                  float *__restrict__ inter, int n, int m, float lambda, int16_t *__restrict__ out, int W, int lx0)
{
    const int line = blockIdx.x * blockDim.x + threadIdx.x;
    const int b = blockIdx.y;
    if (line >= m) return;
    // A wave alone on its SIMD is bound by the length of its instruction stream, so addresses cost nothing here: every
    // access is a buffer access with the line as per-lane offset and the element as a scalar offset, and the element
    // loops have no per-element bound checks outside their last, partial chunk.
    const size_t pl = (size_t)b * n * m;
    const BufRsrc r1 = make_rsrc(u1 + pl), r2 = make_rsrc(u2 + pl), rw = make_rsrc(w + pl), ri = make_rsrc(inter + pl);
    const uint32_t voff = (uint32_t)line * 4u, rowb = (uint32_t)m * 4u;
    auto ldf = [&](BufRsrc r, int i) __attribute__((always_inline)) {
        // (the element index is the same in every lane; saying so keeps the compiler from looping over its values)
        return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, __builtin_amdgcn_readfirstlane((uint32_t)i * rowb), 0));
    };
    auto stf = [&](BufRsrc r, int i, float v) __attribute__((always_inline)) {
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, v), r, voff, __builtin_amdgcn_readfirstlane((uint32_t)i * rowb), 0);
    };
    float lower = 0.0f, prev_inter = 0.0f, p1 = 0.0f, p2 = 0.0f;
    float cw[FGS_CHUNK], c1[FGS_CHUNK], c2[FGS_CHUNK], nw[FGS_CHUNK], n1[FGS_CHUNK], n2[FGS_CHUNK];
    auto load_fwd = [&](int i0, float (&xw)[FGS_CHUNK], float (&x1)[FGS_CHUNK], float (&x2)[FGS_CHUNK]) __attribute__((always_inline)) {
#pragma unroll
        for (int k = 0; k < FGS_CHUNK; k++) {
            const int i = min(i0 + k, n - 1);   // clamped: the tail of the last chunk is loaded, not used
            xw[k] = ldf(rw, i); x1[k] = ldf(r1, i); x2[k] = ldf(r2, i);
        }
    };
    auto fwd = [&](int i, int k) __attribute__((always_inline)) {
        const float upper = lambda * cw[k];
        const float denom = (1.0f - lower - upper) - lower * prev_inter;
        const float ci = upper / denom;
        const float v1 = (c1[k] - lower * p1) / denom;
        const float v2 = (c2[k] - lower * p2) / denom;
        stf(ri, i, ci);
        stf(r1, i, v1);
        stf(r2, i, v2);
        prev_inter = ci; p1 = v1; p2 = v2;
        lower = upper;
    };
    load_fwd(0, cw, c1, c2);
    int i0 = 0;
    for (; i0 + FGS_CHUNK <= n; i0 += FGS_CHUNK) {
        if (i0 + FGS_CHUNK < n) load_fwd(i0 + FGS_CHUNK, nw, n1, n2);
#pragma unroll
        for (int k = 0; k < FGS_CHUNK; k++) fwd(i0 + k, k);
#pragma unroll
        for (int k = 0; k < FGS_CHUNK; k++) { cw[k] = nw[k]; c1[k] = n1[k]; c2[k] = n2[k]; }
    }
#pragma unroll
    for (int k = 0; k < FGS_CHUNK; k++)
        if (i0 + k < n) fwd(i0 + k, k);
    // back substitution from element n-2 down; the values it reads were written above by this same thread
    auto load_bwd = [&](int j0, float (&xw)[FGS_CHUNK], float (&x1)[FGS_CHUNK], float (&x2)[FGS_CHUNK]) __attribute__((always_inline)) {
#pragma unroll
        for (int k = 0; k < FGS_CHUNK; k++) {
            const int i = max(j0 - k, 0);
            xw[k] = ldf(ri, i); x1[k] = ldf(r1, i); x2[k] = ldf(r2, i);
        }
    };
    int16_t *oline = FINAL ? out + (size_t)b * n * W + lx0 + line : nullptr;
    auto bwd = [&](int i, int k) __attribute__((always_inline)) {
        p1 = c1[k] - cw[k] * p1;
        p2 = c2[k] - cw[k] * p2;
        if (FINAL)
            oline[(size_t)i * W] = (int16_t)wls_final_value(p1, p2);
        else {
            stf(r1, i, p1);
            stf(r2, i, p2);
        }
    };
    if (FINAL) oline[(size_t)(n - 1) * W] = (int16_t)wls_final_value(p1, p2);   // the last element keeps its value of the elimination
    if (n >= 2) load_bwd(n - 2, cw, c1, c2);
    int j0 = n - 2;
    for (; j0 - (FGS_CHUNK - 1) >= 0; j0 -= FGS_CHUNK) {
        if (j0 - FGS_CHUNK >= 0) load_bwd(j0 - FGS_CHUNK, nw, n1, n2);
#pragma unroll
        for (int k = 0; k < FGS_CHUNK; k++) bwd(j0 - k, k);
#pragma unroll
        for (int k = 0; k < FGS_CHUNK; k++) { cw[k] = nw[k]; c1[k] = n1[k]; c2[k] = n2[k]; }
    }
#pragma unroll
    for (int k = 0; k < FGS_CHUNK; k++)
        if (j0 - k >= 0) bwd(j0 - k, k);
}

// The same solves for lines whose elements are adjacent in memory (the rows of a row-major plane: the horizontal pass
// without transposed copies of the planes), 64 lines to a block of one wave.  Per chunk of FGS_RCHUNK elements the wave
// moves a 64 x 32 tile of each operand between memory and registers in 16-byte pieces (a lane: an eighth of one line's
// chunk, 8 lines per access, so an access covers 8 runs of 128 contiguous bytes: whole cache lines -- with 16-element
// chunks a wave touched 64-byte halves of lines and the sweep moved 6.8 GB where the column sweep moves 4.8), and the
// tile is turned through LDS so that every thread ends up with the chunk of its own line.  What a thread computes on its line is the sequence of
// k_fgs_sweep; only the route of the operands differs, so the results are identical.  n (the line length) is a multiple of 4.
// `inter` keeps the layout of k_fgs_sweep ([element][line]): it is private to the sweep.
constexpr int FGS_RCHUNK = 32;            // elements of a line per chunk: 128 contiguous bytes a line and access
constexpr int FGS_PPL = FGS_RCHUNK / 4;   // 16-byte pieces of a line's chunk = lanes that share a line in a tile move
constexpr int FGS_LPA = 64 / FGS_PPL;     // lines per access of the wave
constexpr int FGS_TS = FGS_RCHUNK + 4;    // floats between the lines of a tile in LDS

__global__ void __launch_bounds__(64) k_fgs_sweep_rows(float *__restrict__ u1, float *__restrict__ u2, const float *__restrict__ w,
                                                       float *__restrict__ inter, int n, int m, float lambda)
{
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    typedef uint32_t u32x4w __attribute__((ext_vector_type(4)));
    __shared__ __align__(16) float tile[3][64 * FGS_TS];
    const int lane = threadIdx.x, l0 = blockIdx.x * 64, b = blockIdx.y;
    const bool line_ok = l0 + lane < m;
    const int line = min(l0 + lane, m - 1);                 // threads past the last line redo it and store nothing
    const size_t pl = (size_t)b * n * m;
    const BufRsrc r1 = make_rsrc(u1 + pl), r2 = make_rsrc(u2 + pl), rw = make_rsrc(w + pl), ri = make_rsrc(inter + pl);
    const uint32_t ivoff = (uint32_t)line * 4u, irowb = (uint32_t)m * 4u;
    // role in the tile moves: piece cq of the chunk of lines cl, cl + FGS_LPA, cl + 2 * FGS_LPA, ...
    const int cq = lane % FGS_PPL, cl = lane / FGS_PPL;
    uint32_t rowoff[FGS_PPL];
    bool row_ok[FGS_PPL];
#pragma unroll
    for (int r = 0; r < FGS_PPL; r++) {
        row_ok[r] = l0 + cl + FGS_LPA * r < m;
        rowoff[r] = (uint32_t)min(l0 + cl + FGS_LPA * r, m - 1) * (uint32_t)n * 4u;
    }
    struct Quad { f32x4 v[FGS_PPL]; };
    auto load_tile = [&](BufRsrc r, int i0) __attribute__((always_inline)) {
        const uint32_t e = (uint32_t)min(i0 + 4 * cq, n - 4) * 4u;   // a piece past the line's end is loaded from its last piece, never used
        Quad q;
#pragma unroll
        for (int k = 0; k < FGS_PPL; k++) q.v[k] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, rowoff[k] + e, 0, 0));
        return q;
    };
    auto store_tile = [&](BufRsrc r, int i0, const Quad &q) __attribute__((always_inline)) {
        const bool piece_ok = i0 + 4 * cq < n;
        const uint32_t e = (uint32_t)(i0 + 4 * cq) * 4u;
#pragma unroll
        for (int k = 0; k < FGS_PPL; k++)
            if (piece_ok && row_ok[k]) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4w, q.v[k]), r, rowoff[k] + e, 0, 0);
    };
    // registers of the tile moves <-> LDS <-> the chunk of this thread's line
    auto put = [&](int t, const Quad &q) __attribute__((always_inline)) {
#pragma unroll
        for (int k = 0; k < FGS_PPL; k++) *reinterpret_cast<f32x4 *>(&tile[t][(cl + FGS_LPA * k) * FGS_TS + 4 * cq]) = q.v[k];
    };
    auto take = [&](int t) __attribute__((always_inline)) {
        Quad q;
#pragma unroll
        for (int k = 0; k < FGS_PPL; k++) q.v[k] = *reinterpret_cast<const f32x4 *>(&tile[t][(cl + FGS_LPA * k) * FGS_TS + 4 * cq]);
        return q;
    };
    auto get_line = [&](int t, float (&x)[FGS_RCHUNK]) __attribute__((always_inline)) {
#pragma unroll
        for (int k = 0; k < FGS_PPL; k++) {
            const f32x4 v = *reinterpret_cast<const f32x4 *>(&tile[t][lane * FGS_TS + 4 * k]);
            x[4 * k] = v.x; x[4 * k + 1] = v.y; x[4 * k + 2] = v.z; x[4 * k + 3] = v.w;
        }
    };
    auto put_line = [&](int t, const float (&x)[FGS_RCHUNK]) __attribute__((always_inline)) {
#pragma unroll
        for (int k = 0; k < FGS_PPL; k++) {
            const f32x4 v = {x[4 * k], x[4 * k + 1], x[4 * k + 2], x[4 * k + 3]};
            *reinterpret_cast<f32x4 *>(&tile[t][lane * FGS_TS + 4 * k]) = v;
        }
    };
    auto lds_wait = []() __attribute__((always_inline)) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); };

    const int nchunks = (n + FGS_RCHUNK - 1) / FGS_RCHUNK;
    float lower = 0.0f, prev_inter = 0.0f, p1 = 0.0f, p2 = 0.0f;
    float xw[FGS_RCHUNK], x1[FGS_RCHUNK], x2[FGS_RCHUNK];
    // ---- forward elimination ----
    {
        Quad qw = load_tile(rw, 0), q1 = load_tile(r1, 0), q2 = load_tile(r2, 0);
        for (int c = 0; c < nchunks; c++) {
            const int i0 = c * FGS_RCHUNK;
            put(0, qw); put(1, q1); put(2, q2);
            if (c + 1 < nchunks) { qw = load_tile(rw, i0 + FGS_RCHUNK); q1 = load_tile(r1, i0 + FGS_RCHUNK); q2 = load_tile(r2, i0 + FGS_RCHUNK); }
            lds_wait();
            get_line(0, xw); get_line(1, x1); get_line(2, x2);
            lds_wait();
            auto fwd = [&](int k) __attribute__((always_inline)) {
                const float upper = lambda * xw[k];
                const float denom = (1.0f - lower - upper) - lower * prev_inter;
                const float ci = upper / denom;
                const float v1 = (x1[k] - lower * p1) / denom;
                const float v2 = (x2[k] - lower * p2) / denom;
                if (line_ok)
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, ci), ri, ivoff,
                                                          __builtin_amdgcn_readfirstlane((uint32_t)(i0 + k) * irowb), 0);
                x1[k] = v1; x2[k] = v2;
                prev_inter = ci; p1 = v1; p2 = v2;
                lower = upper;
            };
            if (i0 + FGS_RCHUNK <= n) {   // no per-element tests outside the last, partial chunk
#pragma unroll
                for (int k = 0; k < FGS_RCHUNK; k++) fwd(k);
            } else {
#pragma unroll
                for (int k = 0; k < FGS_RCHUNK; k++)
                    if (i0 + k < n) fwd(k);
            }
            put_line(1, x1); put_line(2, x2);
            lds_wait();
            store_tile(r1, i0, take(1));
            store_tile(r2, i0, take(2));
            lds_wait();
        }
    }
    // The back substitution reads what the elimination stored, most of it through other lanes of this wave.
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    // ---- back substitution, from element n-2 down (element n-1 keeps its value) ----
    {
        auto load_inter = [&](int i0, float (&x)[FGS_RCHUNK]) __attribute__((always_inline)) {
#pragma unroll
            for (int k = 0; k < FGS_RCHUNK; k++)
                x[k] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ri, ivoff, __builtin_amdgcn_readfirstlane((uint32_t)min(i0 + k, n - 1) * irowb), 0));
        };
        float nw[FGS_RCHUNK];
        int i0 = (nchunks - 1) * FGS_RCHUNK;
        Quad q1 = load_tile(r1, i0), q2 = load_tile(r2, i0);
        load_inter(i0, xw);
        for (int c = nchunks - 1; c >= 0; c--) {
            i0 = c * FGS_RCHUNK;
            put(1, q1); put(2, q2);
            if (c > 0) { q1 = load_tile(r1, i0 - FGS_RCHUNK); q2 = load_tile(r2, i0 - FGS_RCHUNK); load_inter(i0 - FGS_RCHUNK, nw); }
            lds_wait();
            get_line(1, x1); get_line(2, x2);
            lds_wait();
            auto bwd = [&](int k) __attribute__((always_inline)) {
                p1 = x1[k] - xw[k] * p1;
                p2 = x2[k] - xw[k] * p2;
                x1[k] = p1; x2[k] = p2;
            };
            if (i0 + FGS_RCHUNK - 1 <= n - 2) {
#pragma unroll
                for (int k = FGS_RCHUNK - 1; k >= 0; k--) bwd(k);
            } else {
#pragma unroll
                for (int k = FGS_RCHUNK - 1; k >= 0; k--)
                    if (i0 + k <= n - 2) bwd(k);
            }
            put_line(1, x1); put_line(2, x2);
            lds_wait();
            store_tile(r1, i0, take(1));
            store_tile(r2, i0, take(2));
            lds_wait();
#pragma unroll
            for (int k = 0; k < FGS_RCHUNK; k++) xw[k] = nw[k];
        }
    }
}

// the columns of the output outside the sub-image (all of them when there is no sub-image): the fill value
__global__ void __launch_bounds__(256) k_wls_outside(int W, int H, int lx0, int rw, int fill, int16_t *__restrict__ out)
{
    const int no = W - rw;                                  // columns outside, per row
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y, b = blockIdx.z;
    if (i >= no) return;
    const int x = i < lx0 ? i : i + rw;
    out[((size_t)b * H + y) * W + x] = (int16_t)fill;
}

__global__ void __launch_bounds__(256) k_fill_f32(float *p, size_t n, float v)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}

} // namespace l3d

using namespace l3d;

struct l3d_wls {
    l3d_wls_params p;
    int W = 0, H = 0, device = 0, max_batch = 0, last_batch = 0;
    int lx0 = 0, rx0 = 0, rw = 0;
    hipStream_t stream = nullptr;
    float *lut = nullptr, *discL = nullptr, *discR = nullptr, *u1 = nullptr, *u2 = nullptr, *t1 = nullptr, *t2 = nullptr;
    float *cv = nullptr, *ch = nullptr, *chT = nullptr, *inter = nullptr, *conf = nullptr;
    // the horizontal pass sweeps the rows of the planes as they lie (k_fgs_sweep_rows) when the sub-image width is a
    // multiple of 4; otherwise (or with L3D_WLS_TRANSPOSE=1) it runs on transposed copies of the planes
    bool rows = false;
    int16_t *in_l = nullptr, *in_r = nullptr, *out = nullptr;
    uint8_t *in_g = nullptr;
};

extern "C" void l3d_wls_destroy(l3d_wls *h)
{
    if (!h) return;
    DeviceGuard dg(h->device);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    void *bufs[] = {h->lut, h->discL, h->discR, h->u1, h->u2, h->t1, h->t2, h->cv, h->ch, h->chT, h->inter, h->conf, h->in_l, h->in_r, h->out, h->in_g};
    for (void *q : bufs)
        if (q) (void)hipFree(q);
    if (h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
}

extern "C" int l3d_wls_create(const l3d_wls_params *p, int width, int height, int max_batch, int device, l3d_wls **out)
{
    L3D_REQUIRE(p && out, "l3d_wls_create: null argument");
    *out = nullptr;
    L3D_REQUIRE(width > 0 && height > 0 && max_batch > 0, "l3d_wls_create: bad size");
    L3D_REQUIRE(p->lambda >= 0 && p->sigma_color > 0 && p->fgs_iters >= 1, "l3d_wls_create: bad lambda / sigma_color / iterations");
    L3D_REQUIRE(p->disc_radius >= 0 && p->disc_radius <= 16, "l3d_wls_create: disc_radius %d outside [0, 16]", p->disc_radius);
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        set_error("l3d_wls_create: no HIP device available");
        return L3D_ERR_NO_DEVICE;
    }
    L3D_REQUIRE(device >= 0 && device < ndev, "l3d_wls_create: device %d out of range", device);
    // the smoother addresses an element of a frame's plane with a 32-bit byte offset
    L3D_REQUIRE((size_t)width * height * sizeof(float) < (1ull << 31), "l3d_wls_create: %dx%d frames are too large", width, height);
    l3d_wls *h = new (std::nothrow) l3d_wls();
    L3D_REQUIRE(h, "l3d_wls_create: out of host memory");
    h->p = *p; h->W = width; h->H = height; h->device = device; h->max_batch = max_batch;
    h->lx0 = std::max(0, p->min_disparity + p->num_disparities);
    h->rx0 = std::max(0, -p->min_disparity);
    h->rw = width - h->lx0 - h->rx0;
    DeviceGuard dg(device);
    const size_t B = (size_t)max_batch, px = (size_t)width * height;
    const size_t sub = h->rw > 0 ? B * height * (size_t)h->rw : 0;
    hipError_t e = hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking);
    auto alloc = [&](void **ptr, size_t bytes) {
        if (e == hipSuccess && bytes) e = hipMalloc(ptr, bytes);
    };
    alloc((void **)&h->lut, 256 * sizeof(float));
    h->rows = h->rw >= 4 && h->rw % 4 == 0;
    if (const char *ov = std::getenv("L3D_WLS_TRANSPOSE"))
        if (std::atoi(ov) != 0) h->rows = false;
    float **planes[] = {&h->discL, &h->discR, &h->u1, &h->u2, &h->cv, &h->inter};
    for (float **pl : planes) alloc((void **)pl, sub * sizeof(float));
    if (h->rows)
        alloc((void **)&h->ch, sub * sizeof(float));
    else {
        float **tplanes[] = {&h->t1, &h->t2, &h->chT};
        for (float **pl : tplanes) alloc((void **)pl, sub * sizeof(float));
    }
    alloc((void **)&h->conf, B * px * sizeof(float));
    alloc((void **)&h->in_l, B * px * 2);
    alloc((void **)&h->in_r, B * px * 2);
    alloc((void **)&h->out, B * px * 2);
    alloc((void **)&h->in_g, B * px);
    if (e == hipSuccess) {
        // edge-weight table of the smoother: -exp(-|difference| / sigma_color)
        std::vector<float> lut(256);
        for (int i = 0; i < 256; i++) lut[i] = (float)(-std::exp(-std::sqrt((double)(i * i)) / (float)p->sigma_color));
        e = hipMemcpy(h->lut, lut.data(), 256 * sizeof(float), hipMemcpyHostToDevice);
    }
    if (e == hipSuccess && wls_disc_lds(p->disc_radius) > 64 * 1024)
        e = hipFuncSetAttribute((const void *)k_wls_disc, hipFuncAttributeMaxDynamicSharedMemorySize, (int)wls_disc_lds(p->disc_radius));
    if (e != hipSuccess) {
        set_error("l3d_wls_create: %s", hipGetErrorString(e));
        l3d_wls_destroy(h);
        return L3D_ERR_HIP;
    }
    *out = h;
    return L3D_OK;
}

static int wls_run(l3d_wls *h, const int16_t *dL, const int16_t *dR, const uint8_t *guide, int B, int16_t *out, hipStream_t s)
{
    const int W = h->W, H = h->H, rw = h->rw, r = h->p.disc_radius;
    h->last_batch = B;
    const int fill = 16 * (h->p.min_disparity - 1);
    const size_t npx = (size_t)B * W * H;
    if (rw <= 0) hipLaunchKernelGGL(k_fill_f32, dim3((unsigned)((npx + 255) / 256)), dim3(256), 0, s, h->conf, npx, 0.0f);
    if (rw > 0) {
        dim3 gd(cdiv(rw, DT_W), cdiv(H, DT_H), 2 * B);
        hipLaunchKernelGGL(k_wls_disc, gd, dim3(256), wls_disc_lds(r), s, dL, dR, W, H, h->lx0, h->rx0, rw, r, h->p.disc_rolloff, h->discL, h->discR);
        // (four pixels a thread when rows split into whole groups of four and the caller's maps are aligned for it)
        if (h->rows && W % 4 == 0 && h->lx0 % 4 == 0 && (uintptr_t)dL % 8 == 0 && (uintptr_t)guide % 4 == 0 && !std::getenv("L3D_WLS_CONF1"))
            hipLaunchKernelGGL(k_wls_conf4, dim3(cdiv(W / 4, 256), H, B), dim3(256), 0, s, dL, dR, guide, h->lut, W, H, h->lx0, h->rx0, rw,
                               h->p.lrc_thresh, h->discL, h->discR, h->u1, h->u2, h->cv, h->ch, h->conf);
        else
        hipLaunchKernelGGL(k_wls_conf, dim3(cdiv(W, 256), H, B), dim3(256), 0, s, dL, dR, guide, h->lut, W, H, h->lx0, h->rx0, rw,
                           h->p.lrc_thresh, h->discL, h->discR, h->u1, h->u2, h->cv, h->ch, h->chT, h->conf);
        float lambda = (float)h->p.lambda;
        const dim3 tg_fwd(cdiv(rw, 32), cdiv(H, 32), B), tg_bwd(cdiv(H, 32), cdiv(rw, 32), B);
        for (int it = 0; it < h->p.fgs_iters; it++) {
            // horizontal pass: H lines of length rw, on the planes as they lie or on transposed copies
            if (h->rows)
                hipLaunchKernelGGL(k_fgs_sweep_rows, dim3(cdiv(H, 64), B), dim3(64), 0, s, h->u1, h->u2, h->ch, h->inter, rw, H, lambda);
            else {
                hipLaunchKernelGGL(k_transpose2, tg_fwd, dim3(256), 0, s, h->u1, h->u2, H, rw, h->t1, h->t2);
                hipLaunchKernelGGL(k_fgs_sweep<false>, dim3(cdiv(H, 64), B), dim3(64), 0, s, h->t1, h->t2, h->chT, h->inter, rw, H, lambda, nullptr, 0, 0);
                hipLaunchKernelGGL(k_transpose2, tg_bwd, dim3(256), 0, s, h->t1, h->t2, rw, H, h->u1, h->u2);
            }
            // vertical pass: rw lines of length H; the last one writes the output
            if (it + 1 < h->p.fgs_iters)
                hipLaunchKernelGGL(k_fgs_sweep<false>, dim3(cdiv(rw, 64), B), dim3(64), 0, s, h->u1, h->u2, h->cv, h->inter, H, rw, lambda, nullptr, 0, 0);
            else
                hipLaunchKernelGGL(k_fgs_sweep<true>, dim3(cdiv(rw, 64), B), dim3(64), 0, s, h->u1, h->u2, h->cv, h->inter, H, rw, lambda, out, W, h->lx0);
            lambda *= h->p.fgs_attenuation;
        }
    }
    if (W - std::max(rw, 0) > 0)
        hipLaunchKernelGGL(k_wls_outside, dim3(cdiv(W - std::max(rw, 0), 256), H, B), dim3(256), 0, s, W, H, h->lx0, std::max(rw, 0), fill, out);
    L3D_HIP_CHECK(hipGetLastError());
    return L3D_OK;
}

extern "C" int l3d_wls_filter_device(l3d_wls *h, const int16_t *d_disp_left, const int16_t *d_disp_right, const uint8_t *d_guide, int batch,
                                     int16_t *d_out, void *stream)
{
    L3D_REQUIRE(h && d_disp_left && d_disp_right && d_guide && d_out, "l3d_wls_filter_device: null argument");
    L3D_REQUIRE(batch >= 1 && batch <= h->max_batch, "l3d_wls_filter_device: batch %d outside [1, %d]", batch, h->max_batch);
    DeviceGuard dg(h->device);
    return wls_run(h, d_disp_left, d_disp_right, d_guide, batch, d_out, (hipStream_t)stream);
}

extern "C" int l3d_wls_filter(l3d_wls *h, const int16_t *disp_left, const int16_t *disp_right, const uint8_t *guide, int batch, int16_t *out)
{
    L3D_REQUIRE(h && disp_left && disp_right && guide && out, "l3d_wls_filter: null argument");
    L3D_REQUIRE(batch >= 1 && batch <= h->max_batch, "l3d_wls_filter: batch %d outside [1, %d]", batch, h->max_batch);
    DeviceGuard dg(h->device);
    hipStream_t s = h->stream;
    const size_t npx = (size_t)batch * h->W * h->H;
    L3D_HIP_CHECK(hipMemcpyAsync(h->in_l, disp_left, npx * 2, hipMemcpyHostToDevice, s));
    L3D_HIP_CHECK(hipMemcpyAsync(h->in_r, disp_right, npx * 2, hipMemcpyHostToDevice, s));
    L3D_HIP_CHECK(hipMemcpyAsync(h->in_g, guide, npx, hipMemcpyHostToDevice, s));
    int rc = wls_run(h, h->in_l, h->in_r, h->in_g, batch, h->out, s);
    if (rc) return rc;
    L3D_HIP_CHECK(hipMemcpyAsync(out, h->out, npx * 2, hipMemcpyDeviceToHost, s));
    L3D_HIP_CHECK(hipStreamSynchronize(s));
    return L3D_OK;
}

extern "C" int l3d_wls_fetch_confidence(l3d_wls *h, float *host_dst, size_t bytes)
{
    L3D_REQUIRE(h && host_dst, "l3d_wls_fetch_confidence: null argument");
    const size_t avail = (size_t)h->last_batch * h->W * h->H * sizeof(float);
    L3D_REQUIRE(bytes <= avail, "l3d_wls_fetch_confidence: %zu bytes requested, %zu available", bytes, avail);
    DeviceGuard dg(h->device);
    L3D_HIP_CHECK(hipDeviceSynchronize());
    L3D_HIP_CHECK(hipMemcpy(host_dst, h->conf, bytes, hipMemcpyDeviceToHost));
    return L3D_OK;
}
