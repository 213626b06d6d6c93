// Semi-global matching with 5 and 8 aggregation paths (cv2.STEREO_SGBM_MODE_SGBM and MODE_HH) for gfx950.
//
// The reference builds its matchers with mode = cv2.STEREO_SGBM_MODE_SGBM_3WAY
// (camera/single_usb_stereo_camera.py:273), which sgbm.hip serves; these are the other two modes of the same
// cv2.StereoSGBM_create call, for callers that ask for the full set of paths.
//
// Every path is a one-dimensional recurrence along a line of the image, and the summed cost takes the paths by
// saturating additions in a fixed order; so each path is one pass over the aggregated-cost volume that adds its costs
// to the sum volume, and the passes run one after the other on a stream:
//   C [B][H][W1][Dp] s16   aggregated cost, whole image (written by the vertical kernel of sgbm.hip run with a single stripe)
//   S [B][H][W1][Dp] s16   sum of the path costs so far
// One lane group (LPP lanes, 2 * NP disparities a lane, as in sgbm.hip) walks one line: a row, a column or a diagonal;
// a wave carries 64 / LPP neighbouring lines, whose pixels at a step are neighbours in memory (columns, diagonals) or
// rows apart (rows).  The operands of a step are requested one step ahead of their use.
// A last kernel finds winner, uniqueness and sub-pixel disparity of every column from S; the left-right check and
// the filters after it are the ones of the 3-path mode.
#include "sgbm_dev.h"
#include <algorithm>

namespace l3d {

// (dx, dy) = the step from a pixel to the next one on the path; `overwrite`: this is the first path, S = L instead of
// S += L.  Lines per frame: H rows, W1 columns, or W1 + H - 1 diagonals (those entering through the row where the pass
// starts, then those entering through the side column).
template <int NP, int LPP, bool PAD>
__global__ void __launch_bounds__(64) k_path(SgbmGeom g, int B, int dx, int dy, int overwrite, const uint32_t *__restrict__ Cvol,
                                             uint32_t *__restrict__ Svol)
{
    constexpr int LPW = 64 / LPP; // lines per wave
    constexpr int CW = NP * LPP;  // u32 per volume column
    const int tid = threadIdx.x, grp = tid / LPP, l = tid & (LPP - 1);
    const bool first = l == 0, last = l == LPP - 1;
    const int W1 = g.W1, H = g.H;
    const int nlines = dy == 0 ? H : (dx == 0 ? W1 : W1 + H - 1);
    const long long line = (long long)blockIdx.x * LPW + grp;   // over B * nlines
    const bool on = line < (long long)B * nlines;
    const int b = on ? (int)(line / nlines) : 0;
    const int k = on ? (int)(line - (long long)b * nlines) : 0;
    int x0, y0, len;
    if (dy == 0) {
        y0 = k; x0 = dx > 0 ? 0 : W1 - 1; len = W1;
    } else if (dx == 0) {
        x0 = k; y0 = dy > 0 ? 0 : H - 1; len = H;
    } else {
        const int ys = dy > 0 ? 0 : H - 1, xs = dx > 0 ? 0 : W1 - 1;
        if (k < W1) { x0 = k; y0 = ys; }
        else { x0 = xs; y0 = ys + dy * (k - W1 + 1); }
        len = min(dx > 0 ? W1 - x0 : x0 + 1, dy > 0 ? H - y0 : y0 + 1);
    }
    if (!on) len = 0;
    int maxlen = len;   // the longest line of the wave
#pragma unroll
    for (int o = LPP; o < 64; o <<= 1) maxlen = max(maxlen, __shfl_xor(maxlen, o));
    if (maxlen == 0) return;

    const SgmPenalty pen = make_penalty(g);
    const uint32_t p2v = pk_splat(g.P2);
    PadFlags<NP, PAD> pad;
    pad.init(l, g.D / 2);
    uint32_t st[NP];
#pragma unroll
    for (int p = 0; p < NP; p++) st[p] = pad.initial(p);
    uint32_t mn = 0;
    // element index (u32 words) of this lane's part of the current pixel, and its change per step
    long long idx = (((long long)b * H + y0) * W1 + x0) * CW + l * NP;
    const long long step = ((long long)dy * W1 + dx) * CW;
    auto ld = [&](const uint32_t *vol, long long i) __attribute__((always_inline)) { return *reinterpret_cast<const Pack<NP> *>(vol + i); };
    Pack<NP> c = ld(Cvol, idx), s = ld(Svol, idx), cn = c, sn = s;
    for (int t = 0; t < maxlen; t++) {
        const bool act = t < len;
        const long long nidx = (t + 1 < len) ? idx + step : idx;   // lines that have ended stay on their last pixel
        cn = ld(Cvol, nidx);
        if (!overwrite) sn = ld(Svol, nidx);
        uint32_t cp[NP];
#pragma unroll
        for (int p = 0; p < NP; p++) cp[p] = pk_add(c.v[p], p2v);   // the cost is carried with P2 added (16-bit wrap)
        mn = sgm_step<NP, LPP, PAD>(cp, st, mn, pen, first, last, pad);
        if (act) {
            Pack<NP> o;
#pragma unroll
            // One saturating addition per path, in the order the paths run, equals the definition's single saturation
            // of the whole sum: the cost is carried with P2 added, so every path cost st is >= 0 (it is at least
            // c + P2 - (previous minimum + P2) + previous minimum >= 0 from costs c >= 0) and the partial sums only
            // grow -- once one of them reaches 32767 it stays there, exactly where the full sum would be clamped.
            for (int p = 0; p < NP; p++) o.v[p] = overwrite ? st[p] : pk_adds(s.v[p], st[p]);
            *reinterpret_cast<Pack<NP> *>(Svol + idx) = o;
        }
        c = cn; s = sn;
        idx = nidx;
    }
}

// Winner, uniqueness test and sub-pixel disparity of every column from the summed costs: one lane group per column,
// then one lane per column for the parabola fit (the summed costs pass through LDS for their best-1 / best+1 entries).
template <int NP, int LPP, bool PAD>
__global__ void __launch_bounds__(64) k_wta_s(SgbmGeom g, long long ncols, const uint32_t *__restrict__ Svol, uint32_t *__restrict__ wta,
                                              int16_t *__restrict__ dsub)
{
    constexpr int PPW = 64 / LPP, CW = NP * LPP, D = 2 * CW, REC = D + 8;
    __shared__ __align__(16) int16_t sS[8 + PPW * REC];
    __shared__ uint32_t sK[PPW];
    __shared__ __align__(16) uint32_t sWin[(2 * NP + 4) * NP];
    const int tid = threadIdx.x, grp = tid / LPP, l = tid & (LPP - 1);
    uniq_window_fill<NP>(sWin, tid, 64);
    __syncthreads();
    const int d0 = l * 2 * NP;
    for (long long base = (long long)blockIdx.x * PPW; base < ncols; base += (long long)gridDim.x * PPW) {
        const long long col = min(base + grp, ncols - 1);
        const Pack<NP> S = *reinterpret_cast<const Pack<NP> *>(Svol + col * CW + l * NP);
        const uint32_t key = wta_key<NP, LPP>(g, S.v, d0, sWin);
        *reinterpret_cast<PackAny<NP> *>(sS + 8 + grp * REC + d0) = S;
        if (l == 0) sK[grp] = key;
        __syncthreads();
        if (tid < PPW && base + tid < ncols) {
            const uint32_t kk = sK[tid];
            int dval = 0;
            if (kk != 0xffffffffu) {
                const int best = (int)(kk & 0xffffu);
                const int16_t *rec = sS + 8 + tid * REC + best;
                dval = subpixel_disp(g, best, rec[-1], (int)(kk >> 16) - 32768, rec[1]);
            }
            wta[base + tid] = kk;
            dsub[base + tid] = (int16_t)dval;
        }
        __syncthreads();
    }
}

} // namespace l3d

using namespace l3d;

// mode 0: the paths from the left, top-left, top, top-right and the right; mode 1: all eight.  Leaves (key, sub-pixel
// disparity) per column in wta / dsub, like the horizontal kernel of the 3-path mode.
int sgbm_paths_aggregate(const SgbmGeom &g, int mode, int B, const uint32_t *Cvol, uint32_t *Svol, uint32_t *wta, int16_t *dsub, hipStream_t s)
{
    // in the order in which the definition adds them to the sum
    static const int dirs[8][2] = {{1, 0}, {1, 1}, {0, 1}, {-1, 1}, {-1, 0}, {1, -1}, {0, -1}, {-1, -1}};
    const int ndirs = mode == 1 ? 8 : 5;
    const long long ncols = (long long)B * g.H * g.W1;
    const bool ok = dispatch_variant(g.NPh, g.LPPh, g.Dp != g.D, [&](auto np, auto lpp, auto pd) {
        constexpr int NP = decltype(np)::value, LPP = decltype(lpp)::value;
        constexpr bool PAD = decltype(pd)::value;
        for (int i = 0; i < ndirs; i++) {
            const int dx = dirs[i][0], dy = dirs[i][1];
            const long long nlines = (long long)B * (dy == 0 ? g.H : (dx == 0 ? g.W1 : g.W1 + g.H - 1));
            hipLaunchKernelGGL((k_path<NP, LPP, PAD>), dim3((unsigned)((nlines + 64 / LPP - 1) / (64 / LPP))), dim3(64), 0, s, g, B, dx, dy,
                               i == 0 ? 1 : 0, Cvol, Svol);
        }
        const long long nblk = std::min<long long>((ncols + 64 / LPP - 1) / (64 / LPP), 1 << 20);
        hipLaunchKernelGGL((k_wta_s<NP, LPP, PAD>), dim3((unsigned)nblk), dim3(64), 0, s, g, ncols, Svol, wta, dsub);
    });
    if (!ok) {
        set_error("sgbm: no kernel variant for %d disparities", g.D);
        return L3D_ERR_UNSUPPORTED;
    }
    return L3D_OK;
}
