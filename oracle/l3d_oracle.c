/*
 * l3d_oracle.c -- CPU restatement of the per-frame stereo-depth + laser
 * centre-line path.  TEST INFRASTRUCTURE ONLY: this file is the checker that
 * the HIP kernels are compared against (tests/, the smoke check and the
 * cpu_baseline leg of the benchmark).  Nothing in the product package may
 * import, link or call it.
 *
 * PARITY STATUS: "parity unpinned" against OpenCV.  The reference repository
 * calls cv2.StereoSGBM (MODE_SGBM_3WAY), cv2.ximgproc.DisparityWLSFilter,
 * cv2.GaussianBlur / filter2D, cv2.cvtColor, cv2.reprojectImageTo3D
 * (camera/single_usb_stereo_camera.py:252-346, core/laser_extractor.py:56-82,
 * 174-256).  OpenCV (opencv-python >= 4.5.0, unpinned upper bound; the
 * ximgproc module additionally needs opencv-contrib-python, undeclared) is not
 * present in /root/reference or on this machine, and the reference ships no
 * golden vectors for these stages.  The algorithms below restate OpenCV 4.x's
 * published behaviour (modules/calib3d/src/stereosgbm.cpp,
 * modules/ximgproc/src/{disparity_filters,fgs_filter}.cpp,
 * modules/imgproc/src/{color_rgb,color_hsv}.simd.hpp, smooth.dispatch.cpp,
 * filter.dispatch.cpp, median_blur, modules/calib3d/src/calibration.cpp) and
 * are anchored on the reference's call sites and parameters.  The 3-D
 * reconstruction functions (core/reconstruction.py:30-182) ARE pinned: golden
 * vectors generated from the reference itself live in tests/golden/.
 *
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off -shared).
 */
#include <stdint.h>
#include <stddef.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <limits.h>
#include <float.h>

#define ORC_API __attribute__((visibility("default")))

/* ------------------------------------------------------------------------ */
/* small helpers                                                             */
/* ------------------------------------------------------------------------ */
static inline int imin(int a, int b) { return a < b ? a : b; }
static inline int imax(int a, int b) { return a > b ? a : b; }
static inline int sat16(int v) { return v > 32767 ? 32767 : (v < -32768 ? -32768 : v); }
static inline int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

/* BORDER_REFLECT_101 index (gfedcb|abcdefgh|gfedcba) */
static inline int reflect101(int p, int n)
{
    if (n == 1) return 0;
    while (p < 0 || p >= n) {
        if (p < 0) p = -p;
        else p = 2 * (n - 1) - p;
    }
    return p;
}

/* round half to even, as cvRound / SSE cvtsd2si do */
static inline int round_half_even(double v)
{
    return (int)nearbyint(v); /* default FE_TONEAREST */
}

/* ------------------------------------------------------------------------ */
/* G1  cv2.cvtColor(BGR2GRAY), 8-bit                                          */
/* call sites: camera/single_usb_stereo_camera.py:320-321,                    */
/*             core/laser_extractor.py:60,174                                 */
/* OpenCV 4.x 8-bit path: 15-bit fixed point,                                 */
/*   Y = (R*9798 + G*19235 + B*3735 + (1<<14)) >> 15                          */
/* ------------------------------------------------------------------------ */
ORC_API void orc_bgr2gray(const uint8_t *bgr, int w, int h, ptrdiff_t row_stride, uint8_t *gray)
{
    for (int y = 0; y < h; y++) {
        const uint8_t *s = bgr + (ptrdiff_t)y * row_stride;
        uint8_t *d = gray + (ptrdiff_t)y * w;
        for (int x = 0; x < w; x++) {
            int b = s[3 * x], g = s[3 * x + 1], r = s[3 * x + 2];
            d[x] = (uint8_t)((r * 9798 + g * 19235 + b * 3735 + (1 << 14)) >> 15);
        }
    }
}

/* ------------------------------------------------------------------------ */
/* H1  cv2.cvtColor(BGR2HSV), 8-bit, H in [0,180)                             */
/* call site: core/laser_extractor.py:56                                      */
/* OpenCV RGB2HSV_b: 12-bit fixed-point division tables.                      */
/* ------------------------------------------------------------------------ */
static int g_sdiv[256], g_hdiv180[256], g_hsv_tab_ready = 0;
static void hsv_tables(void)
{
    if (g_hsv_tab_ready) return;
    g_sdiv[0] = g_hdiv180[0] = 0;
    for (int i = 1; i < 256; i++) {
        g_sdiv[i] = round_half_even((255 << 12) / (1. * i));
        g_hdiv180[i] = round_half_even((180 << 12) / (6. * i));
    }
    g_hsv_tab_ready = 1;
}

static inline void bgr2hsv_px(int b, int g, int r, int *ph, int *ps, int *pv)
{
    int v = imax(b, imax(g, r));
    int vmin = imin(b, imin(g, r));
    int diff = v - vmin;
    int vr = (v == r) ? -1 : 0;
    int vg = (v == g) ? -1 : 0;
    int s = (diff * g_sdiv[v] + (1 << 11)) >> 12;
    int hh = (vr & (g - b)) + (~vr & ((vg & (b - r + 2 * diff)) + ((~vg) & (r - g + 4 * diff))));
    hh = (hh * g_hdiv180[diff] + (1 << 11)) >> 12;
    if (hh < 0) hh += 180;
    *ph = clampi(hh, 0, 255);
    *ps = s & 255;
    *pv = v;
}

ORC_API void orc_bgr2hsv(const uint8_t *bgr, int w, int h, ptrdiff_t row_stride, uint8_t *hsv)
{
    hsv_tables();
    for (int y = 0; y < h; y++) {
        const uint8_t *s = bgr + (ptrdiff_t)y * row_stride;
        uint8_t *d = hsv + (ptrdiff_t)y * w * 3;
        for (int x = 0; x < w; x++) {
            int hh, ss, vv;
            bgr2hsv_px(s[3 * x], s[3 * x + 1], s[3 * x + 2], &hh, &ss, &vv);
            d[3 * x] = (uint8_t)hh;
            d[3 * x + 1] = (uint8_t)ss;
            d[3 * x + 2] = (uint8_t)vv;
        }
    }
}

/* ------------------------------------------------------------------------ */
/* G0, G2..G8  cv2.StereoSGBM.compute, mode = STEREO_SGBM_MODE_SGBM_3WAY      */
/* call sites: camera/single_usb_stereo_camera.py:252-274 (parameters),       */
/*             :324 (left), :277,:325 (right matcher)                         */
/* ------------------------------------------------------------------------ */
typedef struct {
    int minDisparity;
    int numDisparities;
    int blockSize;
    int P1, P2;
    int disp12MaxDiff;
    int preFilterCap;
    int uniquenessRatio;
    int speckleWindowSize;
    int speckleRange;
    int mode; /* 2 = MODE_SGBM_3WAY (what the reference uses); 0 = MODE_SGBM (5 paths), 1 = MODE_HH (8 paths) */
} orc_sgbm_params;

typedef struct {
    int W, H, minD, maxD, D, SW2, SH2, P1, P2, uniq, disp12MaxDiff, ftzero;
    int minX1, maxX1, W1, invalid16;
    int nstripes, stripe_sz, stripe_overlap;
} sgbm_geom;

static int sgbm_geometry(const orc_sgbm_params *p, int W, int H, sgbm_geom *g)
{
    memset(g, 0, sizeof(*g));
    g->W = W; g->H = H;
    g->minD = p->minDisparity;
    g->D = p->numDisparities;
    g->maxD = g->minD + g->D;
    g->SW2 = g->SH2 = p->blockSize > 0 ? p->blockSize / 2 : 1;
    g->P1 = p->P1 > 0 ? p->P1 : 2;
    g->P2 = imax(p->P2 > 0 ? p->P2 : 5, g->P1 + 1);
    g->uniq = p->uniquenessRatio >= 0 ? p->uniquenessRatio : 10;
    g->disp12MaxDiff = p->disp12MaxDiff > 0 ? p->disp12MaxDiff : 1;
    g->ftzero = imax(p->preFilterCap, 15) | 1;
    g->minX1 = imax(g->maxD, 0);
    g->maxX1 = W + imin(g->minD, 0);
    g->W1 = g->maxX1 - g->minX1;
    g->invalid16 = (g->minD - 1) * 16;
    g->nstripes = 4;
    g->stripe_sz = (H + 3) / 4; /* ceil(H/4.) */
    g->stripe_overlap = (g->SH2 + 1) + (int)ceil(0.1 * g->stripe_sz);
    if (g->D <= 0 || g->D % 16 != 0) return -1;
    if (W < 3 || H < 1) return -2;
    return 0;
}

/* Per-image-row prefilter: x-Sobel clipped to [-ftzero, ftzero] + ftzero
 * ("gradient channel") and the raw intensity ("intensity channel"); columns 0
 * and W-1 of both channels hold ftzero. */
static void prefilter_row(const uint8_t *img, int W, int H, int y, int ftzero, uint8_t *grad, uint8_t *raw)
{
    const uint8_t *r0 = img + (size_t)y * W;
    const uint8_t *rn = y > 0 ? r0 - W : r0;
    const uint8_t *rs = y < H - 1 ? r0 + W : r0;
    grad[0] = grad[W - 1] = (uint8_t)ftzero;
    raw[0] = raw[W - 1] = (uint8_t)ftzero;
    for (int x = 1; x < W - 1; x++) {
        int s = (r0[x + 1] - r0[x - 1]) * 2 + rn[x + 1] - rn[x - 1] + rs[x + 1] - rs[x - 1];
        grad[x] = (uint8_t)(clampi(s, -ftzero, ftzero) + ftzero);
        raw[x] = r0[x];
    }
}

/* half-pixel interpolated min / max of a prefiltered row (Birchfield-Tomasi) */
static void bt_minmax_row(const uint8_t *v, int W, uint8_t *v0, uint8_t *v1)
{
    for (int x = 0; x < W; x++) {
        int c = v[x];
        int l = x > 0 ? (c + v[x - 1]) / 2 : c;
        int r = x < W - 1 ? (c + v[x + 1]) / 2 : c;
        v0[x] = (uint8_t)imin(c, imin(l, r));
        v1[x] = (uint8_t)imax(c, imax(l, r));
    }
}

typedef struct {
    uint8_t *g1, *g1lo, *g1hi, *r1, *r1lo, *r1hi; /* left: gradient/raw + BT min/max */
    uint8_t *g2, *g2lo, *g2hi, *r2, *r2lo, *r2hi; /* right */
} bt_planes;

static void bt_planes_alloc(bt_planes *b, int W, int H)
{
    size_t n = (size_t)W * H;
    uint8_t *m = (uint8_t *)malloc(12 * n);
    b->g1 = m; b->g1lo = m + n; b->g1hi = m + 2 * n; b->r1 = m + 3 * n; b->r1lo = m + 4 * n; b->r1hi = m + 5 * n;
    b->g2 = m + 6 * n; b->g2lo = m + 7 * n; b->g2hi = m + 8 * n; b->r2 = m + 9 * n; b->r2lo = m + 10 * n; b->r2hi = m + 11 * n;
}

static void bt_planes_fill(bt_planes *b, const uint8_t *left, const uint8_t *right, int W, int H, int ftzero)
{
    for (int y = 0; y < H; y++) {
        size_t o = (size_t)y * W;
        prefilter_row(left, W, H, y, ftzero, b->g1 + o, b->r1 + o);
        prefilter_row(right, W, H, y, ftzero, b->g2 + o, b->r2 + o);
        bt_minmax_row(b->g1 + o, W, b->g1lo + o, b->g1hi + o);
        bt_minmax_row(b->r1 + o, W, b->r1lo + o, b->r1hi + o);
        bt_minmax_row(b->g2 + o, W, b->g2lo + o, b->g2hi + o);
        bt_minmax_row(b->r2 + o, W, b->r2lo + o, b->r2hi + o);
    }
}

/* Exported for stage-level GPU checks: 12 planes of W*H bytes in the order of
 * the bt_planes struct. */
ORC_API int orc_sgbm_prefilter(const orc_sgbm_params *p, const uint8_t *left, const uint8_t *right,
                               int W, int H, uint8_t *planes12)
{
    sgbm_geom g;
    if (sgbm_geometry(p, W, H, &g)) return -1;
    bt_planes b;
    bt_planes_alloc(&b, W, H);
    bt_planes_fill(&b, left, right, W, H, g.ftzero);
    memcpy(planes12, b.g1, (size_t)12 * W * H);
    free(b.g1);
    return 0;
}

static inline int bt_cost1(int u, int u0, int u1, int v, int v0, int v1)
{
    int c0 = imax(0, imax(u - v1, v0 - u));
    int c1 = imax(0, imax(v - u1, u0 - v));
    return imin(c0, c1);
}

/* pixel cost row k: pix[xc*D + di], xc in [0,W1), disparity d = di + minD,
 * left column x = xc + minX1, right column x - d. */
static void pixcost_row(const sgbm_geom *g, const bt_planes *b, int k, int16_t *pix)
{
    const int W = g->W, D = g->D;
    size_t o = (size_t)k * W;
    for (int xc = 0; xc < g->W1; xc++) {
        int x = xc + g->minX1;
        int u = b->g1[o + x], u0 = b->g1lo[o + x], u1 = b->g1hi[o + x];
        int t = b->r1[o + x], t0 = b->r1lo[o + x], t1 = b->r1hi[o + x];
        int16_t *dst = pix + (size_t)xc * D;
        for (int di = 0; di < D; di++) {
            int xr = x - (di + g->minD);
            int cg = bt_cost1(u, u0, u1, b->g2[o + xr], b->g2lo[o + xr], b->g2hi[o + xr]);
            int ci = bt_cost1(t, t0, t1, b->r2[o + xr], b->r2lo[o + xr], b->r2hi[o + xr]);
            dst[di] = (int16_t)(cg + (ci >> 2));
        }
    }
}

/* horizontal box sum of a pixel-cost row; columns replicate-clamped to the
 * cost range [0, W1-1] */
static void hsum_row(const sgbm_geom *g, const int16_t *pix, int16_t *hs)
{
    const int D = g->D, W1 = g->W1, SW2 = g->SW2;
    for (int di = 0; di < D; di++) {
        int s = pix[di] * (SW2 + 1);
        for (int j = 1; j <= SW2; j++) s += pix[(size_t)imin(j, W1 - 1) * D + di];
        hs[di] = (int16_t)s;
    }
    for (int xc = 1; xc < W1; xc++) {
        const int16_t *add = pix + (size_t)imin(xc + SW2, W1 - 1) * D;
        const int16_t *sub = pix + (size_t)imax(xc - SW2 - 1, 0) * D;
        int16_t *cur = hs + (size_t)xc * D;
        const int16_t *prv = cur - D;
        for (int di = 0; di < D; di++) cur[di] = (int16_t)(prv[di] + add[di] - sub[di]);
    }
}

/* One step of the SGM recurrence with 16-bit saturating arithmetic:
 *   out[d] = C[d] + min(prev[d], prev[d-1]+P1, prev[d+1]+P1, prevMin+P2) - (prevMin+P2)
 * out may alias prev. Returns min_d out[d]. */
static int sgm_step(const int16_t *C, const int16_t *prev, int prevMin, int D, int P1, int P2, int16_t *out)
{
    int16_t tmp[1024];
    memcpy(tmp, prev, (size_t)D * sizeof(int16_t));
    int minP2 = sat16(prevMin + P2);
    int P1s = sat16(P1);
    int newMin = 32767;
    for (int d = 0; d < D; d++) {
        int l = d > 0 ? tmp[d - 1] : 32767;
        int r = d < D - 1 ? tmp[d + 1] : 32767;
        int a = sat16(imin(l, r) + P1s);
        int m = imin(a, imin(tmp[d], minP2));
        int rr = sat16(m - minP2);
        int o = sat16(C[d] + rr);
        out[d] = (int16_t)o;
        newMin = imin(newMin, o);
    }
    return newMin;
}

/* Stage outputs a caller may request (any pointer may be NULL):
 *   Cvol, Tvol, Svol : H * W1 * D int16, rows owned by their stripe
 *   raw              : H * W int16, disparity before median / speckle     */
typedef struct {
    int16_t *Cvol, *Tvol, *Svol, *raw, *median;
} sgbm_dumps;

static int sgbm_3way(const orc_sgbm_params *p, const uint8_t *left, const uint8_t *right,
                     int W, int H, int16_t *disp, const sgbm_dumps *dump)
{
    sgbm_geom g;
    int rc = sgbm_geometry(p, W, H, &g);
    if (rc) return rc;
    const int D = g.D, W1 = g.W1;
    if (D > 1024) return -3;
    for (size_t i = 0; i < (size_t)W * H; i++) disp[i] = (int16_t)g.invalid16;
    if (W1 <= 0) return 0;
    if (g.stripe_overlap > g.stripe_sz) return -4; /* image too small for the 4-stripe scheme */

    bt_planes bt;
    bt_planes_alloc(&bt, W, H);
    bt_planes_fill(&bt, left, right, W, H, g.ftzero);

    const size_t rowN = (size_t)W1 * D;
    const int NR = 2 * g.SH2 + 2;
    int16_t *pix = (int16_t *)malloc(rowN * sizeof(int16_t));
    int16_t *ring = (int16_t *)malloc(rowN * NR * sizeof(int16_t));
    int *ring_row = (int *)malloc(NR * sizeof(int));
    int16_t *C = (int16_t *)malloc(rowN * sizeof(int16_t));
    int16_t *T = (int16_t *)malloc(rowN * sizeof(int16_t));
    int16_t *Lh = (int16_t *)malloc(rowN * sizeof(int16_t));
    int16_t *topMin = (int16_t *)malloc(W1 * sizeof(int16_t));
    int16_t *Rb = (int16_t *)malloc(D * sizeof(int16_t));
    int16_t *zero = (int16_t *)calloc(D, sizeof(int16_t));
    int16_t *disp2 = (int16_t *)malloc(W * sizeof(int16_t));
    int16_t *disp2cost = (int16_t *)malloc(W * sizeof(int16_t));

    for (int n = 0; n < g.nstripes; n++) {
        int start = imax(imin(n * g.stripe_sz - g.stripe_overlap, H), 0);
        int end = imin((n + 1) * g.stripe_sz, H);
        int own0 = n * g.stripe_sz; /* first row this stripe contributes to the output */
        if (start >= end) continue;
        for (int i = 0; i < NR; i++) ring_row[i] = -1;
        memset(C, 0, rowN * sizeof(int16_t));
        memset(T, 0, rowN * sizeof(int16_t));
        memset(topMin, 0, W1 * sizeof(int16_t));

        for (int y = start; y < end; y++) {
            /* ---- G2/G3: block-aggregated BT cost for row y of this stripe ---- */
            int k0 = (y == start) ? start : y + g.SH2, k1 = y + g.SH2;
            for (int k = k0; k <= k1; k++) {
                int kk = imin(k, H - 1);
                int16_t *hs = ring + (size_t)(kk % NR) * rowN;
                if (ring_row[kk % NR] != kk) {
                    pixcost_row(&g, &bt, kk, pix);
                    hsum_row(&g, pix, hs);
                    ring_row[kk % NR] = kk;
                }
                if (y == start) {
                    int scale = (k == start) ? g.SH2 + 1 : 1;
                    for (size_t i = 0; i < rowN; i++) C[i] = (int16_t)(C[i] + hs[i] * scale);
                } else {
                    int ks = imax(y - g.SH2 - 1, start);
                    const int16_t *sub = ring + (size_t)(ks % NR) * rowN;
                    for (size_t i = 0; i < rowN; i++) C[i] = (int16_t)(C[i] + hs[i] - sub[i]);
                }
            }
            int owned = (y >= own0);
            if (owned && dump && dump->Cvol) memcpy(dump->Cvol + (size_t)y * rowN, C, rowN * sizeof(int16_t));

            /* ---- G4: left->right and top->down paths ---- */
            int leftMin = 0;
            for (int xc = 0; xc < W1; xc++) {
                const int16_t *Cx = C + (size_t)xc * D;
                const int16_t *prev = xc > 0 ? Lh + (size_t)(xc - 1) * D : zero;
                leftMin = sgm_step(Cx, prev, leftMin, D, g.P1, g.P2, Lh + (size_t)xc * D);
                topMin[xc] = (int16_t)sgm_step(Cx, T + (size_t)xc * D, topMin[xc], D, g.P1, g.P2, T + (size_t)xc * D);
            }
            if (owned && dump && dump->Tvol) memcpy(dump->Tvol + (size_t)y * rowN, T, rowN * sizeof(int16_t));
            if (!owned) continue; /* rows in the overlap only carry the top->down state */

            int16_t *drow = disp + (size_t)y * W;
            for (int x = 0; x < W; x++) { disp2[x] = (int16_t)g.invalid16; disp2cost[x] = 32767; }

            /* ---- G4 right->left path, G5 WTA / uniqueness / sub-pixel ---- */
            memset(Rb, 0, D * sizeof(int16_t));
            int rightMin = 0;
            for (int xc = W1 - 1; xc >= 0; xc--) {
                const int16_t *Cx = C + (size_t)xc * D;
                int16_t *Sx = Lh + (size_t)xc * D;
                rightMin = sgm_step(Cx, Rb, rightMin, D, g.P1, g.P2, Rb);
                int best = 0, minS = 32767;
                for (int d = 0; d < D; d++) {
                    int s = sat16(sat16(Rb[d] + Sx[d]) + T[(size_t)xc * D + d]);
                    Sx[d] = (int16_t)s;
                    if (s < minS) { minS = s; best = d; }
                }
                if (dump && dump->Svol) memcpy(dump->Svol + (size_t)y * rowN + (size_t)xc * D, Sx, D * sizeof(int16_t));
                if (g.uniq > 0) {
                    int d;
                    for (d = 0; d < D; d++)
                        if (Sx[d] * (100 - g.uniq) < minS * 100 && abs(d - best) > 1) break;
                    if (d < D) continue;
                }
                int x2 = xc + g.minX1 - best - g.minD;
                if (x2 >= 0 && x2 < W && disp2cost[x2] > minS) {
                    disp2cost[x2] = (int16_t)minS;
                    disp2[x2] = (int16_t)(best + g.minD);
                }
                int dsub;
                if (0 < best && best < D - 1) {
                    int denom2 = imax(Sx[best - 1] + Sx[best + 1] - 2 * Sx[best], 1);
                    dsub = best * 16 + ((Sx[best - 1] - Sx[best + 1]) * 16 + denom2) / (denom2 * 2);
                } else
                    dsub = best * 16;
                drow[xc + g.minX1] = (int16_t)(dsub + g.minD * 16);
            }

            /* ---- G6: left-right consistency on the single disparity row ---- */
            for (int x = g.minX1; x < g.maxX1; x++) {
                int d1 = drow[x];
                if (d1 == g.invalid16) continue;
                int _d = d1 >> 4;
                int d_ = (d1 + 15) >> 4;
                int _x = x - _d, x_ = x - d_;
                if (0 <= _x && _x < W && disp2[_x] >= g.minD && abs(disp2[_x] - _d) > g.disp12MaxDiff &&
                    0 <= x_ && x_ < W && disp2[x_] >= g.minD && abs(disp2[x_] - d_) > g.disp12MaxDiff)
                    drow[x] = (int16_t)g.invalid16;
            }
        }
    }
    free(bt.g1); free(pix); free(ring); free(ring_row); free(C); free(T); free(Lh);
    free(topMin); free(Rb); free(zero); free(disp2); free(disp2cost);
    if (dump && dump->raw) memcpy(dump->raw, disp, (size_t)W * H * sizeof(int16_t));
    return 0;
}

/* cv2.StereoSGBM.compute, mode = MODE_SGBM (0: one pass, paths from the left, top-left, top, top-right plus a
 * right->left path, 5 in all) and MODE_HH (1: a second pass bottom->top adds the four opposite paths, 8 in all).
 * The reference itself always asks for MODE_SGBM_3WAY (camera/single_usb_stereo_camera.py:273); these two modes are
 * restated from OpenCV 4.x computeDisparitySGBM for callers of the same matcher class.  Parity unpinned.
 *   - the aggregated cost is the same sliding block sum as above, over the whole image (no stripes), and is stored
 *     with P2 added to every entry; a path step is L = C' + min(prev[d], prev[d-1]+P1, prev[d+1]+P1, minPrev+P2)
 *     - (minPrev+P2), so a path entering the image (previous costs all zero) starts with L = C;
 *   - the summed cost takes the paths by saturating additions in the order left, top-left, top, top-right (first pass),
 *     right, bottom-left(x-1,y+1), bottom, bottom-right (second pass) resp. the right->left path (MODE_SGBM);
 *   - winner, uniqueness, sub-pixel step and the left-right check are those of the 3-path mode, run on the rows in
 *     the order of the last pass. */
static int sgbm_fulldp(const orc_sgbm_params *p, const uint8_t *left, const uint8_t *right,
                       int W, int H, int16_t *disp, const sgbm_dumps *dump)
{
    sgbm_geom g;
    int rc = sgbm_geometry(p, W, H, &g);
    if (rc) return rc;
    const int D = g.D, W1 = g.W1;
    if (D > 1024) return -3;
    for (size_t i = 0; i < (size_t)W * H; i++) disp[i] = (int16_t)g.invalid16;
    if (W1 <= 0) return 0;
    const int npasses = p->mode == 1 ? 2 : 1;

    bt_planes bt;
    bt_planes_alloc(&bt, W, H);
    bt_planes_fill(&bt, left, right, W, H, g.ftzero);

    const size_t rowN = (size_t)W1 * D;
    const int NR = 2 * g.SH2 + 2;
    int16_t *pix = (int16_t *)malloc(rowN * sizeof(int16_t));
    int16_t *ring = (int16_t *)malloc(rowN * NR * sizeof(int16_t));
    int *ring_row = (int *)malloc(NR * sizeof(int));
    int16_t *Cvol = (int16_t *)malloc(rowN * H * sizeof(int16_t));   /* with P2 added */
    int16_t *Svol = (int16_t *)calloc(rowN * H, sizeof(int16_t));
    /* path costs of the row in work and of the row before it: [2][W1 + 2 columns][4 paths][D], column c at c + 1 */
    const size_t lrRow = (size_t)(W1 + 2) * 4 * D;
    int16_t *Lr = (int16_t *)malloc(2 * lrRow * sizeof(int16_t));
    int16_t *minLr = (int16_t *)malloc(2 * (size_t)(W1 + 2) * 4 * sizeof(int16_t));
    int16_t *Rb = (int16_t *)malloc(D * sizeof(int16_t));
    int16_t *disp2 = (int16_t *)malloc(W * sizeof(int16_t));
    int16_t *disp2cost = (int16_t *)malloc(W * sizeof(int16_t));

    /* aggregated cost of every row */
    for (int i = 0; i < NR; i++) ring_row[i] = -1;
    for (int y = 0; y < H; y++) {
        int16_t *C = Cvol + (size_t)y * rowN;
        if (y == 0)
            for (size_t i = 0; i < rowN; i++) C[i] = (int16_t)g.P2;
        else
            memcpy(C, C - rowN, rowN * sizeof(int16_t));
        int k0 = (y == 0) ? 0 : y + g.SH2, k1 = y + g.SH2;
        for (int k = k0; k <= k1; k++) {
            int kk = imin(k, H - 1);
            int16_t *hs = ring + (size_t)(kk % NR) * rowN;
            if (ring_row[kk % NR] != kk) {
                pixcost_row(&g, &bt, kk, pix);
                hsum_row(&g, pix, hs);
                ring_row[kk % NR] = kk;
            }
            if (y == 0) {
                int scale = (k == 0) ? g.SH2 + 1 : 1;
                for (size_t i = 0; i < rowN; i++) C[i] = (int16_t)(C[i] + hs[i] * scale);
            } else {
                const int16_t *sub = ring + (size_t)(imax(y - g.SH2 - 1, 0) % NR) * rowN;
                for (size_t i = 0; i < rowN; i++) C[i] = (int16_t)(C[i] + hs[i] - sub[i]);
            }
        }
    }
    if (dump && dump->Cvol)   /* without the added P2, as the 3-path mode reports it */
        for (size_t i = 0; i < rowN * H; i++) dump->Cvol[i] = (int16_t)(Cvol[i] - g.P2);

    for (int pass = 1; pass <= npasses; pass++) {
        const int y1 = pass == 1 ? 0 : H - 1, y2 = pass == 1 ? H : -1, dy = pass == 1 ? 1 : -1;
        const int x1 = pass == 1 ? 0 : W1 - 1, x2 = pass == 1 ? W1 : -1, dx = pass == 1 ? 1 : -1;
        memset(Lr, 0, 2 * lrRow * sizeof(int16_t));
        memset(minLr, 0, 2 * (size_t)(W1 + 2) * 4 * sizeof(int16_t));
        int lrID = 0;
        for (int y = y1; y != y2; y += dy) {
            const int16_t *C = Cvol + (size_t)y * rowN;
            int16_t *S = Svol + (size_t)y * rowN;
            int16_t *cur = Lr + (size_t)lrID * lrRow, *prv = Lr + (size_t)(1 - lrID) * lrRow;
            int16_t *mcur = minLr + (size_t)lrID * (W1 + 2) * 4, *mprv = minLr + (size_t)(1 - lrID) * (W1 + 2) * 4;
            for (int x = x1; x != x2; x += dx) {
                const int16_t *Cp = C + (size_t)x * D;
                int16_t *Sp = S + (size_t)x * D;
                /* predecessors: (x - dx, this row), (x - 1, x, x + 1 of the row done before) */
                const int16_t *pr[4] = {cur + ((size_t)(x - dx + 1) * 4 + 0) * D, prv + ((size_t)(x - 1 + 1) * 4 + 1) * D,
                                        prv + ((size_t)(x + 1) * 4 + 2) * D, prv + ((size_t)(x + 1 + 1) * 4 + 3) * D};
                const int pm[4] = {mcur[(x - dx + 1) * 4 + 0], mprv[(x - 1 + 1) * 4 + 1], mprv[(x + 1) * 4 + 2], mprv[(x + 1 + 1) * 4 + 3]};
                for (int r = 0; r < 4; r++) {
                    int16_t *out = cur + ((size_t)(x + 1) * 4 + r) * D;
                    mcur[(x + 1) * 4 + r] = (int16_t)sgm_step(Cp, pr[r], pm[r], D, g.P1, g.P2, out);
                    for (int d = 0; d < D; d++) Sp[d] = (int16_t)sat16(Sp[d] + out[d]);
                }
            }
            if (pass == npasses) {
                int16_t *drow = disp + (size_t)y * W;
                for (int x = 0; x < W; x++) { disp2[x] = (int16_t)g.invalid16; disp2cost[x] = 32767; }
                memset(Rb, 0, D * sizeof(int16_t));
                int rightMin = 0;
                for (int xc = W1 - 1; xc >= 0; xc--) {
                    int16_t *Sx = S + (size_t)xc * D;
                    if (npasses == 1) {
                        rightMin = sgm_step(C + (size_t)xc * D, Rb, rightMin, D, g.P1, g.P2, Rb);
                        for (int d = 0; d < D; d++) Sx[d] = (int16_t)sat16(Sx[d] + Rb[d]);
                    }
                    int best = -1, minS = 32767;
                    for (int d = 0; d < D; d++)
                        if (Sx[d] < minS) { minS = Sx[d]; best = d; }
                    if (best < 0) best = 0;   /* every cost saturated: cannot happen below 32767, kept defined */
                    int d;
                    for (d = 0; d < D; d++)
                        if (Sx[d] * (100 - g.uniq) < minS * 100 && abs(d - best) > 1) break;
                    if (d < D) continue;
                    int x2c = xc + g.minX1 - best - g.minD;
                    if (x2c >= 0 && x2c < W && disp2cost[x2c] > minS) {
                        disp2cost[x2c] = (int16_t)minS;
                        disp2[x2c] = (int16_t)(best + g.minD);
                    }
                    int dsub;
                    if (0 < best && best < D - 1) {
                        int denom2 = imax(Sx[best - 1] + Sx[best + 1] - 2 * Sx[best], 1);
                        dsub = best * 16 + ((Sx[best - 1] - Sx[best + 1]) * 16 + denom2) / (denom2 * 2);
                    } else
                        dsub = best * 16;
                    drow[xc + g.minX1] = (int16_t)(dsub + g.minD * 16);
                }
                for (int x = g.minX1; x < g.maxX1; x++) {
                    int d1 = drow[x];
                    if (d1 == g.invalid16) continue;
                    int _d = d1 >> 4;
                    int d_ = (d1 + 15) >> 4;
                    int _x = x - _d, x_ = x - d_;
                    if (0 <= _x && _x < W && disp2[_x] >= g.minD && abs(disp2[_x] - _d) > g.disp12MaxDiff &&
                        0 <= x_ && x_ < W && disp2[x_] >= g.minD && abs(disp2[x_] - d_) > g.disp12MaxDiff)
                        drow[x] = (int16_t)g.invalid16;
                }
            }
            lrID = 1 - lrID;
        }
    }
    if (dump && dump->Svol) memcpy(dump->Svol, Svol, rowN * H * sizeof(int16_t));
    free(bt.g1); free(pix); free(ring); free(ring_row); free(Cvol); free(Svol); free(Lr); free(minLr);
    free(Rb); free(disp2); free(disp2cost);
    if (dump && dump->raw) memcpy(dump->raw, disp, (size_t)W * H * sizeof(int16_t));
    return 0;
}

/* G7a: cv::medianBlur(disp, disp, 3) on CV_16S, replicated border */
ORC_API void orc_median3_s16(const int16_t *src, int W, int H, int16_t *dst)
{
    for (int y = 0; y < H; y++) {
        int ym = imax(y - 1, 0), yp = imin(y + 1, H - 1);
        for (int x = 0; x < W; x++) {
            int xm = imax(x - 1, 0), xp = imin(x + 1, W - 1);
            int16_t v[9] = {
                src[(size_t)ym * W + xm], src[(size_t)ym * W + x], src[(size_t)ym * W + xp],
                src[(size_t)y * W + xm],  src[(size_t)y * W + x],  src[(size_t)y * W + xp],
                src[(size_t)yp * W + xm], src[(size_t)yp * W + x], src[(size_t)yp * W + xp]};
            for (int i = 1; i < 9; i++) { /* insertion sort */
                int16_t t = v[i];
                int j = i - 1;
                while (j >= 0 && v[j] > t) { v[j + 1] = v[j]; j--; }
                v[j + 1] = t;
            }
            dst[(size_t)y * W + x] = v[4];
        }
    }
}

/* G7b: cv::filterSpeckles(img, newVal, maxSpeckleSize, maxDiff): 4-connected
 * components under |a-b| <= maxDiff (pixels == newVal excluded); components of
 * at most maxSpeckleSize pixels are set to newVal. */
ORC_API void orc_filter_speckles(int16_t *img, int W, int H, int newVal, int maxSpeckleSize, int maxDiff)
{
    size_t n = (size_t)W * H;
    int *labels = (int *)calloc(n, sizeof(int));
    int *stack = (int *)malloc(n * sizeof(int));
    uint8_t *rtype = (uint8_t *)calloc(n + 1, 1);
    int cur = 0;
    for (int i = 0; i < H; i++)
        for (int j = 0; j < W; j++) {
            size_t idx = (size_t)i * W + j;
            if (img[idx] == newVal) continue;
            if (labels[idx]) {
                if (rtype[labels[idx]]) img[idx] = (int16_t)newVal;
                continue;
            }
            int sp = 0, count = 0;
            cur++;
            labels[idx] = cur;
            stack[sp++] = (int)idx;
            while (sp > 0) {
                int q = stack[--sp];
                count++;
                int qy = q / W, qx = q % W;
                int dp = img[q];
                if (qy < H - 1 && !labels[q + W] && img[q + W] != newVal && abs(dp - img[q + W]) <= maxDiff) { labels[q + W] = cur; stack[sp++] = q + W; }
                if (qy > 0 && !labels[q - W] && img[q - W] != newVal && abs(dp - img[q - W]) <= maxDiff) { labels[q - W] = cur; stack[sp++] = q - W; }
                if (qx < W - 1 && !labels[q + 1] && img[q + 1] != newVal && abs(dp - img[q + 1]) <= maxDiff) { labels[q + 1] = cur; stack[sp++] = q + 1; }
                if (qx > 0 && !labels[q - 1] && img[q - 1] != newVal && abs(dp - img[q - 1]) <= maxDiff) { labels[q - 1] = cur; stack[sp++] = q - 1; }
            }
            if (count <= maxSpeckleSize) { rtype[cur] = 1; img[idx] = (int16_t)newVal; }
        }
    free(labels); free(stack); free(rtype);
}

static int sgbm_full(const orc_sgbm_params *p, const uint8_t *left, const uint8_t *right,
                     int W, int H, int16_t *disp, const sgbm_dumps *dump)
{
    if (p->mode < 0 || p->mode > 2) return -10;
    int rc = p->mode == 2 ? sgbm_3way(p, left, right, W, H, disp, dump) : sgbm_fulldp(p, left, right, W, H, disp, dump);
    if (rc) return rc;
    int16_t *tmp = (int16_t *)malloc((size_t)W * H * sizeof(int16_t));
    orc_median3_s16(disp, W, H, tmp);
    memcpy(disp, tmp, (size_t)W * H * sizeof(int16_t));
    free(tmp);
    if (dump && dump->median) memcpy(dump->median, disp, (size_t)W * H * sizeof(int16_t));
    if (p->speckleWindowSize > 0)
        orc_filter_speckles(disp, W, H, (p->minDisparity - 1) * 16, p->speckleWindowSize, 16 * p->speckleRange);
    return 0;
}

ORC_API int orc_sgbm_compute(const orc_sgbm_params *p, const uint8_t *left, const uint8_t *right,
                             int W, int H, int16_t *disp)
{
    return sgbm_full(p, left, right, W, H, disp, NULL);
}

ORC_API int orc_sgbm_compute_debug(const orc_sgbm_params *p, const uint8_t *left, const uint8_t *right,
                                   int W, int H, int16_t *disp, int16_t *Cvol, int16_t *Tvol,
                                   int16_t *Svol, int16_t *raw, int16_t *median)
{
    sgbm_dumps d = {Cvol, Tvol, Svol, raw, median};
    return sgbm_full(p, left, right, W, H, disp, &d);
}

/* geometry query used by tests: out = {minX1, W1, stripe_sz, stripe_overlap, invalid16, ftzero, P1, P2} */
ORC_API int orc_sgbm_geometry(const orc_sgbm_params *p, int W, int H, int *out8)
{
    sgbm_geom g;
    int rc = sgbm_geometry(p, W, H, &g);
    out8[0] = g.minX1; out8[1] = g.W1; out8[2] = g.stripe_sz; out8[3] = g.stripe_overlap;
    out8[4] = g.invalid16; out8[5] = g.ftzero; out8[6] = g.P1; out8[7] = g.P2;
    return rc;
}

/* ------------------------------------------------------------------------ */
/* G9  cv2.ximgproc.DisparityWLSFilter.filter(dL, left_gray, None, dR)        */
/* call site: camera/single_usb_stereo_camera.py:280-282, 328-332             */
/* ------------------------------------------------------------------------ */
typedef struct {
    double lambda;       /* setLambda, reference 8000 */
    double sigma_color;  /* setSigmaColor, reference 1.5 */
    int min_disparity;   /* of the LEFT matcher */
    int num_disparities;
    int lrc_thresh;      /* 24 */
    int disc_radius;     /* ceil(0.5 * blockSize) for SGBM */
    float disc_rolloff;  /* 0.001 */
    int fgs_iters;       /* 3 */
    float fgs_attenuation; /* 0.25 */
} orc_wls_params;

/* depth-discontinuity confidence of one disparity sub-image (x in [x0,x0+rw)):
 *   var = box(d^2) - box(d)^2 over a (2r+1)^2 window; conf = max(0, 1 - rolloff * var).
 * The filter converts the sub-image to a float matrix of its own and calls cv::boxFilter with the
 * default border, so the window is extended by BORDER_REFLECT_101 of the sub-image (not by the
 * columns outside it).  boxFilter on float input accumulates in double (exact here: the values are
 * integers) and scales by the double 1/(window area) before rounding to float once. */
static void wls_disc_map(const int16_t *disp, int W, int H, int x0, int rw, int r, float rolloff, float *out /*H*rw*/)
{
    const double scale = 1.0 / ((2 * r + 1) * (2 * r + 1));
    int64_t *s1 = (int64_t *)malloc((size_t)H * rw * sizeof(int64_t));
    int64_t *s2 = (int64_t *)malloc((size_t)H * rw * sizeof(int64_t));
    for (int y = 0; y < H; y++)
        for (int x = 0; x < rw; x++) {
            int64_t a = 0, b = 0;
            for (int j = -r; j <= r; j++) {
                int v = disp[(size_t)y * W + x0 + reflect101(x + j, rw)];
                a += v; b += (int64_t)v * v;
            }
            s1[(size_t)y * rw + x] = a; s2[(size_t)y * rw + x] = b;
        }
    for (int y = 0; y < H; y++)
        for (int x = 0; x < rw; x++) {
            int64_t a = 0, b = 0;
            for (int j = -r; j <= r; j++) {
                int yy = reflect101(y + j, H);
                a += s1[(size_t)yy * rw + x]; b += s2[(size_t)yy * rw + x];
            }
            float m1 = (float)((double)a * scale);
            float m2 = (float)((double)b * scale);
            float var = m2 - m1 * m1;
            float c = 1.0f - rolloff * var;
            out[(size_t)y * rw + x] = c > 0.0f ? c : 0.0f;
        }
    free(s1); free(s2);
}

/* Fast Global Smoother (Min et al. 2014) as in cv::ximgproc: solves
 * (I + lambda_t * A) u = f along every row, then every column, num_iter times,
 * lambda_t *= attenuation after each iteration.  Edge weights come from a LUT:
 * w = -exp(-|I_p - I_q| / sigma).  'planes' images are filtered with shared
 * coefficients.  All arithmetic in float, sequential Thomas algorithm. */
static void fgs_filter(const uint8_t *guide, ptrdiff_t gstride, int w, int h, float lambda0, float sigma,
                       int iters, float atten, float **planes, int nplanes)
{
    float lut[256];
    for (int i = 0; i < 256; i++) lut[i] = (float)(-exp(-sqrt((double)(i * i)) / sigma));
    float *ch = (float *)malloc((size_t)w * h * sizeof(float));
    float *cv = (float *)malloc((size_t)w * h * sizeof(float));
    float *inter = (float *)malloc((size_t)imax(w, h) * sizeof(float));
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            int c = guide[y * gstride + x];
            ch[(size_t)y * w + x] = x < w - 1 ? lut[abs(c - guide[y * gstride + x + 1])] : 0.0f;
            cv[(size_t)y * w + x] = y < h - 1 ? lut[abs(c - guide[(y + 1) * gstride + x])] : 0.0f;
        }
    float lambda = lambda0;
    for (int it = 0; it < iters; it++) {
        /* horizontal */
        for (int y = 0; y < h; y++) {
            const float *c = ch + (size_t)y * w;
            for (int pl = 0; pl < nplanes; pl++) {
                float *u = planes[pl] + (size_t)y * w;
                float lower = 0.0f;
                float prev_inter = 0.0f, prev_u = 0.0f;
                for (int x = 0; x < w; x++) {
                    float upper = lambda * c[x];
                    float denom = (1.0f - lower - upper) - lower * prev_inter;
                    inter[x] = upper / denom;
                    u[x] = (u[x] - lower * prev_u) / denom;
                    prev_inter = inter[x]; prev_u = u[x];
                    lower = upper;
                }
                for (int x = w - 2; x >= 0; x--) u[x] = u[x] - inter[x] * u[x + 1];
            }
        }
        /* vertical */
        for (int x = 0; x < w; x++) {
            for (int pl = 0; pl < nplanes; pl++) {
                float *u = planes[pl] + x;
                float lower = 0.0f, prev_inter = 0.0f, prev_u = 0.0f;
                for (int y = 0; y < h; y++) {
                    float upper = lambda * cv[(size_t)y * w + x];
                    float denom = (1.0f - lower - upper) - lower * prev_inter;
                    inter[y] = upper / denom;
                    float val = (u[(size_t)y * w] - lower * prev_u) / denom;
                    u[(size_t)y * w] = val;
                    prev_inter = inter[y]; prev_u = val;
                    lower = upper;
                }
                for (int y = h - 2; y >= 0; y--) u[(size_t)y * w] = u[(size_t)y * w] - inter[y] * u[(size_t)(y + 1) * w];
            }
        }
        lambda *= atten;
    }
    free(ch); free(cv); free(inter);
}

/* conf_out (optional): H*W float confidence map (x255), zero outside the ROI */
ORC_API int orc_wls_filter(const orc_wls_params *p, const int16_t *dL, const int16_t *dR, const uint8_t *guide,
                           int W, int H, int16_t *out, float *conf_out)
{
    const int left_off = imax(0, p->min_disparity + p->num_disparities);
    const int right_off = imax(0, -p->min_disparity);
    const int rw = W - left_off - right_off;
    const int16_t fill = (int16_t)(16 * (p->min_disparity - 1));
    for (size_t i = 0; i < (size_t)W * H; i++) out[i] = fill;
    if (conf_out) memset(conf_out, 0, (size_t)W * H * sizeof(float));
    if (rw <= 0) return 0;
    const int lx0 = left_off, rx0 = right_off; /* left-view ROI x0, right-view ROI x0 */
    float *discL = (float *)malloc((size_t)H * rw * sizeof(float));
    float *discR = (float *)malloc((size_t)H * rw * sizeof(float));
    wls_disc_map(dL, W, H, lx0, rw, p->disc_radius, p->disc_rolloff, discL);
    wls_disc_map(dR, W, H, rx0, rw, p->disc_radius, p->disc_rolloff, discR);
    float *conf = (float *)calloc((size_t)H * rw, sizeof(float));
    float *dc = (float *)malloc((size_t)H * rw * sizeof(float));
    for (int y = 0; y < H; y++)
        for (int x = 0; x < rw; x++) {
            int j = lx0 + x;
            int dl = dL[(size_t)y * W + j];
            int ridx = j - (dl >> 4);
            float c = 0.0f;
            if (ridx >= rx0 && ridx < rx0 + rw) {
                if (abs(dl + dR[(size_t)y * W + ridx]) < p->lrc_thresh) {
                    float a = discL[(size_t)y * rw + x], b = discR[(size_t)y * rw + (ridx - rx0)];
                    c = a < b ? a : b;
                }
            }
            c = 255.0f * c;
            conf[(size_t)y * rw + x] = c;
            dc[(size_t)y * rw + x] = c * (float)dl;
            if (conf_out) conf_out[(size_t)y * W + j] = c;
        }
    float *planes[2] = {dc, conf};
    fgs_filter(guide + lx0, W, rw, H, (float)p->lambda, (float)p->sigma_color, p->fgs_iters, p->fgs_attenuation, planes, 2);
    for (int y = 0; y < H; y++)
        for (int x = 0; x < rw; x++) {
            float t = 1.0f / (conf[(size_t)y * rw + x] + 1e-43f);
            float v = dc[(size_t)y * rw + x] * t;
            int iv;
            if (!(v == v)) iv = -32768; /* NaN -> INT_MIN -> saturates */
            else if (v >= 32767.0f) iv = 32767;
            else if (v <= -32768.0f) iv = -32768;
            else iv = sat16(round_half_even((double)v));
            out[(size_t)y * W + lx0 + x] = (int16_t)iv;
        }
    free(discL); free(discR); free(conf); free(dc);
    return 0;
}

/* ------------------------------------------------------------------------ */
/* G10..G12  disparity/16 -> reprojectImageTo3D Z -> validity masks           */
/* call site: camera/single_usb_stereo_camera.py:335-357                      */
/* Q == NULL selects the uncalibrated fallback (Z = 0.06*350/d).              */
/* xyz (optional, needs Q): full H*W*3 reprojectImageTo3D output, unmasked.   */
/* ------------------------------------------------------------------------ */
ORC_API void orc_depth_from_disp16(const int16_t *disp16, int W, int H, const double *Q, float *depth, float *xyz)
{
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++) {
            size_t i = (size_t)y * W + x;
            float d = (float)disp16[i] / 16.0f;
            float z;
            if (Q) {
                double dd = d;
                double hp[4];
                for (int r = 0; r < 4; r++) {
                    double s = 0.0;
                    s += Q[4 * r + 0] * (double)x;
                    s += Q[4 * r + 1] * (double)y;
                    s += Q[4 * r + 2] * dd;
                    s += Q[4 * r + 3] * 1.0;
                    hp[r] = s;
                }
                float X = (float)((double)(float)hp[0] / hp[3]);
                float Y = (float)((double)(float)hp[1] / hp[3]);
                float Z = (float)((double)(float)hp[2] / hp[3]);
                if (xyz) { xyz[3 * i] = X; xyz[3 * i + 1] = Y; xyz[3 * i + 2] = Z; }
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
}

/* ------------------------------------------------------------------------ */
/* S1..S4  FastStegerExtractor.extract_centerline                             */
/* core/laser_extractor.py:160-261                                            */
/* ------------------------------------------------------------------------ */
/* cv::getGaussianKernel(ksize, sigma, CV_32F) with ksize = round(sigma*8+1)|1 */
ORC_API int orc_gauss_kernel(double sigma, float *taps, int max_taps)
{
    int ks = round_half_even(sigma * 4 * 2 + 1) | 1;
    if (ks > max_taps) return -1;
    double sum = 0, t[512];
    double scale2x = -0.5 / (sigma * sigma);
    for (int i = 0; i < ks; i++) {
        double x = i - (ks - 1) * 0.5;
        t[i] = exp(scale2x * x * x);
        sum += t[i];
    }
    for (int i = 0; i < ks; i++) taps[i] = (float)(t[i] * (1.0 / sum));
    return ks;
}

/* Gaussian blur of a float image: row pass (taps accumulated left to right),
 * then column pass (centre tap first, then symmetric pairs), REFLECT_101. */
static void gauss_blur_f32(const float *src, int w, int h, const float *taps, int ks, float *dst)
{
    const int r = ks / 2;
    float *tmp = (float *)malloc((size_t)w * h * sizeof(float));
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            float s = 0.0f;
            for (int k = 0; k < ks; k++) s = s + taps[k] * src[(size_t)y * w + reflect101(x + k - r, w)];
            tmp[(size_t)y * w + x] = s;
        }
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            float s = taps[r] * tmp[(size_t)y * w + x];
            for (int k = 1; k <= r; k++) {
                float a = tmp[(size_t)reflect101(y - k, h) * w + x];
                float b = tmp[(size_t)reflect101(y + k, h) * w + x];
                s = s + taps[r + k] * (a + b);
            }
            dst[(size_t)y * w + x] = s;
        }
    free(tmp);
}

/* cv2.GaussianBlur(src_f32, (0, 0), sigma, sigma) as restated above, exported for the golden-vector generator
 * (tests/golden/make_laser_golden.py), which runs the reference's own per-pixel code on these images. */
ORC_API int orc_gauss_blur_f32(const float *src, int w, int h, double sigma, float *dst)
{
    float taps[512];
    int ks = orc_gauss_kernel(sigma, taps, 512);
    if (ks < 0) return -1;
    gauss_blur_f32(src, w, h, taps, ks, dst);
    return 0;
}

/* Steger decision for one pixel from the five derivative images.
 * returns 1 and writes (ox, oy) offsets if accepted. */
static int steger_point(float dx, float dy, float dxx, float dyy, float dxy, double *ox, double *oy)
{
    double a = dxx, b = dxy, c = dyy;
    double nx, ny;
    if (b == 0.0) {
        /* diagonal matrix: eigenvalues (a, c), eigenvectors (1,0), (0,1) */
        if (fabs(a) >= fabs(c)) { nx = 1.0; ny = 0.0; }
        else { nx = 0.0; ny = 1.0; }
    } else {
        double m = 0.5 * (a + c);
        double hd = 0.5 * (a - c);
        double rr = sqrt(hd * hd + b * b);
        double lam = (m >= 0.0) ? m + rr : m - rr; /* eigenvalue of largest magnitude */
        /* eigenvector: (b, lam - a) or (lam - c, b), pick the better conditioned */
        double v1x = b, v1y = lam - a;
        double v2x = lam - c, v2y = b;
        double n1 = v1x * v1x + v1y * v1y, n2 = v2x * v2x + v2y * v2y;
        if (n1 >= n2) { double s = 1.0 / sqrt(n1); nx = v1x * s; ny = v1y * s; }
        else { double s = 1.0 / sqrt(n2); nx = v2x * s; ny = v2y * s; }
    }
    double den = nx * nx * a + 2.0 * nx * ny * b + ny * ny * c;
    if (!(fabs(den) > 1e-10)) return 0;
    double t = -(nx * (double)dx + ny * (double)dy) / den;
    double tx = t * nx, ty = t * ny;
    if (fabs(tx) <= 0.5 && fabs(ty) <= 0.5) { *ox = tx; *oy = ty; return 1; }
    return 0;
}

/* img: u8, channels 1 (gray) or 3 (BGR); roi = {x,y,w,h} or NULL.
 * xy: capacity*2 floats (x,y) in raster order; *count = number found (may
 * exceed capacity; only the first 'capacity' are stored).
 * deriv (optional): 5 planes (dx,dy,dxx,dyy,dxy) of rw*rh floats. */
ORC_API int orc_steger_extract(const uint8_t *img, int channels, int W, int H, ptrdiff_t row_stride,
                               const int *roi, double sigma, int threshold,
                               float *xy, int capacity, int *count, float *deriv)
{
    if (channels != 1 && channels != 3) return -1;
    int x0 = 0, y0 = 0, rw = W, rh = H;
    if (roi) {
        /* numpy slicing semantics of gray[y:y+h, x:x+w] for non-negative x, y */
        x0 = roi[0]; y0 = roi[1];
        if (x0 < 0 || y0 < 0) return -2;
        rw = imax(0, imin(W, x0 + roi[2]) - imin(x0, W));
        rh = imax(0, imin(H, y0 + roi[3]) - imin(y0, H));
    }
    *count = 0;
    if (rw <= 0 || rh <= 0) return 0;
    float taps[512];
    int ks = orc_gauss_kernel(sigma, taps, 512);
    if (ks < 0) return -3;
    size_t n = (size_t)rw * rh;
    uint8_t *g8 = (uint8_t *)malloc(n);
    float *gf = (float *)malloc(n * sizeof(float));
    float *bl = (float *)malloc(n * sizeof(float));
    for (int y = 0; y < rh; y++) {
        const uint8_t *s = img + (ptrdiff_t)(y + y0) * row_stride;
        for (int x = 0; x < rw; x++) {
            int v;
            if (channels == 3) {
                const uint8_t *px = s + 3 * (x + x0);
                v = (px[2] * 9798 + px[1] * 19235 + px[0] * 3735 + (1 << 14)) >> 15;
            } else
                v = s[x + x0];
            g8[(size_t)y * rw + x] = (uint8_t)v;
            gf[(size_t)y * rw + x] = (float)v;
        }
    }
    gauss_blur_f32(gf, rw, rh, taps, ks, bl);
    int cnt = 0;
    for (int j = 0; j < rh; j++)
        for (int i = 0; i < rw; i++) {
            int im = reflect101(i - 1, rw), ip = reflect101(i + 1, rw);
            int jm = reflect101(j - 1, rh), jp = reflect101(j + 1, rh);
            float c = bl[(size_t)j * rw + i];
            float l = bl[(size_t)j * rw + im], r = bl[(size_t)j * rw + ip];
            float u = bl[(size_t)jm * rw + i], d = bl[(size_t)jp * rw + i];
            float ul = bl[(size_t)jm * rw + im];
            /* cv2.filter2D is correlation with anchor = ksize/2 */
            float dx = l - c;
            float dy = u - c;
            float dxx = (l + (-2.0f * c)) + r;
            float dyy = (u + (-2.0f * c)) + d;
            float dxy = ((ul - u) - l) + c;
            if (deriv) {
                size_t o = (size_t)j * rw + i;
                deriv[o] = dx; deriv[n + o] = dy; deriv[2 * n + o] = dxx; deriv[3 * n + o] = dyy; deriv[4 * n + o] = dxy;
            }
            if (g8[(size_t)j * rw + i] > threshold) {
                double ox, oy;
                if (steger_point(dx, dy, dxx, dyy, dxy, &ox, &oy)) {
                    if (cnt < capacity) {
                        xy[2 * cnt] = (float)((double)i + ox + (double)x0);
                        xy[2 * cnt + 1] = (float)((double)j + oy + (double)y0);
                    }
                    cnt++;
                }
            }
        }
    *count = cnt;
    free(g8); free(gf); free(bl);
    return 0;
}

/* ------------------------------------------------------------------------ */
/* R1..R3  Reconstructor (core/reconstruction.py:30-182)                      */
/* ------------------------------------------------------------------------ */
/* reconstruct_from_depth: xy = n (u,v) doubles; K row-major 3x3; out n*3.   */
ORC_API int orc_reconstruct_from_depth(const double *uv, int n, const float *depth, int W, int H,
                                       const double *K, double *xyz, int *n_out)
{
    double fx = K[0], fy = K[4], cx = K[2], cy = K[5];
    int m = 0;
    for (int i = 0; i < n; i++) {
        double u = uv[2 * i], v = uv[2 * i + 1];
        /* Python int() truncates toward zero */
        long iu = (long)u, iv = (long)v;
        if (!(u == u) || !(v == v)) continue;
        if (0 <= iu && iu < W && 0 <= iv && iv < H) {
            float dep = depth[(size_t)iv * W + iu];
            if (dep > 0) {
                double z = (double)dep / 1000.0;
                xyz[3 * m] = (u - cx) * z / fx;
                xyz[3 * m + 1] = (v - cy) * z / fy;
                xyz[3 * m + 2] = z;
                m++;
            }
        }
    }
    *n_out = m;
    return 0;
}

/* reconstruct_point (+ optional refraction) for n pixels; NaN rows kept. */
ORC_API int orc_reconstruct_plane(const double *uv, int n, const double *Kinv, const double *plane,
                                  int refraction, double n_water, double *xyz)
{
    const double a = plane[0], b = plane[1], c = plane[2], d = plane[3];
    for (int i = 0; i < n; i++) {
        double u = uv[2 * i], v = uv[2 * i + 1];
        double r[3];
        for (int k = 0; k < 3; k++) r[k] = Kinv[3 * k] * u + Kinv[3 * k + 1] * v + Kinv[3 * k + 2] * 1.0;
        double nr = sqrt(r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);
        r[0] /= nr; r[1] /= nr; r[2] /= nr;
        double den = a * r[0] + b * r[1] + c * r[2];
        double *o = xyz + 3 * i;
        if (fabs(den) < 1e-10) { o[0] = o[1] = o[2] = NAN; continue; }
        double t = -d / den;
        if (t < 0) { o[0] = o[1] = o[2] = NAN; continue; }
        double P[3] = {t * r[0], t * r[1], t * r[2]};
        if (refraction) {
            double n1 = 1.0, n2 = n_water;
            double cos1 = -(r[2]);
            double sin1 = sqrt(1 - cos1 * cos1);
            double sin2 = (n1 / n2) * sin1;
            if (!(sin2 > 1)) {
                double cos2 = sqrt(1 - sin2 * sin2);
                double q[3] = {(n1 / n2) * r[0], (n1 / n2) * r[1], (n1 / n2) * r[2] + ((n1 / n2) * cos1 - cos2)};
                double nq = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2]);
                q[0] /= nq; q[1] /= nq; q[2] /= nq;
                double den2 = a * q[0] + b * q[1] + c * q[2];
                if (!(fabs(den2) < 1e-10)) {
                    double t2 = -d / den2;
                    P[0] = t2 * q[0]; P[1] = t2 * q[1]; P[2] = t2 * q[2];
                }
            }
        }
        o[0] = P[0]; o[1] = P[1]; o[2] = P[2];
    }
    return 0;
}

ORC_API const char *orc_version(void) { return "l3d-oracle 0.1"; }

/* ------------------------------------------------------------------------ */
/* H0..H4  SimpleLaserExtractor.extract_centerline                            */
/* core/laser_extractor.py:45-100                                             */
/*   cv2.cvtColor(BGR2HSV) + inRange, cv2.cvtColor(BGR2GRAY) + threshold,     */
/*   bitwise_and, morphologyEx CLOSE then OPEN with a 3x3 rectangle,          */
/*   findContours(RETR_EXTERNAL, CHAIN_APPROX_SIMPLE), contourArea > min_area,*/
/*   drawContours(filled), per-row intensity-weighted centroid.               */
/* Restated sequentially, the way OpenCV does it: Suzuki-Abe border following */
/* of each top-level 8-connected component from its raster-first pixel,       */
/* shoelace area of the traced chain, even-odd scanline fill of the chain     */
/* polygon plus the chain pixels themselves.                                  */
/* ------------------------------------------------------------------------ */
static void morph3x3(const uint8_t *src, int W, int H, int dilate, uint8_t *dst)
{
    /* pixels outside the image never win (erode: +inf, dilate: -inf) */
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++) {
            int acc = dilate ? 0 : 255;
            for (int dy = -1; dy <= 1; dy++)
                for (int dx = -1; dx <= 1; dx++) {
                    int yy = y + dy, xx = x + dx;
                    if (yy < 0 || yy >= H || xx < 0 || xx >= W) continue;
                    int v = src[(size_t)yy * W + xx];
                    acc = dilate ? imax(acc, v) : imin(acc, v);
                }
            dst[(size_t)y * W + x] = (uint8_t)acc;
        }
}


/* External contours of a binary mask the way cv2.findContours(RETR_EXTERNAL) orders and traces them: every top-level
 * 8-connected component (one that touches background 4-connected to the image frame), in raster order of its first
 * pixel, Suzuki-Abe border following from that pixel.  visit(chain, nchain, PW, ctx) gets the traced chain as indices
 * into the image padded by a 1-pixel zero frame (row = idx / PW - 1, column = idx % PW - 1). */
typedef void (*orc_chain_fn)(const int *chain, int nchain, int PW, void *ctx);
static void trace_external_contours(const uint8_t *m0, int W, int H, orc_chain_fn visit, void *ctx)
{
    const int PW = W + 2, PH = H + 2;
    uint8_t *f = (uint8_t *)calloc((size_t)PW * PH, 1);
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++) f[(size_t)(y + 1) * PW + x + 1] = m0[(size_t)y * W + x] ? 1 : 0;
    int *comp = (int *)calloc((size_t)PW * PH, sizeof(int));     /* 8-connected foreground labels, 1.. */
    uint8_t *outer = (uint8_t *)calloc((size_t)PW * PH, 1);      /* background 4-connected to the frame */
    int *stack = (int *)malloc((size_t)PW * PH * sizeof(int));
    /* outer background */
    {
        int sp = 0;
        stack[sp++] = 0; outer[0] = 1;
        while (sp) {
            int q = stack[--sp], qy = q / PW, qx = q % PW;
            const int dy4[4] = {0, 0, 1, -1}, dx4[4] = {1, -1, 0, 0};
            for (int k = 0; k < 4; k++) {
                int yy = qy + dy4[k], xx = qx + dx4[k];
                if (yy < 0 || yy >= PH || xx < 0 || xx >= PW) continue;
                int t = yy * PW + xx;
                if (!f[t] && !outer[t]) { outer[t] = 1; stack[sp++] = t; }
            }
        }
    }
    /* direction table, counter-clockwise on screen: E, NE, N, NW, W, SW, S, SE (dy, dx) */
    static const int DY[8] = {0, -1, -1, -1, 0, 1, 1, 1}, DX[8] = {1, 1, 0, -1, -1, -1, 0, 1};
    int ncomp = 0;
    int *chain = (int *)malloc((size_t)PW * PH * 4 * sizeof(int)); /* a border pixel is visited at most 4 times */
    for (int i = 1; i <= H; i++)
        for (int j = 1; j <= W; j++) {
            int s = i * PW + j;
            if (!f[s] || comp[s]) continue;
            /* new component: flood fill (8-connectivity), note whether it touches the outer background */
            ncomp++;
            int sp = 0, top_level = 0;
            stack[sp++] = s; comp[s] = ncomp;
            while (sp) {
                int q = stack[--sp], qy = q / PW, qx = q % PW;
                for (int k = 0; k < 8; k++) {
                    int t = (qy + DY[k]) * PW + qx + DX[k];
                    if (f[t] && !comp[t]) { comp[t] = ncomp; stack[sp++] = t; }
                    if ((k & 1) == 0 && !f[t] && outer[t]) top_level = 1;
                }
            }
            if (!top_level) continue; /* lies inside a hole of another component: not an external contour */
            /* Suzuki-Abe border following from the raster-first pixel (i, j); its west neighbour is background */
            int nchain = 0;
            int i1 = -1, j1 = -1;
            for (int k = 0; k < 8; k++) { /* clockwise from W: W, NW, N, NE, E, SE, S, SW */
                int dir = (4 - k + 8) & 7;
                if (f[(i + DY[dir]) * PW + j + DX[dir]]) { i1 = i + DY[dir]; j1 = j + DX[dir]; break; }
            }
            chain[nchain++] = s;
            if (i1 >= 0) {
                int i2 = i1, j2 = j1, i3 = i, j3 = j;
                for (;;) {
                    /* direction of (i2,j2) seen from (i3,j3) */
                    int d2 = 0;
                    for (int k = 0; k < 8; k++)
                        if (i3 + DY[k] == i2 && j3 + DX[k] == j2) d2 = k;
                    int i4 = -1, j4 = -1;
                    for (int k = 1; k <= 8; k++) { /* counter-clockwise, starting after (i2,j2) */
                        int dir = (d2 + k) & 7;
                        if (f[(i3 + DY[dir]) * PW + j3 + DX[dir]]) { i4 = i3 + DY[dir]; j4 = j3 + DX[dir]; break; }
                    }
                    if (i4 == i && j4 == j && i3 == i1 && j3 == j1) break;
                    i2 = i3; j2 = j3; i3 = i4; j3 = j4;
                    chain[nchain++] = i3 * PW + j3;
                }
            }
            visit(chain, nchain, PW, ctx);
        }
    free(f); free(comp); free(outer); free(stack); free(chain);
}

/* cv2.contourArea: |sum(prev.x * p.y - prev.y * p.x)| / 2 */
static double chain_area(const int *chain, int nchain, int PW)
{
    double a00 = 0;
    int prev = chain[nchain - 1];
    for (int k = 0; k < nchain; k++) {
        int p = chain[k];
        double px_ = prev % PW, py_ = prev / PW, qx = p % PW, qy = p / PW;
        a00 += px_ * qy - py_ * qx;
        prev = p;
    }
    return fabs(a00 * 0.5);
}

/* cv2.drawContours(..., thickness=-1) of one contour: chain pixels + even-odd interior */
static void fill_chain(const int *chain, int nchain, int PW, int W, int H, uint8_t *fin, uint8_t value)
{
    (void)H;
    const int PH = H + 2;
    double *cross_x = (double *)malloc(((size_t)nchain + 4) * sizeof(double));
    int ymin = PH, ymax = -1;
    for (int k = 0; k < nchain; k++) {
        int p = chain[k];
        fin[(size_t)(p / PW - 1) * W + (p % PW - 1)] = value;
        ymin = imin(ymin, p / PW); ymax = imax(ymax, p / PW);
    }
    for (int y = ymin; y <= ymax; y++) {
        int nc = 0;
        for (int k = 0; k < nchain; k++) {
            int a = chain[k], b = chain[(k + 1) % nchain];
            int y1 = a / PW, y2 = b / PW;
            if ((y1 <= y) == (y2 <= y)) continue;
            cross_x[nc++] = (y1 < y2) ? (double)(a % PW) : (double)(b % PW); /* x at the endpoint on row y */
        }
        for (int x = 1; x <= W; x++) {
            int c = 0;
            for (int k = 0; k < nc; k++) c += cross_x[k] > x;
            if (c & 1) fin[(size_t)(y - 1) * W + (x - 1)] = value;
        }
    }
    free(cross_x);
}

struct simple_fill_ctx { int W, H; double min_area; uint8_t *fin; };
static void simple_fill_visit(const int *chain, int nchain, int PW, void *vctx)
{
    struct simple_fill_ctx *c = (struct simple_fill_ctx *)vctx;
    if (!(chain_area(chain, nchain, PW) > c->min_area)) return;
    fill_chain(chain, nchain, PW, c->W, c->H, c->fin, 255);
}


/* The pieces of the mask pipeline on their own, for the golden-vector generator (tests/golden/make_laser_golden.py). */
ORC_API void orc_morph3x3(const uint8_t *src, int W, int H, int dilate, uint8_t *dst) { morph3x3(src, W, H, dilate, dst); }

struct collect_ctx { int *xy; int cap_pts; int *lens; int max_contours; int n, used; };
static void collect_visit(const int *chain, int nchain, int PW, void *vctx)
{
    struct collect_ctx *c = (struct collect_ctx *)vctx;
    if (c->n < c->max_contours && c->used + nchain <= c->cap_pts) {
        for (int k = 0; k < nchain; k++) {
            c->xy[2 * (c->used + k)] = chain[k] % PW - 1;
            c->xy[2 * (c->used + k) + 1] = chain[k] / PW - 1;
        }
        c->lens[c->n] = nchain;
        c->used += nchain;
    }
    c->n++;
}
/* Traced external contours as (x, y) points, contour after contour; lens[i] = points of contour i.  Returns the number
 * of contours found (only those that fit both capacities are stored). */
ORC_API int orc_find_external_contours(const uint8_t *mask, int W, int H, int *xy, int cap_pts, int *lens, int max_contours)
{
    struct collect_ctx c = {xy, cap_pts, lens, max_contours, 0, 0};
    trace_external_contours(mask, W, H, collect_visit, &c);
    return c.n;
}
/* Fills one traced contour (n points (x, y)) into img. */
ORC_API void orc_fill_contour(const int *xy, int n, int W, int H, uint8_t *img, int value)
{
    const int PW = W + 2;
    int *chain = (int *)malloc((size_t)n * sizeof(int));
    for (int k = 0; k < n; k++) chain[k] = (xy[2 * k + 1] + 1) * PW + xy[2 * k] + 1;
    fill_chain(chain, n, PW, W, H, img, (uint8_t)value);
    free(chain);
}

/* dbg (optional): 3 planes W*H: combined mask, mask after morphology, final filled mask */
ORC_API int orc_simple_extract(const uint8_t *bgr, int W, int H, ptrdiff_t row_stride, const int *lower, const int *upper,
                               int brightness_threshold, double min_area, double *xy, int *count, uint8_t *dbg)
{
    hsv_tables();
    const size_t n = (size_t)W * H;
    uint8_t *gray = (uint8_t *)malloc(n), *m0 = (uint8_t *)malloc(n), *m1 = (uint8_t *)malloc(n), *fin = (uint8_t *)calloc(n, 1);
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++) {
            const uint8_t *px = bgr + (ptrdiff_t)y * row_stride + 3 * x;
            int hh, ss, vv;
            bgr2hsv_px(px[0], px[1], px[2], &hh, &ss, &vv);
            int g = (px[2] * 9798 + px[1] * 19235 + px[0] * 3735 + (1 << 14)) >> 15;
            gray[(size_t)y * W + x] = (uint8_t)g;
            int in = hh >= lower[0] && hh <= upper[0] && ss >= lower[1] && ss <= upper[1] && vv >= lower[2] && vv <= upper[2];
            m0[(size_t)y * W + x] = (in && g > brightness_threshold) ? 255 : 0;
        }
    if (dbg) memcpy(dbg, m0, n);
    morph3x3(m0, W, H, 1, m1); morph3x3(m1, W, H, 0, m0);   /* close = erode(dilate) */
    morph3x3(m0, W, H, 0, m1); morph3x3(m1, W, H, 1, m0);   /* open  = dilate(erode) */
    if (dbg) memcpy(dbg + n, m0, n);

    {
        struct simple_fill_ctx ctx = {W, H, min_area, fin};
        trace_external_contours(m0, W, H, simple_fill_visit, &ctx);
    }
    if (dbg) memcpy(dbg + 2 * n, fin, n);
    /* per-row intensity-weighted centroid */
    int m = 0;
    for (int y = 0; y < H; y++) {
        double sxw = 0, sw = 0;
        int any = 0;
        for (int x = 0; x < W; x++)
            if (fin[(size_t)y * W + x]) { any = 1; sxw += (double)x * gray[(size_t)y * W + x]; sw += gray[(size_t)y * W + x]; }
        if (any && sw > 0) { xy[2 * m] = sxw / sw; xy[2 * m + 1] = (double)y; m++; }
    }
    *count = m;
    free(gray); free(m0); free(m1); free(fin);
    return 0;
}

/* ------------------------------------------------------------------------ */
/* Rectification: cv2.initUndistortRectifyMap + cv2.remap(INTER_LINEAR)      */
/* (camera/single_usb_stereo_camera.py:190-206, 313-314).  Parity unpinned:  */
/* restated from OpenCV 4.x modules/calib3d/src/undistort.dispatch.cpp and   */
/* modules/imgproc/src/imgwarp.cpp.                                          */
/* ------------------------------------------------------------------------ */

/* Map of the rectified pixel (j, i) into the distorted source image.
 * K: 3x3 camera matrix (row major); dist: k1 k2 p1 p2 k3 k4 k5 k6 s1 s2 s3 s4
 * (12 doubles, unused ones 0); iR: inverse of (newCameraMatrix[:, :3] * R).
 * OpenCV's scalar loop accumulates _x += ir[0] along a row while its vector
 * paths multiply; the product form j*ir[0] + (i*ir[1] + ir[2]) is used here. */
ORC_API void orc_init_undistort_rectify_map(const double *K, const double *dist, const double *iR, int W, int H,
                                            float *mapx, float *mapy)
{
    const double u0 = K[2], v0 = K[5], fx = K[0], fy = K[4];
    const double k1 = dist[0], k2 = dist[1], p1 = dist[2], p2 = dist[3], k3 = dist[4], k4 = dist[5], k5 = dist[6], k6 = dist[7];
    const double s1 = dist[8], s2 = dist[9], s3 = dist[10], s4 = dist[11];
    for (int i = 0; i < H; i++) {
        for (int j = 0; j < W; j++) {
            const double _x = j * iR[0] + (i * iR[1] + iR[2]);
            const double _y = j * iR[3] + (i * iR[4] + iR[5]);
            const double _w = j * iR[6] + (i * iR[7] + iR[8]);
            const double w = 1. / _w, x = _x * w, y = _y * w;
            const double x2 = x * x, y2 = y * y;
            const double r2 = x2 + y2, _2xy = 2 * x * y;
            const double kr = (1 + ((k3 * r2 + k2) * r2 + k1) * r2) / (1 + ((k6 * r2 + k5) * r2 + k4) * r2);
            const double xd = (x * kr + p1 * _2xy + p2 * (r2 + 2 * x2) + s1 * r2 + s2 * r2 * r2);
            const double yd = (y * kr + p1 * (r2 + 2 * y2) + p2 * _2xy + s3 * r2 + s4 * r2 * r2);
            mapx[(size_t)i * W + j] = (float)(fx * xd + u0);
            mapy[(size_t)i * W + j] = (float)(fy * yd + v0);
        }
    }
}

/* remap of a u8 image with `ch` interleaved channels, float32 maps, bilinear, constant border 0.
 * OpenCV turns the maps into fixed point first: s = cvRound(map * 32) (round half to even), integer part
 * s >> 5 saturated to int16, fraction s & 31; the four tap weights are (32-fx)(32-fy)*32 ... fx*fy*32 (they
 * sum to 2^15 exactly, so the table's sum correction never fires) and the result is (sum + 2^14) >> 15. */
static inline int remap_fix(float m)
{
    const double v = (double)(m * 32.0f);
    if (!(v > -2147483648.0 && v < 2147483647.0)) return INT_MIN; /* NaN / overflow: cvtsd2si's "integer indefinite" */
    return (int)nearbyint(v);
}
static inline int sat_s16(int v) { return v < -32768 ? -32768 : (v > 32767 ? 32767 : v); }

ORC_API void orc_remap_bilinear_u8(const uint8_t *src, int sw, int sh, int ch, const float *mapx, const float *mapy,
                                   int dw, int dh, uint8_t *dst)
{
    for (int i = 0; i < dh; i++) {
        for (int j = 0; j < dw; j++) {
            const int sx = remap_fix(mapx[(size_t)i * dw + j]), sy = remap_fix(mapy[(size_t)i * dw + j]);
            const int fx = sx & 31, fy = sy & 31;
            const int ix = sat_s16(sx >> 5), iy = sat_s16(sy >> 5);
            const int w00 = (32 - fx) * (32 - fy) * 32, w01 = fx * (32 - fy) * 32, w10 = (32 - fx) * fy * 32, w11 = fx * fy * 32;
            for (int c = 0; c < ch; c++) {
                int p[4];
                for (int t = 0; t < 4; t++) {
                    const int x = ix + (t & 1), y = iy + (t >> 1);
                    p[t] = (x >= 0 && x < sw && y >= 0 && y < sh) ? src[((size_t)y * sw + x) * ch + c] : 0;
                }
                dst[((size_t)i * dw + j) * ch + c] = (uint8_t)((p[0] * w00 + p[1] * w01 + p[2] * w10 + p[3] * w11 + (1 << 14)) >> 15);
            }
        }
    }
}

/* ------------------------------------------------------------------------ */
/* Steger variants (improved_steger.py:39-126, 128-223, 250-344): float      */
/* GaussianBlur, Sobel (ksize 3) of Sobel for the second derivatives, the    */
/* eigenvector of the larger |eigenvalue| kept only when that eigenvalue is  */
/* negative (bright line).  Parity unpinned (OpenCV's float summation order  */
/* in Sobel and LAPACK's float32 eigenvectors are restated, not observed).   */
/* The scalar arithmetic after the Hessian follows NumPy < 2.0 (the          */
/* reference's requirement): float32 * float32 stays float32, a Python       */
/* number times a float32 scalar becomes float64.                            */
/* ------------------------------------------------------------------------ */
typedef struct { int ok; float lam; double ox, oy; } steger2_res;

static steger2_res steger2_point(float dx, float dy, float dxx, float dyy, float dxy)
{
    steger2_res r = {0, 0.0f, 0.0, 0.0};
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
        if (n1 >= n2) { const double s = 1.0 / sqrt(n1); nx = v1x * s; ny = v1y * s; }
        else { const double s = 1.0 / sqrt(n2); nx = v2x * s; ny = v2y * s; }
    }
    const float lamf = (float)lam, nxf = (float)nx, nyf = (float)ny;
    if (lamf >= 0.0f) return r;
    const float t1 = (nxf * nxf) * dxx, t3 = (nyf * nyf) * dyy;
    const double t2 = ((2.0 * (double)nxf) * (double)nyf) * (double)dxy;
    const double den = ((double)t1 + t2) + (double)t3;
    if (fabs(den) < 1e-6) return r;
    const float num = -(nxf * dx + nyf * dy);
    const double t = (double)num / den;
    r.ok = 1; r.lam = lamf; r.ox = t * (double)nxf; r.oy = t * (double)nyf;
    return r;
}

/* the five derivative planes of the reference: dx, dy = Sobel of the blurred image, dxx = Sobel_x(dx),
 * dyy = Sobel_y(dy), dxy = Sobel_y(dx); 3-tap passes summed as (a + c) + 2b, BORDER_REFLECT_101 */
static void sobel3_f32(const float *src, int W, int H, int xorder, float *tmp, float *dst)
{
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++) {
            const float l = src[(size_t)y * W + reflect101(x - 1, W)], c = src[(size_t)y * W + x], r = src[(size_t)y * W + reflect101(x + 1, W)];
            tmp[(size_t)y * W + x] = xorder ? r - l : (l + r) + c * 2.0f;
        }
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++) {
            const float u = tmp[(size_t)reflect101(y - 1, H) * W + x], c = tmp[(size_t)y * W + x], d = tmp[(size_t)reflect101(y + 1, H) * W + x];
            dst[(size_t)y * W + x] = xorder ? (u + d) + c * 2.0f : d - u;
        }
}

/* cv2.Sobel(src_f32, CV_32F, dx, dy, ksize=3) with (dx, dy) = (1, 0) or (0, 1), exported for the golden-vector generator
 * (tests/golden/make_improved_steger_golden.py), which runs the reference's own per-pixel code on these images. */
ORC_API int orc_sobel3_f32(const float *src, int W, int H, int xorder, float *dst)
{
    float *tmp = (float *)malloc((size_t)W * H * sizeof(float));
    if (!tmp) return -1;
    sobel3_f32(src, W, H, xorder, tmp, dst);
    free(tmp);
    return 0;
}

/* variant 1: every accepted pixel (raster order); variant 2: the accepted pixel of largest |eigenvalue| per row
 * (first one on ties), y = row; variant 3: as 2 but only |offset_x| <= 0.5 is tested and candidates are the
 * pixels of `mask`.  gray8: brightness image (test gray8 > threshold; ignored when mask is given).
 * src: image that is blurred (float of the gray image).  xy: doubles. */
static int steger2_core(const uint8_t *gray8, const uint8_t *mask, const float *src, int W, int H, double sigma, int threshold,
                        double resp_thr, int variant, double *xy, int capacity, int *count)
{
    float taps[512];
    int ks = orc_gauss_kernel(sigma, taps, 512);
    if (ks < 0) return -3;
    const size_t n = (size_t)W * H;
    float *bl = (float *)malloc(n * sizeof(float)), *tmp = (float *)malloc(n * sizeof(float));
    float *dx = (float *)malloc(n * sizeof(float)), *dy = (float *)malloc(n * sizeof(float));
    float *dxx = (float *)malloc(n * sizeof(float)), *dyy = (float *)malloc(n * sizeof(float)), *dxy = (float *)malloc(n * sizeof(float));
    gauss_blur_f32(src, W, H, taps, ks, bl);
    sobel3_f32(bl, W, H, 1, tmp, dx);
    sobel3_f32(bl, W, H, 0, tmp, dy);
    sobel3_f32(dx, W, H, 1, tmp, dxx);
    sobel3_f32(dy, W, H, 0, tmp, dyy);
    sobel3_f32(dx, W, H, 0, tmp, dxy);
    int cnt = 0;
    for (int y = 1; y < H - 1; y++) {
        int have = 0;
        float best_resp = -1.0f;
        double best_x = 0.0;
        for (int x = 1; x < W - 1; x++) {
            const size_t o = (size_t)y * W + x;
            if (mask ? !(mask[o] > 0) : !(gray8[o] > threshold)) continue;
            const steger2_res r = steger2_point(dx[o], dy[o], dxx[o], dyy[o], dxy[o]);
            if (!r.ok) continue;
            if (variant == 1) {
                if (fabs(r.ox) <= resp_thr && fabs(r.oy) <= resp_thr) {
                    const double cx = (double)x + r.ox, cy = (double)y + r.oy;
                    if (0 <= cx && cx < W && 0 <= cy && cy < H) {
                        if (cnt < capacity) { xy[2 * cnt] = cx; xy[2 * cnt + 1] = cy; }
                        cnt++;
                    }
                }
            } else {
                const int okk = variant == 2 ? (fabs(r.ox) <= resp_thr && fabs(r.oy) <= resp_thr) : (fabs(r.ox) <= 0.5);
                if (okk) {
                    const float resp = fabsf(r.lam);
                    if (resp > best_resp) { best_resp = resp; best_x = (double)x + r.ox; have = 1; }
                }
            }
        }
        if (variant != 1 && have && 0 <= best_x && best_x < W) {
            if (cnt < capacity) { xy[2 * cnt] = best_x; xy[2 * cnt + 1] = (double)y; }
            cnt++;
        }
    }
    *count = cnt;
    free(bl); free(tmp); free(dx); free(dy); free(dxx); free(dyy); free(dxy);
    return 0;
}

ORC_API int orc_steger_improved(const uint8_t *img, int channels, int W, int H, double sigma, int threshold, double resp_thr,
                                int per_row, double *xy, int capacity, int *count)
{
    if (channels != 1 && channels != 3) return -1;
    const size_t n = (size_t)W * H;
    uint8_t *g8 = (uint8_t *)malloc(n);
    float *gf = (float *)malloc(n * sizeof(float));
    for (size_t i = 0; i < n; i++) {
        const uint8_t *px = img + i * channels;
        g8[i] = channels == 3 ? (uint8_t)((px[2] * 9798 + px[1] * 19235 + px[0] * 3735 + (1 << 14)) >> 15) : px[0];
        gf[i] = (float)g8[i];
    }
    const int rc = steger2_core(g8, NULL, gf, W, H, sigma, threshold, resp_thr, per_row ? 2 : 1, xy, capacity, count);
    free(g8); free(gf);
    return rc;
}

/* HybridLaserExtractor.extract_centerline (improved_steger.py:250-344): colour + brightness mask, 3x3 close and
 * open, the image zeroed outside the mask, then the per-row variant on the mask's pixels.  mask_out (may be NULL)
 * receives the cleaned mask. */
ORC_API int orc_hybrid_extract(const uint8_t *bgr, int W, int H, const int *lower, const int *upper, int threshold, double sigma,
                               double *xy, int capacity, int *count, uint8_t *mask_out)
{
    hsv_tables();
    const size_t n = (size_t)W * H;
    uint8_t *m0 = (uint8_t *)malloc(n), *m1 = (uint8_t *)malloc(n);
    float *gf = (float *)malloc(n * sizeof(float));
    for (size_t i = 0; i < n; i++) {
        const uint8_t *px = bgr + 3 * i;
        int hh, ss, vv;
        bgr2hsv_px(px[0], px[1], px[2], &hh, &ss, &vv);
        const int g = (px[2] * 9798 + px[1] * 19235 + px[0] * 3735 + (1 << 14)) >> 15;
        const int in = hh >= lower[0] && hh <= upper[0] && ss >= lower[1] && ss <= upper[1] && vv >= lower[2] && vv <= upper[2];
        m0[i] = (in && g > threshold) ? 255 : 0;
    }
    morph3x3(m0, W, H, 1, m1); morph3x3(m1, W, H, 0, m0);
    morph3x3(m0, W, H, 0, m1); morph3x3(m1, W, H, 1, m0);
    for (size_t i = 0; i < n; i++) {
        const uint8_t *px = bgr + 3 * i;
        gf[i] = m0[i] ? (float)((px[2] * 9798 + px[1] * 19235 + px[0] * 3735 + (1 << 14)) >> 15) : 0.0f;
    }
    if (mask_out) memcpy(mask_out, m0, n);
    const int rc = steger2_core(NULL, m0, gf, W, H, sigma, 0, 0.5, 3, xy, capacity, count);
    free(m0); free(m1); free(gf);
    return rc;
}

/* ------------------------------------------------------------------------ */
/* Disparity-domain triangulation (improved_reconstruction.py:37-186).       */
/* PINNED: golden vectors generated from the reference itself                 */
/* (tests/golden/reconstruction_golden.json, disparity_cases*).              */
/* ------------------------------------------------------------------------ */
static double disp_at(const void *disp, int is_f64, size_t i) { return is_f64 ? ((const double *)disp)[i] : (double)((const float *)disp)[i]; }

/* Python int(round(v)) for a finite double in int range: round half to even */
static int py_round_int(double v, int *ok)
{
    *ok = (v == v) && fabs(v) < 2147483000.0;
    return *ok ? (int)nearbyint(v) : 0;
}

static int cmp_double(const void *a, const void *b) { const double x = *(const double *)a, y = *(const double *)b; return (x > y) - (x < y); }

/* window_size == 0: reconstruct_from_disparity (nearest pixel); otherwise reconstruct_with_interpolation (median of the valid
 * disparities of a window_size x window_size window).  xyz: float32 triples of the kept points, input order. */
ORC_API int orc_triangulate_disparity(const double *uv, int n, const void *disp, int is_f64, int W, int H, const double *Q,
                                      int window_size, double min_disparity, float *xyz, int *n_out)
{
    const double fx = Q[2 * 4 + 3], baseline = 1.0 / Q[3 * 4 + 2], cx = -Q[0 * 4 + 3], cy = -Q[1 * 4 + 3];
    const int hw = window_size / 2;
    double *win = (double *)malloc(sizeof(double) * (size_t)(window_size > 0 ? window_size * window_size : 1));
    int m = 0;
    for (int i = 0; i < n; i++) {
        int okx, oky;
        const int px = py_round_int(uv[2 * i], &okx), py = py_round_int(uv[2 * i + 1], &oky);
        if (!okx || !oky) continue;
        double d;
        if (window_size == 0) {
            if (px < 0 || px >= W || py < 0 || py >= H) continue;
            d = disp_at(disp, is_f64, (size_t)py * W + px);
            if (d < min_disparity || isnan(d) || isinf(d)) continue;
        } else {
            if (px < hw || px >= W - hw || py < hw || py >= H - hw) continue;
            int k = 0;
            for (int y = py - hw; y <= py + hw; y++)
                for (int x = px - hw; x <= px + hw; x++) {
                    const double v = disp_at(disp, is_f64, (size_t)y * W + x);
                    if (v >= min_disparity && !isnan(v) && !isinf(v)) win[k++] = v;
                }
            if (k == 0) continue;
            qsort(win, (size_t)k, sizeof(double), cmp_double);
            if (k & 1) d = win[k / 2];
            else if (is_f64) d = (win[k / 2 - 1] + win[k / 2]) / 2.0;
            else d = (double)(((float)win[k / 2 - 1] + (float)win[k / 2]) / 2.0f);   /* np.median of float32 averages in float32 */
        }
        const double Z = (fx * baseline) / d;
        const double X = (px - cx) * Z / fx, Y = (py - cy) * Z / fx;   /* Y is divided by fx as well: the reference's formula */
        if (Z > 0 && Z < 10.0) {
            xyz[3 * m] = (float)X; xyz[3 * m + 1] = (float)Y; xyz[3 * m + 2] = (float)Z;
            m++;
        }
    }
    free(win);
    *n_out = m;
    return 0;
}

/* create_laser_depth_map: depth (float32, [H][W], zero elsewhere) at the rounded laser pixels */
ORC_API int orc_laser_depth_map(const double *uv, int n, const void *disp, int is_f64, int W, int H, const double *Q, float *depth)
{
    const double fx = Q[2 * 4 + 3], baseline = 1.0 / Q[3 * 4 + 2];
    memset(depth, 0, sizeof(float) * (size_t)W * H);
    for (int i = 0; i < n; i++) {
        int okx, oky;
        const int px = py_round_int(uv[2 * i], &okx), py = py_round_int(uv[2 * i + 1], &oky);
        if (!okx || !oky || px < 0 || px >= W || py < 0 || py >= H) continue;
        const double d = disp_at(disp, is_f64, (size_t)py * W + px);
        if (d > 1.0 && !isnan(d)) {
            const double z = (fx * baseline) / d;
            if (0 < z && z < 10.0) depth[(size_t)py * W + px] = (float)z;
        }
    }
    return 0;
}

/* ------------------------------------------------------------------------ */
/* Point-cloud post-processing: PointCloudProcessor without Open3D            */
/* (utils/point_cloud.py:55-79, 108-132, 239-278).  PINNED: golden vectors    */
/* generated from the reference itself (tests/golden/pointcloud_golden.npz).  */
/* ------------------------------------------------------------------------ */

/* numpy's pairwise summation of a contiguous double array (np.add.reduce, hence np.mean / np.std of 1-D data) */
static double np_pairwise_sum(const double *a, size_t n)
{
    if (n < 8) {
        double res = 0.;
        for (size_t i = 0; i < n; i++) res += a[i];
        return res;
    } else if (n <= 128) {
        double r[8];
        size_t i;
        for (i = 0; i < 8; i++) r[i] = a[i];
        for (i = 8; i < n - (n % 8); i += 8)
            for (int j = 0; j < 8; j++) r[j] += a[i + j];
        double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; i++) res += a[i];
        return res;
    } else {
        size_t n2 = n / 2;
        n2 -= n2 % 8;
        return np_pairwise_sum(a, n2) + np_pairwise_sum(a + n2, n - n2);
    }
}

typedef struct { long long k[3]; int idx; } vox_rec;
static int cmp_vox(const void *a, const void *b)
{
    const vox_rec *x = (const vox_rec *)a, *y = (const vox_rec *)b;
    for (int c = 0; c < 3; c++)
        if (x->k[c] != y->k[c]) return x->k[c] < y->k[c] ? -1 : 1;
    return (x->idx > y->idx) - (x->idx < y->idx);
}
typedef struct { int head, start, count; } vox_seg;
static int cmp_seg(const void *a, const void *b) { return ((const vox_seg *)a)->head - ((const vox_seg *)b)->head; }

/* _simple_voxel_downsample: voxel = floor(p / size) per axis; one output per voxel in order of first appearance,
 * = the mean of its points accumulated in input order */
ORC_API int orc_voxel_downsample(const double *pts, int n, double voxel_size, double *out, int *n_out)
{
    vox_rec *rec = (vox_rec *)malloc(sizeof(vox_rec) * (size_t)(n > 0 ? n : 1));
    vox_seg *seg = (vox_seg *)malloc(sizeof(vox_seg) * (size_t)(n > 0 ? n : 1));
    for (int i = 0; i < n; i++) {
        for (int c = 0; c < 3; c++) rec[i].k[c] = (long long)floor(pts[3 * i + c] / voxel_size);
        rec[i].idx = i;
    }
    qsort(rec, (size_t)n, sizeof(vox_rec), cmp_vox);
    int ns = 0;
    for (int i = 0; i < n; i++) {
        if (i == 0 || rec[i].k[0] != rec[i - 1].k[0] || rec[i].k[1] != rec[i - 1].k[1] || rec[i].k[2] != rec[i - 1].k[2]) {
            seg[ns].head = rec[i].idx; seg[ns].start = i; seg[ns].count = 0;
            ns++;
        }
        seg[ns - 1].count++;
    }
    qsort(seg, (size_t)ns, sizeof(vox_seg), cmp_seg);
    for (int s = 0; s < ns; s++) {
        double acc[3] = {0., 0., 0.};
        for (int j = 0; j < seg[s].count; j++) {
            const double *p = pts + 3 * (size_t)rec[seg[s].start + j].idx;
            if (j == 0) { acc[0] = p[0]; acc[1] = p[1]; acc[2] = p[2]; }
            else { acc[0] += p[0]; acc[1] += p[1]; acc[2] += p[2]; }
        }
        for (int c = 0; c < 3; c++) out[3 * s + c] = acc[c] / (double)seg[s].count;
    }
    *n_out = ns;
    free(rec); free(seg);
    return 0;
}

/* the k smallest Euclidean distances from point i to the cloud (itself included), ascending; brute force */
static void knn_dists(const double *pts, int n, int i, int k, double *best)
{
    int m = 0;
    for (int j = 0; j < n; j++) {
        double s = 0.;
        for (int c = 0; c < 3; c++) { const double d = pts[3 * i + c] - pts[3 * j + c]; s += d * d; }
        if (m < k || s < best[m - 1]) {
            int p = m < k ? m++ : k - 1;
            while (p > 0 && best[p - 1] > s) { best[p] = best[p - 1]; p--; }
            best[p] = s;
        }
    }
    for (int j = 0; j < m; j++) best[j] = sqrt(best[j]);
}

/* _simple_outlier_removal: the predicate mean < mean + ratio * std of the nb nearest-neighbour distances */
ORC_API int orc_sor_simple(const double *pts, int n, int nb_neighbors, double std_ratio, double *out, int *n_out)
{
    if (n < nb_neighbors) { memcpy(out, pts, sizeof(double) * 3 * (size_t)n); *n_out = n; return 0; }
    const int k = nb_neighbors + 1 < n ? nb_neighbors + 1 : n;   /* cKDTree pads missing neighbours with inf */
    double *best = (double *)malloc(sizeof(double) * (size_t)(nb_neighbors + 1)), *dev = (double *)malloc(sizeof(double) * (size_t)(nb_neighbors + 1));
    int m = 0;
    for (int i = 0; i < n; i++) {
        knn_dists(pts, n, i, k, best);
        for (int j = k; j < nb_neighbors + 1; j++) best[j] = INFINITY;
        const double *d = best + 1;
        const double mean = np_pairwise_sum(d, (size_t)nb_neighbors) / (double)nb_neighbors;
        for (int j = 0; j < nb_neighbors; j++) { const double x = d[j] - mean; dev[j] = x * x; }
        const double sd = sqrt(np_pairwise_sum(dev, (size_t)nb_neighbors) / (double)nb_neighbors);
        if (mean < (mean + std_ratio * sd)) { memcpy(out + 3 * (size_t)m, pts + 3 * (size_t)i, 24); m++; }
    }
    *n_out = m;
    free(best); free(dev);
    return 0;
}

/* avg_point_spacing of compute_point_cloud_metrics: np.mean of the nearest-neighbour distances */
ORC_API double orc_avg_point_spacing(const double *pts, int n)
{
    double *nn = (double *)malloc(sizeof(double) * (size_t)n), best[2];
    for (int i = 0; i < n; i++) { knn_dists(pts, n, i, 2, best); nn[i] = best[1]; }
    const double r = np_pairwise_sum(nn, (size_t)n) / (double)n;
    free(nn);
    return r;
}
