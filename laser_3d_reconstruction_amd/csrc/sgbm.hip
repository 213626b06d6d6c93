// Semi-global block matching (cv2.STEREO_SGBM_MODE_SGBM_3WAY semantics) for gfx950.
//
// Replaces stereo_matcher.compute / right_matcher.compute of the reference
// (camera/single_usb_stereo_camera.py:252-277, 324-325).
//
// Data layout in HBM (per handle, sized for max_batch frames):
//   lrec   [B][H][W]      uint2   left image, per pixel: Sobel-clipped gradient value and its
//                                 Birchfield-Tomasi half-pixel min/max, raw intensity and its min/max
//   rrev   [B][6][H][W]   u8      the same six quantities of the right image as separate planes,
//                                 mirrored in x so that increasing disparity = increasing address
//   Tvol   [B][H+4][W1][D] s16    top->down path cost; row y of stripe n at y + n + 1, the row before a stripe's first own
//                                 row = the path state entering it (path-only scheme, the default)
//   Mvol   [B][H+4][W1]   u32     minimum over d of the same rows, in both halves
//   Cvol, Tvol [B][H][W1][D] s16  aggregated cost and path cost, stored side by side instead, only when a cost of the
//                                 top->down recurrence could saturate (block size >= 15 or P2 > 32767)
//   ckpt   [B][H][ceil(W1/4)][D] s16  left->right path state at every segment start (checkpoints)
//   wta    [B][H][W1]     u32     (biased min S << 16 | argmin) or 0xFFFFFFFF (uniqueness reject)
//   dsub   [B][H][W1]     s16     sub-pixel disparity (x16) of the column
//   raw    [B][H][W]      s16     disparity after the left-right check
//   med    [B][H][W]      s16     after the 3x3 median (speckle filter runs in place on the output)
//   labels [B][H][W]      s32     union-find parents of the speckle filter; sizes, tsizes [B][H][W] s32
//
// Work decomposition: a disparity vector of one pixel lives in a group of LPP lanes inside one
// DPP row (LPP = 8 up to 64 disparities, else 16); each lane holds 2*NP consecutive disparities
// as NP packed int16 pairs (D = 2*NP*LPP), so the d-1 / d+1 neighbours of the SGM recurrence are
// one row_shr / row_shl DPP move plus v_alignbit, the min over d is a 3- or 4-step DPP butterfly,
// and all cost arithmetic is v_pk_*_i16 with the clamp bit (the 16-bit saturating arithmetic the
// recurrence is defined in).  A 64-lane wave therefore carries 64/LPP independent pixels: image
// columns in the vertical kernel, image rows in the horizontal kernel.
#include "sgbm_dev.h"
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <new>
#include <type_traits>
#include <vector>

namespace l3d {

// ---------------------------------------------------------------------------------------------
// K1: x-Sobel prefilter + Birchfield-Tomasi half-pixel min/max for both images
// ---------------------------------------------------------------------------------------------
// Per pixel and image six quantities: the clipped x-Sobel value gc, the minimum glo and maximum ghi of gc and the two
// half-pixel means (gc + g[x-1]) / 2, (gc + g[x+1]) / 2, and the same three (rc, rlo, rhi) for the intensity.
// One wave per strip of PF_ROWS image rows x 62 groups of 4 pixels of one image (blockIdx.z = 2 * frame + side); a
// lane owns one group of 4 consecutive pixels and walks down the strip with a window of three rows in registers.
// CH = 1: gray input; CH = 3: BGR input, converted on the way in with the fixed-point weights of
// cv2.cvtColor(BGR2GRAY) (camera/single_usb_stereo_camera.py:320-321), so the fused depth entry point reads each
// colour frame once and never writes a gray plane.  Per row a lane loads its own four pixels as whole words (12 bytes
// of BGR), takes the gray words of the groups left and right of it from its neighbour lanes (lanes 0 and 63 only
// supply those), computes the clipped x-Sobel value of six pixels and the six quantities of its four, and stores them
// as two 16-byte stores (left records) or one 4-byte store per plane, mirrored in x (right image).
// No LDS and no barrier on purpose: the kernel runs beside the other half batch's scan kernels, whose blocks leave
// less than 1 KB of a CU's LDS free -- a tiled variant with 1.4 KB of LDS per block was faster alone and made the whole
// step 4 % slower, because its blocks could only start where a block of the vertical kernel had left for good.
constexpr int PF_ROWS = 8, PF_GROUPS = 62;

template <int CH>
__global__ void __launch_bounds__(256) k_prefilter(const uint8_t *__restrict__ left, const uint8_t *__restrict__ right,
                                                   int W, int H, int ftzero, uint2 *__restrict__ lrec, uint8_t *__restrict__ rrev)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int ngx = (W + 3) / 4, nwx = (ngx + PF_GROUPS - 1) / PF_GROUPS;
    const int wid = blockIdx.x * 4 + wave;                  // wave index over (row strip, strip of groups)
    const int sy = wid / nwx, sx = wid - sy * nwx;
    const int ys = sy * PF_ROWS;
    if (ys >= H) return;                                    // whole wave
    const int b = blockIdx.z >> 1, side = blockIdx.z & 1;
    const int gq = sx * PF_GROUPS + lane - 1, xq = gq * 4;  // this lane's group and its first pixel
    const size_t plane = (size_t)W * H;
    const uint8_t *img = (side == 0 ? left : right) + (size_t)b * plane * CH;
    const bool have = gq >= 0 && xq < W;                    // the group holds image pixels
    const bool whole = have && xq + 4 <= W;
    const bool owner = have && lane >= 1 && lane <= PF_GROUPS;
    // The bytes of the lane's four pixels in the PF_ROWS + 2 image rows of the strip's window (rows clamped: the Sobel
    // kernel replicates the first / last row) are requested together, before the first of them is used: per row a wave
    // is one short chain of dependent steps, and with a single row in flight it spent its time waiting for memory.
    constexpr int NW = CH == 3 ? 3 : 1;
    auto row_ptr = [&](int r) { return img + (size_t)min(max(ys - 1 + r, 0), H - 1) * W * CH; };   // r: row of the window
    uint32_t raw[PF_ROWS + 2][NW];
#pragma unroll
    for (int r = 0; r < PF_ROWS + 2; r++) {
#pragma unroll
        for (int k = 0; k < NW; k++) raw[r][k] = 0;
        if (whole) __builtin_memcpy(raw[r], row_ptr(r) + (size_t)xq * CH, 4 * NW);
    }
    // gray values of this lane's four pixels in window row r
    auto gray_word = [&](int r) -> uint32_t {
        uint32_t v = 0;
        if (whole) {
            if (CH == 3) {
                const uint32_t *w = raw[r];
                const uint32_t p0 = ((w[0] >> 16 & 0xffu) * 9798u + (w[0] >> 8 & 0xffu) * 19235u + (w[0] & 0xffu) * 3735u + (1u << 14)) >> 15;
                const uint32_t p1 = ((w[NW - 2] >> 8 & 0xffu) * 9798u + (w[NW - 2] & 0xffu) * 19235u + (w[0] >> 24) * 3735u + (1u << 14)) >> 15;
                const uint32_t p2 = ((w[NW - 1] & 0xffu) * 9798u + (w[NW - 2] >> 24) * 19235u + (w[NW - 2] >> 16 & 0xffu) * 3735u + (1u << 14)) >> 15;
                const uint32_t p3 = ((w[NW - 1] >> 24) * 9798u + (w[NW - 1] >> 16 & 0xffu) * 19235u + (w[NW - 1] >> 8 & 0xffu) * 3735u + (1u << 14)) >> 15;
                v = p0 | (p1 << 8) | (p2 << 16) | (p3 << 24);
            } else
                v = raw[r][0];
        } else if (have) {
            const uint8_t *row = row_ptr(r);
#pragma unroll
            for (int k = 0; k < 4; k++) {                   // the image's last, partial group: pixel by pixel (pixels past the edge are never used)
                const uint8_t *px = row + (size_t)min(xq + k, W - 1) * CH;
                const uint32_t q = CH == 3 ? ((uint32_t)px[2] * 9798u + (uint32_t)px[1] * 19235u + (uint32_t)px[0] * 3735u + (1u << 14)) >> 15 : px[0];
                v |= q << (8 * k);
            }
        }
        return v;
    };
    // window of three rows: [row][0..2] = gray words of the groups gq-1, gq, gq+1
    uint32_t wn[3], w0[3], ws[3];
    auto fill = [&](uint32_t (&w)[3], int r) __attribute__((always_inline)) {
        w[1] = gray_word(r);
        w[0] = __shfl_up(w[1], 1);
        w[2] = __shfl_down(w[1], 1);
    };
    fill(w0, 0);
    fill(ws, 1);
    const size_t img_off = (size_t)b * plane;
    const int yend = min(ys + PF_ROWS, H);
    // per pair of the six pixels xq - 1 .. xq + 4: 0xffff in the halves whose pixel is not an image border column
    uint32_t inner[3];
#pragma unroll
    for (int jj = 0; jj < 3; jj++) {
        const int xa = xq - 1 + 2 * jj, xb = xa + 1;
        inner[jj] = ((xa <= 0 || xa >= W - 1) ? 0u : 0xffffu) | ((xb <= 0 || xb >= W - 1) ? 0u : 0xffff0000u);
    }
    const uint32_t fz = pk_splat(ftzero), nfz = pk_splat(-ftzero);
#pragma unroll
    for (int ry = 0; ry < PF_ROWS; ry++) {
        const int y = ys + ry;
        if (y >= yend) continue;                            // whole wave
#pragma unroll
        for (int k = 0; k < 3; k++) { wn[k] = w0[k]; w0[k] = ws[k]; }
        fill(ws, ry + 2);
        if (!owner) continue;
        // Pairs of adjacent pixels as the 16-bit halves of a word: the window's bytes are widened by byte permutes and the
        // arithmetic is packed (all values lie in 0 .. 1020, nothing saturates).  Byte i of a window row is pixel xq - 4 + i.
        auto even_pair = [](const uint32_t (&w)[3], int m) -> uint32_t {   // (byte 2m, byte 2m + 1), m = 1 .. 4
            return m == 1 ? __builtin_amdgcn_perm(w[0], w[0], 0x0c030c02u) : m == 2 ? __builtin_amdgcn_perm(w[1], w[1], 0x0c010c00u)
                 : m == 3 ? __builtin_amdgcn_perm(w[1], w[1], 0x0c030c02u) : __builtin_amdgcn_perm(w[2], w[2], 0x0c010c00u);
        };
        // clipped x-Sobel value and intensity of pixels xq - 1 .. xq + 4 as three pairs; image columns 0 and W-1 (and
        // beyond) read ftzero in both channels
        uint32_t G[3], R[3];
#pragma unroll
        for (int jj = 0; jj < 3; jj++) {
            const uint32_t d0 = pk_sub(even_pair(w0, jj + 2), even_pair(w0, jj + 1));
            const uint32_t dn = pk_sub(even_pair(wn, jj + 2), even_pair(wn, jj + 1));
            const uint32_t dsv = pk_sub(even_pair(ws, jj + 2), even_pair(ws, jj + 1));
            uint32_t v = pk_add(pk_add(d0, d0), pk_add(dn, dsv));
            v = pk_add(pk_min(pk_max(v, nfz), fz), fz);
            G[jj] = (v & inner[jj]) | (fz & ~inner[jj]);
        }
        R[0] = __builtin_amdgcn_perm(w0[1], w0[0], 0x0c040c03u);   // bytes 3, 4
        R[1] = __builtin_amdgcn_perm(w0[1], w0[1], 0x0c020c01u);   // bytes 5, 6
        R[2] = __builtin_amdgcn_perm(w0[2], w0[1], 0x0c040c03u);   // bytes 7, 8
#pragma unroll
        for (int jj = 0; jj < 3; jj++) R[jj] = (R[jj] & inner[jj]) | (fz & ~inner[jj]);
        // value, half-pixel minimum and maximum of the lane's four pixels as two pairs per quantity.  (The reference takes
        // the value itself in place of the half-pixel mean at the image's first and last column; with ftzero in those
        // columns and beyond them the mean is that value anyway.)
        const v2u one = {1, 1};
        auto rec_pair = [&](uint32_t m, uint32_t c, uint32_t pn, uint32_t &lo, uint32_t &hi) __attribute__((always_inline)) {
            const uint32_t l = as_u32u(as_v2u(pk_add(c, m)) >> one), r = as_u32u(as_v2u(pk_add(c, pn)) >> one);
            lo = pku_min(c, pku_min(l, r));
            hi = pku_max(c, pku_max(l, r));
        };
        uint32_t gc[2], glo[2], ghi[2], rc[2], rlo[2], rhi[2];
#pragma unroll
        for (int q = 0; q < 2; q++) {
            gc[q] = __builtin_amdgcn_alignbit(G[q + 1], G[q], 16);
            rc[q] = __builtin_amdgcn_alignbit(R[q + 1], R[q], 16);
            rec_pair(G[q], gc[q], G[q + 1], glo[q], ghi[q]);
            rec_pair(R[q], rc[q], R[q + 1], rlo[q], rhi[q]);
        }
        if (side == 0) {
            uint2 *o = lrec + img_off + (size_t)y * W + xq;
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int q = i >> 1;
                const uint32_t s1 = (i & 1) ? 0x0c0c0602u : 0x0c0c0400u, s2 = (i & 1) ? 0x0c060100u : 0x0c040100u;
                const uint32_t wg = __builtin_amdgcn_perm(ghi[q], __builtin_amdgcn_perm(glo[q], gc[q], s1), s2);
                const uint32_t wr = __builtin_amdgcn_perm(rhi[q], __builtin_amdgcn_perm(rlo[q], rc[q], s1), s2);
                if (xq + i < W) o[i] = make_uint2(wg, wr);
            }
        } else {
            // mirrored in x: pixels xq .. xq+3 land on bytes W-1-xq down to W-4-xq
            uint8_t *o = rrev + (size_t)b * 6 * plane + (size_t)y * W + (W - 1 - xq);
            const uint32_t *qs[6] = {gc, glo, ghi, rc, rlo, rhi};
            if (xq + 4 <= W) {
#pragma unroll
                for (int jj = 0; jj < 6; jj++) {
                    const uint32_t qw = __builtin_amdgcn_perm(qs[jj][0], qs[jj][1], 0x04060002u);   // pixels 3, 2, 1, 0
                    __builtin_memcpy(o - 3 + jj * plane, &qw, 4);
                }
            } else {
#pragma unroll
                for (int i = 0; i < 4; i++)
                    if (xq + i < W) {
#pragma unroll
                        for (int jj = 0; jj < 6; jj++) (o - i)[jj * plane] = (uint8_t)(qs[jj][i >> 1] >> (16 * (i & 1)));
                    }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// K2: pixel cost -> box aggregation -> top->down path, one block per (NT/LPP columns, stripe, frame), NT threads
// ---------------------------------------------------------------------------------------------
// Birchfield-Tomasi cost of one channel for a packed pair of disparities
__device__ __forceinline__ uint32_t bt_pair(uint32_t u, uint32_t u0, uint32_t u1, uint32_t v, uint32_t v0, uint32_t v1)
{
    const uint32_t c0 = pku_max(pku_subs(u, v1), pku_subs(v0, u));
    const uint32_t c1 = pku_max(pku_subs(v, u1), pku_subs(u0, v));
    return pku_min(c0, c1);
}

// LDS footprint of the vertical kernel in u32 words: two input stages, the pixel-cost tile, the ring.
// One stage = the 6 right-image plane runs widened to 16 bits per value, twice (starting at byte 0 and at byte 1),
// + 6 words per tile column for the left pixel; `roles` = 8-byte fetches per row
__host__ __device__ inline int vertical_stage_roles(int tileC, int D) { return 6 * ((tileC - 1 + D + 3) / 4 + 1) + tileC; }
// Words between the two copies: at least the 6 widened plane runs, placed so that the reads of a half wave fall on
// different LDS banks.  A lane reads NP consecutive words, the lanes of its group the next ones (stride NP words), and
// the groups of adjacent columns start one value apart: in alternating copies, every second time one word lower.  With
// 8 lanes per pixel a half wave (32 lanes: one ds_read_b32 pass over the 32 banks) holds four groups:
//   NP = 4: a group covers every fourth bank; a displacement of 2 banks modulo 4 puts four adjacent groups on the four
//           residues modulo 4;
//   NP = 2, 6: a group covers eight of the sixteen even (or odd) banks, and the same set moved by 16 is the other eight;
//           the groups that share a parity sit in different copies, so the displacement is 16 modulo 32.
// With 16 lanes per pixel (two groups a half wave) an odd NP wants 16 modulo 32 as well and NP = 2, 6 any odd
// displacement; NP = 4 and 8 cover their banks twice whatever the displacement.  (Found by counting, per bank, the
// distinct words a half wave requests, for every displacement modulo 32; a stride of 3, 5, 7 words on 8 lanes has no
// conflict-free displacement.)  With the rule of NP = 4 applied to NP = 6 every read of the staged planes took two
// LDS cycles: two fifths of the kernel's LDS cycles at 96 disparities were bank conflicts.
// `swz`: the variant that swaps word pairs inside every second block of 32 words of a staged plane (see SWZ in the
// kernel): its planes are two words longer and its copies an odd number of words apart.
__host__ __device__ inline int vertical_copy_stride(int tileC, int D, int LPP, int NP, bool swz = false)
{
    const int w = 6 * (2 * ((tileC - 1 + D + 3) / 4 + 1) + (swz ? 2 : 0));
    if (swz) return w | 1;
    const bool want16 = LPP == 8 ? (NP == 2 || NP == 6) : (NP % 2 == 1);
    const bool want_odd = LPP == 16 && (NP == 2 || NP == 6);
    if (want16) return (w + 15) / 32 * 32 + 16;
    if (want_odd) return w | 1;
    return (w + 3) / 4 * 4 + 2;
}
__host__ __device__ inline int vertical_stage_words(int tileC, int D, int LPP, int NP, bool swz = false)
{
    return 2 * vertical_copy_stride(tileC, D, LPP, NP, swz) + 6 * tileC;
}
__host__ __device__ inline bool vertical_stage_swizzled(int LPP, int NP, bool ring_in_regs) { return ring_in_regs && LPP == 16 && NP == 4; }
__host__ __device__ inline size_t vertical_lds_words(int LPP, int NP, int SW2, int SH2, int D, bool ring_in_regs = false, int NT = 256)
{
    const int tileC = NT / LPP + 2 * SW2;
    // (the register-ring variants keep two pixel-cost tiles and need one barrier per row)
    return 2 * (size_t)vertical_stage_words(tileC, D, LPP, NP, vertical_stage_swizzled(LPP, NP, ring_in_regs)) +
           (size_t)tileC * LPP * NP * (ring_in_regs ? 2 : 1) +
           (ring_in_regs ? 0 : (size_t)(2 * SH2 + 1) * NT * NP);
}
struct __attribute__((packed, aligned(1))) UnalignedPair {
    uint32_t lo, hi;
};
constexpr int kStageLoadsMax = 3; // staging words per thread and row (host checks the bound)

// The inputs of an image row reach the lanes through LDS: the block needs tileC left records and,
// per right-image plane, one run of tileC-1+D consecutive bytes (all disparities of all its tile
// columns overlap in it).  Those ~1 KB are fetched once per row by the first few lanes of the block,
// one row ahead, instead of 2*NP bytes per plane per lane from the vector cache.  A lane's 2*NP
// bytes start at an arbitrary byte of the run and misaligned LDS reads stall the LDS pipeline, so
// the run is staged twice, starting at byte 0 and at byte 1, and already widened to one 16-bit value
// per byte (the form the packed cost arithmetic needs): every lane reads whole aligned words from
// the copy that matches the parity of its offset and uses them as they are.  The six values of the
// left pixel are staged as ready-made 16-bit splats for the same reason.
// Every global access is (block-uniform 64-bit base) + (per-lane 32-bit offset fixed for the whole
// kernel): bases advance in scalar registers and the vector ALU spends nothing on addresses.
// CSH2 > 0: a variant for block size 2*CSH2+1 that keeps the ring of box-sum rows in registers, as a shift register
// of the 2*CSH2+1 rows of the vertical window (rows above the stripe's first row are copies of it).  With block sizes
// from 7 on the ring in LDS (42 KB and more per block) leaves two blocks per CU; in registers it leaves three, and
// four for block size 7 when the kernel is held to 128 registers (a few loop-invariant values spill).
// BS2 >= 0: a variant for block size 2*BS2+1 with the LDS ring (BS2 = 2 is built for the common shapes): every LDS
// offset of the row loop is then a compile-time constant and travels in the instructions' offset fields.
// UF: the path costs carry an offset of P2 and the recurrence runs in its unsigned form (SgbmGeom::uform); the volume
// then holds T + P2 and the per-column minimum + P2.
// NT: threads per block.  512 for the register-ring variants with 16 lanes per pixel: a block of 256 threads owns 16
// columns there and computes up to 10 margin columns beside them; with 32 own columns the margin's share halves.
template <int NP, int LPP, bool PAD, int CSH2 = 0, int BS2 = -1, bool UF = false, int NT = 256>
__global__ void __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu((CSH2 == 3 || (CSH2 > 0 && LPP == 16)) ? 4 : 1)))
k_cost_vertical(SgbmGeom g, const uint2 *__restrict__ lrec, const uint8_t *__restrict__ rrev,
                                                       uint32_t *__restrict__ Cvol, uint32_t *__restrict__ Tvol, uint32_t *__restrict__ Mvol)
{
    constexpr int GPB = NT / LPP;          // image columns per block
    constexpr int CW = NP * LPP;            // u32 per volume column (= D / 2)
    // The tile has 2*SW2 margin columns beyond the block's own GPB.  Their pixel costs are computed one 32-bit word
    // (two disparities) per thread, spread over all threads of the block in rounds of NT words.  (A second
    // whole-column pass by the first lane groups, the obvious alternative, doubles the work of the first wave, and
    // the other three wait for it at the barrier.)  The first MRK rounds are unrolled with their staging index in
    // a register: all rounds of the register-ring variants, and of block sizes up to 5 otherwise.
    constexpr int MRK = (2 * (CSH2 > 0 ? CSH2 : 2) * CW + NT - 1) / NT;
    // The register-ring variants need little LDS, so they keep two pixel-cost tiles and alternate between them: the
    // barrier that separated one row's tile reads from the next row's tile writes falls away, and the inputs of a row
    // are put into LDS one row earlier, so that the one remaining barrier of a row publishes them together with the
    // tile.  The box sums of two adjacent columns share all but one of their tile columns; the two lane groups of
    // such a pair add half of the shared columns each and swap the halves (PAIRBOX).
#ifndef L3D_V_ONEBAR
#define L3D_V_ONEBAR 1
#endif
#ifndef L3D_V_PAIRBOX
#define L3D_V_PAIRBOX 1
#endif
#ifndef L3D_V_PAIRSPLIT
#define L3D_V_PAIRSPLIT 1
#endif
#ifndef L3D_V_SWZ
#define L3D_V_SWZ 1
#endif
    // SWZ (16 lanes per pixel, 4 words a lane): the 16 lanes of a group read words 4 apart, i.e. 64 words over 32 banks --
    // lanes l and l + 8 meet on a bank whatever the displacement of the copies.  Inside every second block of 32 words
    // of a staged plane the two word pairs of each group of four change places (index ^ 2): lanes 8 apart then differ by
    // 2 banks modulo 4, a group covers two residues modulo 4, and with the copies an odd number of words apart the
    // other group of the half wave covers the other two.  A lane's four words are no longer consecutive, so it keeps an
    // address for each.
    constexpr bool SWZ = CSH2 > 0 && LPP == 16 && NP == 4 && L3D_V_SWZ;
    constexpr bool ONEBAR = CSH2 > 0 && L3D_V_ONEBAR;
    constexpr bool PAIRBOX = CSH2 > 0 && L3D_V_PAIRBOX;
    extern __shared__ uint32_t smem[];
    const int tid = threadIdx.x, grp = tid / LPP, l = tid & (LPP - 1);
    const bool first = l == 0, last = l == LPP - 1;
    // logical block: (column block, stripe, frame), column blocks of one stripe adjacent on one XCD
    const dim3 blk = xcd_block();
    const int bx = blk.x, n = blk.y, b = blk.z;
    constexpr int KS2 = CSH2 > 0 ? CSH2 : BS2;   // half block size when the variant fixes it
    const int W = g.W, H = g.H, W1 = g.W1, SW2 = KS2 >= 0 ? KS2 : g.SW2, SH2 = KS2 >= 0 ? KS2 : g.SH2;
    // (the box sums of the register-ring variants take their trip count from the run-time value on purpose: with a
    // compile-time count the compiler unrolls that loop and hoists every read -- 300 to 440 registers, the kernel
    // twice as slow)
    const int SW2box = CSH2 > 0 ? g.SW2 : SW2;
    constexpr int DP = 2 * CW;              // disparities per volume column (g.Dp)
    const int start = max(min(n * g.stripe_sz - g.stripe_overlap, H), 0);
    const int end = min((n + 1) * g.stripe_sz, H);
    const int own0 = n * g.stripe_sz;
    if (start >= end) return;
    // ring of box-sum rows: the 2*SH2+1 rows of the vertical window.  The row entering the window takes the slot of
    // the row leaving it, which is therefore read first (slots are thread-private: no barrier involved).
    const int NR = 2 * SH2 + 1;
    const int tileC = GPB + 2 * SW2;
    const int nd = (tileC - 1 + DP + 3) / 4 + 1;    // u32 words per staged right-image plane
    const int nroles = 6 * nd + tileC;              // = vertical_stage_roles(tileC, D)
    const int PP = 2 * nd + (SWZ ? 2 : 0);          // words between the staged planes of a copy
    const int cstride = vertical_copy_stride(tileC, DP, LPP, NP, SWZ);
    const int stageW = 2 * cstride + 6 * tileC;     // = vertical_stage_words(tileC, D)
    auto swz = [](int t) { return SWZ ? (t ^ (((t >> 5) & 1) << 1)) : t; };   // word index inside a staged plane -> its place
    uint32_t *stage = smem;                         // [2][ 2 copies [cstride >= 6 planes [2 * nd]] | tileC left pixels [6] ]
    uint32_t *pixT0 = smem + 2 * stageW;            // [tileC][LPP][NP] (ONEBAR: two of them)
    const int tileW = tileC * LPP * NP;
    uint32_t *ring = pixT0 + tileW;                 // [NR][NT][NP], thread-private slots (LDS-ring variants only)
    const int xa = bx * GPB;
    const int xc_own = xa + grp;
    const bool active = xc_own < W1;
    const size_t plane = (size_t)W * H;
    const char *lrec_b = reinterpret_cast<const char *>(lrec + (size_t)b * plane);
    const uint8_t *rrev_b = rrev + (size_t)b * 6 * plane;
    const SgmPenalty pen = make_penalty(g);

    // image column of tile column c (replicated at the border of the cost range)
    auto tile_x = [&](int c) { return min(max(xa - SW2 + c, 0), W1 - 1) + g.minX1; };
    const int x_hi = tile_x(tileC - 1);
    const int rr_lo = W - 1 - x_hi + g.minD;        // mirrored index of the first staged right-image byte
    // staging roles of this thread: every role fetches 8 bytes -- a word of a right-image plane run and the word after
    // it (the copy that starts at byte 1 needs the first byte of the next word), or one left record
    uint32_t s_goff[kStageLoadsMax], s_lidx[kStageLoadsMax];
    bool s_left[kStageLoadsMax], s_on[kStageLoadsMax];
#pragma unroll
    for (int e = 0; e < kStageLoadsMax; e++) {
        const int i = min(tid + NT * e, nroles - 1);
        s_on[e] = tid + NT * e < nroles;
        s_left[e] = i >= 6 * nd;
        const int w = i - 6 * nd;                    // left: tile column
        const int j = i / nd, idx = i - j * nd;      // right: plane, word
        s_lidx[e] = s_left[e] ? (uint32_t)(2 * cstride + 6 * w) : (uint32_t)(j * PP + swz(2 * idx)); // the second copy lies cstride words further
        s_goff[e] = s_left[e] ? (uint32_t)tile_x(w) * (uint32_t)sizeof(uint2) : (uint32_t)((size_t)j * plane) + (uint32_t)(rr_lo + 4 * idx);
    }
    // byte offset inside a staged plane of the first disparity a lane needs for a tile column, as a word index into
    // the stage: copy (offset & 1), word (offset >> 1)
    auto stage_word = [&](int c, int lane, int wofs) {
        const int o = x_hi - tile_x(c) + lane * 2 * NP;
        return (uint32_t)((o & 1) * cstride + swz((o >> 1) + wofs));
    };
    // (SWZ: an address per word; otherwise the words follow the first)
    uint32_t qword[SWZ ? NP : 1];
#pragma unroll
    for (int w = 0; w < (SWZ ? NP : 1); w++) qword[w] = stage_word(grp, l, w);
    // margin word j = word j of the tile behind the block's own columns: (tile column GPB + j / CW, lane, word)
    const int nmargin = 2 * SW2 * CW;
    auto margin_word = [&](int j) {
        const int ch = j / CW, rem = j - ch * CW, lane = rem / NP;
        return stage_word(GPB + ch, lane, rem - lane * NP);
    };
    uint32_t mword[MRK];
#pragma unroll
    for (int r = 0; r < MRK; r++) mword[r] = margin_word(min(tid + NT * r, max(nmargin - 1, 0)));

    PadFlags<NP, PAD> pad;
    pad.init(l, g.D / 2);
    uint32_t Cst[NP], Tst[NP], hs[NP];
#pragma unroll
    for (int p = 0; p < NP; p++) { Cst[p] = 0; Tst[p] = UF ? pen.P2a : pad.initial(p); }
    uint32_t topMin = UF ? pen.P2a : 0u;
    uint32_t sreg_lo[kStageLoadsMax], sreg_hi[kStageLoadsMax];

    auto stage_fetch = [&](int k) __attribute__((always_inline)) {
        const char *lrow = lrec_b + (size_t)k * W * sizeof(uint2);
        const char *rrow = reinterpret_cast<const char *>(rrev_b + (size_t)k * W);
#pragma unroll
        for (int e = 0; e < kStageLoadsMax; e++) {
            if (e * NT < nroles && s_on[e]) {
                const UnalignedPair v = *reinterpret_cast<const UnalignedPair *>((s_left[e] ? lrow : rrow) + s_goff[e]);
                sreg_lo[e] = v.lo;
                sreg_hi[e] = v.hi;
            }
        }
    };
    auto stage_put = [&](int buf) __attribute__((always_inline)) {
        uint32_t *st = stage + buf * stageW;
#pragma unroll
        for (int e = 0; e < kStageLoadsMax; e++) {
            if (e * NT < nroles && s_on[e]) {
                const uint32_t lo = sreg_lo[e], hi = sreg_hi[e];
                uint32_t *o = st + s_lidx[e];
                if (s_left[e]) {
                    // value, half-pixel minimum and maximum of the gradient image, then of the intensity image
                    o[0] = __builtin_amdgcn_perm(lo, lo, 0x0c000c00u); o[1] = __builtin_amdgcn_perm(lo, lo, 0x0c010c01u);
                    o[2] = __builtin_amdgcn_perm(lo, lo, 0x0c020c02u); o[3] = __builtin_amdgcn_perm(hi, hi, 0x0c000c00u);
                    o[4] = __builtin_amdgcn_perm(hi, hi, 0x0c010c01u); o[5] = __builtin_amdgcn_perm(hi, hi, 0x0c020c02u);
                } else {
                    // bytes 0..3 of the word as two pairs, and the same from byte 1 on
                    o[0] = __builtin_amdgcn_perm(hi, lo, 0x0c010c00u); o[1] = __builtin_amdgcn_perm(hi, lo, 0x0c030c02u);
                    o[cstride] = __builtin_amdgcn_perm(hi, lo, 0x0c020c01u); o[cstride + 1] = __builtin_amdgcn_perm(hi, lo, 0x0c040c03u);
                }
            }
        }
    };
    // pixel costs of the row staged in `buf` -> LDS tile -> horizontal box sum for this thread's column -> ring slot.
    // Barrier A separates the previous row's tile reads from this row's tile writes (and publishes the
    // stage written since the last barrier); barrier B publishes the tile.
    auto pix_row = [&](uint32_t *pixT, int buf) __attribute__((always_inline)) {
        const uint32_t *st = stage + buf * stageW;
        {
            const uint32_t *lp = st + 2 * cstride + 6 * grp;
            const uint32_t u = lp[0], u0 = lp[1], u1 = lp[2], t = lp[3], t0 = lp[4], t1 = lp[5];
            uint32_t q[6][NP];
#pragma unroll
            for (int j = 0; j < 6; j++) {
#pragma unroll
                for (int w = 0; w < NP; w++) q[j][w] = st[j * PP + (SWZ ? qword[SWZ ? w : 0] : qword[0] + w)];
            }
            Pack<NP> px;
#pragma unroll
            for (int p = 0; p < NP; p++) {
                const uint32_t cg = bt_pair(u, u0, u1, q[0][p], q[1][p], q[2][p]);
                const uint32_t ci = bt_pair(t, t0, t1, q[3][p], q[4][p], q[5][p]);
                const v2u two = {2, 2};
                px.v[p] = cg + as_u32u(as_v2u(ci) >> two);
            }
            *reinterpret_cast<Pack<NP> *>(pixT + (grp * LPP + l) * NP) = px;
        }
        auto margin_cost = [&](int j, uint32_t mw) __attribute__((always_inline)) {
            const uint32_t *lp = st + 2 * cstride + 6 * (GPB + j / CW);
            const uint32_t u = lp[0], u0 = lp[1], u1 = lp[2], t = lp[3], t0 = lp[4], t1 = lp[5];
            uint32_t q[6];
#pragma unroll
            for (int k = 0; k < 6; k++) q[k] = st[k * PP + mw];
            const uint32_t cg = bt_pair(u, u0, u1, q[0], q[1], q[2]);
            const uint32_t ci = bt_pair(t, t0, t1, q[3], q[4], q[5]);
            const v2u two = {2, 2};
            pixT[GPB * CW + j] = cg + as_u32u(as_v2u(ci) >> two);
        };
#pragma unroll
        for (int r = 0; r < MRK; r++) {
            if (tid + NT * r < nmargin) margin_cost(tid + NT * r, mword[r]);
        }
        if (CSH2 == 0) {
#pragma unroll 1
            for (int j = tid + NT * MRK; j < nmargin; j += NT) margin_cost(j, margin_word(j));
        }
    };
    auto box_row = [&](int slot, int buf) __attribute__((always_inline)) {
        lds_barrier();
        pix_row(pixT0, buf);
        lds_barrier();
    };
    // PAIRBOX: columns 2k and 2k + 1 of the block sit in the two lane groups that the exchange below connects (the halves
    // of a DPP row with 8 lanes per pixel, two adjacent DPP rows with 16).  Their windows [c, c + 2 SW2] and
    // [c + 1, c + 1 + 2 SW2] share 2 SW2 tile columns: the even group adds the first SW2 of those, the odd group the
    // last SW2, each takes the other's partial sum, and adds the one column that is its own.  SW2 + 1 tile columns read
    // and SW2 + 1 additions per word instead of 2 SW2 + 1 and 2 SW2 (sums of 16-bit words in any order are the same).
    // Bank conflicts of these reads (8 lanes per pixel, a tile column = 8 * NP words, NP = 6: four columns apart is the same
    // bank): the four groups of a half wave must read columns that differ modulo 4.  The two even groups start two columns
    // apart and each odd group SW2 columns behind its partner, which is fine for odd SW2; for even SW2 the odd groups walk
    // their columns rotated by one (start at the second, take the first last), which needs a second base for that last read.
    const int pair_s = grp & 1;
    constexpr bool PB_ROT = LPP == 8 && CSH2 > 0 && CSH2 % 2 == 0;
    const int pb_first = grp + 1 + pair_s * (SW2 - 1);                                       // first column of this group's half
    const uint32_t pb_part = (uint32_t)(((pb_first + (PB_ROT ? pair_s : 0)) * LPP + l) * NP);
    const uint32_t pb_last = (uint32_t)(((pb_first + (PB_ROT ? pair_s * (1 - SW2) : 0)) * LPP + l) * NP);   // base of read SW2 - 1
    const uint32_t pb_own = (uint32_t)(((grp + pair_s * 2 * SW2) * LPP + l) * NP);           // the column only this window holds
    auto box_sum_pair = [&](const uint32_t *pixT) __attribute__((always_inline)) {
        constexpr int HALF = CSH2 > 0 ? CSH2 : 1;
        // (SPLIT: the tile columns are requested in two groups, which holds fewer of them in registers at a time)
        constexpr int G1 = L3D_V_PAIRSPLIT ? (HALF + 1) / 2 : HALF;
        Pack<NP> px[HALF + 1];
#pragma unroll
        for (int j = 0; j < G1; j++) px[j] = *reinterpret_cast<const Pack<NP> *>(pixT + (j == HALF - 1 ? pb_last : pb_part) + j * LPP * NP);
        uint32_t a[NP];
#pragma unroll
        for (int p = 0; p < NP; p++) {
            a[p] = px[0].v[p];
#pragma unroll
            for (int j = 1; j < G1; j++) a[p] = pk_add(a[p], px[j].v[p]);
        }
        if (L3D_V_PAIRSPLIT) {
            // the second group's reads stay behind the first group's additions
#pragma unroll
            for (int p = 0; p < NP; p++) asm volatile("" : "+v"(a[p]) : : "memory");
        }
#pragma unroll
        for (int j = G1; j < HALF; j++) px[j] = *reinterpret_cast<const Pack<NP> *>(pixT + (j == HALF - 1 ? pb_last : pb_part) + j * LPP * NP);
        px[HALF] = *reinterpret_cast<const Pack<NP> *>(pixT + pb_own);
#pragma unroll
        for (int p = 0; p < NP; p++) {
#pragma unroll
            for (int j = G1; j < HALF; j++) a[p] = pk_add(a[p], px[j].v[p]);
            const uint32_t other = LPP == 8 ? dpp_row_any<DPP_ROW_ROR8>(a[p]) : (uint32_t)__builtin_amdgcn_ds_swizzle((int)a[p], 0x401f);   // lane ^ 8 / lane ^ 16
            hs[p] = pk_add(pk_add(a[p], other), px[HALF].v[p]);
        }
    };
    auto box_sum = [&](int slot, const uint32_t *pixT) __attribute__((always_inline)) {
        if constexpr (PAIRBOX) {
            box_sum_pair(pixT);
            return;
        }
#pragma unroll
        for (int p = 0; p < NP; p++) hs[p] = 0;
        // A loop that waits for each tile column exposes one LDS round trip per column: for block size 5, the common
        // case, the five columns are read together before they are added.  (The same for the larger blocks of the
        // register-ring variants, or in predicated groups of five for any size, costs more registers than it gains.)
        if (SW2 == 2) {
            Pack<NP> px[5];
#pragma unroll
            for (int j = 0; j < 5; j++) px[j] = *reinterpret_cast<const Pack<NP> *>(pixT + ((grp + j) * LPP + l) * NP);
#pragma unroll
            for (int j = 0; j < 5; j++) {
#pragma unroll
                for (int p = 0; p < NP; p++) hs[p] = pk_add(hs[p], px[j].v[p]);
            }
        } else
            for (int j = 0; j <= 2 * SW2box; j++) {
                const Pack<NP> px = *reinterpret_cast<const Pack<NP> *>(pixT + ((grp + j) * LPP + l) * NP);
#pragma unroll
                for (int p = 0; p < NP; p++) hs[p] = pk_add(hs[p], px.v[p]);
            }
        if (CSH2 == 0) {
            Pack<NP> o;
#pragma unroll
            for (int p = 0; p < NP; p++) o.v[p] = hs[p];
            *reinterpret_cast<Pack<NP> *>(ring + (slot * NT + tid) * NP) = o;
        }
    };

    // cost of the stripe's first row: rows above it replicate that row
    constexpr int RNR = CSH2 > 0 ? 2 * CSH2 + 1 : 1;
    uint32_t rr[RNR][NP];   // CSH2 > 0: box sums of rows y-SH2 .. y+SH2, oldest first when the row loop starts, then a ring
    auto first_rows = [&](int k, int j) __attribute__((always_inline)) {
        const int kk = min(k, H - 1);
        stage_fetch(kk);
        lds_barrier();            // the previous row's stage reads are done
        stage_put(0);
        box_row(kk % NR, 0);
        box_sum(kk % NR, pixT0);
        const int scale = (k == start) ? SH2 + 1 : 1;
#pragma unroll
        for (int p = 0; p < NP; p++) Cst[p] = pk_add(Cst[p], pk_mul(hs[p], scale));
        if (CSH2 > 0) {
#pragma unroll
            for (int r = 0; r < RNR; r++)
                if (j == 0 ? r <= CSH2 : r == CSH2 + j) {
#pragma unroll
                    for (int p = 0; p < NP; p++) rr[r][p] = hs[p];
                }
        }
    };
    if (CSH2 > 0) {
#pragma unroll
        for (int j = 0; j <= CSH2; j++) first_rows(start + j, j);
    } else {
        for (int k = start; k <= start + SH2; k++) first_rows(k, 0);
    }
    if (UF) {
#pragma unroll
        for (int p = 0; p < NP; p++) Cst[p] = pk_add(Cst[p], pen.P2a);
    }

    const uint32_t voff = (uint32_t)(((size_t)xc_own * CW + l * NP) * sizeof(uint32_t));
    const size_t row_bytes = (size_t)W1 * CW * sizeof(uint32_t);
    // volume row of image row `start`: plain, or shifted by the stripe's padding rows
    const size_t vrow = g.tonly ? (size_t)b * g.HR + start + n + 1 : (size_t)b * H + start;
    const int first_stored = g.tonly ? own0 - 1 : own0;   // tonly also keeps the state entering the stripe's first own row
    char *Cout = reinterpret_cast<char *>(Cvol) + vrow * row_bytes;
    char *Tout = reinterpret_cast<char *>(Tvol) + vrow * row_bytes;
    uint32_t *Mout = Mvol + vrow * W1 + xc_own;
    // ring slots (row index mod NR) of the row entering and the row leaving the vertical window
    int slot_add = (start + 1 + SH2) % NR, slot_sub = start % NR;
    const int slot_last = (H - 1) % NR;
    // The step from row y to y + 1 needs image row y + 1 + SH2.  Its inputs are fetched into registers
    // during iteration y - 1 and put into stage (y & 1) at the top of iteration y, so they are
    // published by barrier A of iteration y; readers of that stage buffer finished two iterations
    // earlier.  Waiting for a fetch also waits for every older store, so the wait sits a whole
    // iteration after both were issued: by then they have long completed.
    // ONEBAR: row y + 1 + SH2 is fetched during iteration y - 2, put into stage (y & 1) during iteration y - 1 (published by
    // that iteration's barrier), turned into pixel costs in tile (y & 1) before the barrier of iteration y and box-summed
    // after it.  Writers of a tile or stage buffer follow its last readers by at least one barrier.
    auto row_in = [&](int y) { return (y + 1 < end) && (y + 1 + SH2 <= H - 1); };   // iteration y takes a new image row into the window
    if constexpr (ONEBAR) {
        if (row_in(start)) {
            stage_fetch(start + 1 + SH2);
            lds_barrier();                 // the reads of the first rows are done
            stage_put(start & 1);
            if (row_in(start + 1)) stage_fetch(start + 2 + SH2);
            lds_barrier();
        }
    } else if (start + 1 < end && start + 1 + SH2 <= H - 1) stage_fetch(start + 1 + SH2);
    // Register ring (CSH2 > 0): the rows are walked in groups of RNR with the group unrolled, so that "the slot of the
    // oldest row" is a compile-time index k -- the new row's sums replace the oldest row's in place and nothing is
    // shifted (a shift register cost 2 * CSH2 * NP register moves per row and lane: 54 of ~400 instructions of the
    // slowest wave at block size 9).
    // (The variants with a compile-time block size and the LDS ring walk the rows in pairs, so that which of the two
    // stage buffers a row uses is a compile-time constant as well.)
    constexpr int GRPN = CSH2 > 0 ? RNR : (BS2 >= 0 ? 2 : 1);
    for (int y0 = start; y0 < end; y0 += GRPN) {
    unroll_while<0, GRPN>([&](auto kc) __attribute__((always_inline)) -> bool {
        constexpr int k = decltype(kc)::value;
        const int y = y0 + k;
        if (y >= end) return false;
        const int buf = (CSH2 == 0 && BS2 >= 0) ? (k & 1) : (y & 1);
        const bool box_now = (y + 1 < end) && (y + 1 + SH2 <= H - 1);
        const bool fetch_next = (y + 2 < end) && (y + 2 + SH2 <= H - 1);
        if constexpr (UF) topMin = sgm_step_u<NP, LPP>(Cst, Tst, topMin, pen, first, last);
        else topMin = sgm_step<NP, LPP, PAD>(Cst, Tst, topMin, pen, first, last, pad);
        uint32_t *tile = pixT0 + (ONEBAR ? (y & 1) * tileW : 0);
        if constexpr (ONEBAR) {
            if (box_now) pix_row(tile, buf);
            if (row_in(y + 1)) stage_put(buf ^ 1);
            if (row_in(y + 2)) stage_fetch(y + 3 + SH2);
        } else {
            if (box_now) stage_put(buf);
            if (fetch_next) stage_fetch(y + 2 + SH2);
        }
        if (y >= first_stored && active) {
            Pack<NP> pc, pt;
#pragma unroll
            for (int p = 0; p < NP; p++) { pc.v[p] = Cst[p]; pt.v[p] = Tst[p]; }
            *reinterpret_cast<Pack<NP> *>(Tout + voff) = pt;
            if (g.tonly) {
                if (first) *Mout = topMin;
            } else
                *reinterpret_cast<Pack<NP> *>(Cout + voff) = pc;
        }
        Cout += row_bytes;
        Tout += row_bytes;
        Mout += W1;
        if (y + 1 < end) {
            Pack<NP> sub;
            if constexpr (CSH2 > 0) {
#pragma unroll
                for (int p = 0; p < NP; p++) sub.v[p] = rr[k][p];
            } else
                sub = *reinterpret_cast<const Pack<NP> *>(ring + (slot_sub * NT + tid) * NP);
            if (box_now) {
                if constexpr (ONEBAR) lds_barrier();
                else box_row(slot_add, buf);
                box_sum(slot_add, tile);
            } else if constexpr (CSH2 > 0) {
                // below the last image row: that row again, the newest entry of the ring
#pragma unroll
                for (int p = 0; p < NP; p++) hs[p] = rr[(k + RNR - 1) % RNR][p];
            } else {
                const Pack<NP> o = *reinterpret_cast<const Pack<NP> *>(ring + (slot_last * NT + tid) * NP);
#pragma unroll
                for (int p = 0; p < NP; p++) hs[p] = o.v[p];
            }
            if constexpr (CSH2 > 0) {
#pragma unroll
                for (int p = 0; p < NP; p++) rr[k][p] = hs[p];
            }
#pragma unroll
            for (int p = 0; p < NP; p++) Cst[p] = pk_sub(pk_add(Cst[p], hs[p]), sub.v[p]);
            slot_add = slot_add + 1 == NR ? 0 : slot_add + 1;
            if (y + 1 - SH2 > start) slot_sub = slot_sub + 1 == NR ? 0 : slot_sub + 1;
        }
        return true;
    });
    }
}

// ---------------------------------------------------------------------------------------------
// K4: single-map left-right consistency check; one block per image row.
// The row is short, so what bounds this kernel is the latency of its dependent loads (key -> right
// image column -> winner of that column).  Every thread therefore owns LR_ILP columns and issues the
// loads of all of them, to clamped addresses, before it uses any.
// ---------------------------------------------------------------------------------------------
constexpr int LR_ILP = 5;     // 320 columns a turn of the wave: rows of 640, 1280 and 1920 columns divide evenly (measured at 640: 4 / 5 / 10 -> 0.129 / 0.114 / 0.139 ms per 256 frames)
constexpr int LR_THREADS = 64; // one wave per row: measured fastest, and the barriers cost nothing

__global__ void __launch_bounds__(256) k_lrcheck(SgbmGeom g, const uint32_t *__restrict__ wta, const int16_t *__restrict__ dsub,
                                                 int16_t *__restrict__ raw)
{
    extern __shared__ uint32_t key2[]; // [W]: (cost, 0xffff - column) of the best left pixel that lands on a right column
    const int row = blockIdx.x; // over B*H
    const int W = g.W, W1 = g.W1, bd = blockDim.x, tid = threadIdx.x;
    const uint32_t *w = wta + (size_t)row * W1;
    const int16_t *ds = dsub + (size_t)row * W1;
    if (g.disp12MaxDiff > g.D + 1) {
        // The winner of a right-image column and a pixel's own disparity differ by at most D + 1, so the check cannot
        // fire (the matchers behind the disparity filter are configured this way: disp12MaxDiff = 10^6): the row is
        // the accepted sub-pixel disparities in place, without the second map and its dependent loads.
        int16_t *o = raw + (size_t)row * W;
        for (int x = tid; x < W; x += bd) {
            const bool in = x >= g.minX1 && x < g.maxX1;
            const int xc = in ? x - g.minX1 : 0;
            const uint32_t kw = w[xc];
            const int d = ds[xc];
            o[x] = (int16_t)((in && kw != 0xffffffffu) ? d : g.invalid16);
        }
        return;
    }
    for (int x = tid; x < W; x += bd) key2[x] = 0xffffffffu;
    __syncthreads();
    for (int base = 0; base < W1; base += LR_ILP * bd) {
        uint32_t k[LR_ILP];
#pragma unroll
        for (int i = 0; i < LR_ILP; i++) {
            const int xc = base + i * bd + tid;
            k[i] = w[min(xc, W1 - 1)];
            if (xc >= W1) k[i] = 0xffffffffu;
        }
#pragma unroll
        for (int i = 0; i < LR_ILP; i++) {
            const int xc = base + i * bd + tid;
            const int best = (int)(k[i] & 0xffffu);
            const uint32_t cost = k[i] >> 16;   // biased by 32768; a cost of SHRT_MAX never beats the initial SHRT_MAX
            const int x2 = xc + g.minX1 - best - g.minD;
            if (cost != 65535u && x2 >= 0 && x2 < W) atomicMin(&key2[x2], (cost << 16) | (uint32_t)(0xffff - xc));
        }
    }
    __syncthreads();
    int16_t *out = raw + (size_t)row * W;
    for (int base = 0; base < W; base += LR_ILP * bd) {
        uint32_t kw[LR_ILP];
        int d1[LR_ILP];
#pragma unroll
        for (int i = 0; i < LR_ILP; i++) {
            const int x = base + i * bd + tid;
            const bool in = x >= g.minX1 && x < g.maxX1 && x < W;
            const int xc = in ? x - g.minX1 : 0;
            kw[i] = w[xc];
            d1[i] = ds[xc];
            if (!in) kw[i] = 0xffffffffu;
        }
        // the two right-image columns a sub-pixel disparity can land on, and what was recorded there
        uint32_t ka[LR_ILP], kb[LR_ILP];
        bool ina[LR_ILP], inb[LR_ILP];
#pragma unroll
        for (int i = 0; i < LR_ILP; i++) {
            const int x = base + i * bd + tid;
            const int _x = x - (d1[i] >> 4), x_ = x - ((d1[i] + 15) >> 4);
            ina[i] = _x >= 0 && _x < W;
            inb[i] = x_ >= 0 && x_ < W;
            ka[i] = key2[ina[i] ? _x : 0];
            kb[i] = key2[inb[i] ? x_ : 0];
        }
        // The disparity of the left pixel that won a right-image column follows from the two columns (it was recorded at
        // column x2 = its own column - its disparity), so its key is not read again: one dependent load fewer per pixel.
#pragma unroll
        for (int i = 0; i < LR_ILP; i++) {
            const int x = base + i * bd + tid;
            if (x >= W) continue;
            int val = g.invalid16;
            if (kw[i] != 0xffffffffu) {
                val = d1[i];
                if (val != g.invalid16) {
                    const int _d = val >> 4, d_ = (val + 15) >> 4;
                    const int va = ka[i] == 0xffffffffu ? g.invalid16 : (0xffff - (int)(ka[i] & 0xffffu)) + g.minX1 - (x - _d);
                    const int vb = kb[i] == 0xffffffffu ? g.invalid16 : (0xffff - (int)(kb[i] & 0xffffu)) + g.minX1 - (x - d_);
                    const bool c1 = ina[i] && va >= g.minD && abs(va - _d) > g.disp12MaxDiff;
                    const bool c2 = inb[i] && vb >= g.minD && abs(vb - d_) > g.disp12MaxDiff;
                    if (c1 && c2) val = g.invalid16;
                }
            }
            out[x] = (int16_t)val;
        }
    }
}

// The same check for rows of at most TURNS * LR_ILP * LR_THREADS columns, one wave per row: every global value the wave
// needs (the keys for the second map, and per image column the key and the sub-pixel disparity again) is requested before
// the first is used, so a row costs one memory round trip and two passes over LDS instead of one round trip per turn and
// pass.  Same operations on the same values as k_lrcheck.
template <int TURNS>
__global__ void __launch_bounds__(LR_THREADS) k_lrcheck_row(SgbmGeom g, const uint32_t *__restrict__ wta, const int16_t *__restrict__ dsub,
                                                            int16_t *__restrict__ raw)
{
    extern __shared__ uint32_t key2[];
    constexpr int N = TURNS * LR_ILP;      // columns per thread
    const int row = blockIdx.x, tid = threadIdx.x;
    const int W = g.W, W1 = g.W1;
    const uint32_t *w = wta + (size_t)row * W1;
    const int16_t *ds = dsub + (size_t)row * W1;
    uint32_t k[N], kw[N];
    int d1[N];
#pragma unroll
    for (int i = 0; i < N; i++) {
        const int xc = i * LR_THREADS + tid;
        k[i] = w[min(xc, W1 - 1)];
        if (xc >= W1) k[i] = 0xffffffffu;
    }
#pragma unroll
    for (int i = 0; i < N; i++) {
        const int x = i * LR_THREADS + tid;
        const bool in = x >= g.minX1 && x < g.maxX1 && x < W;
        const int xc = in ? x - g.minX1 : 0;
        kw[i] = w[xc];
        d1[i] = ds[xc];
        if (!in) kw[i] = 0xffffffffu;
    }
    for (int x = tid; x < W; x += LR_THREADS) key2[x] = 0xffffffffu;
    __syncthreads();
#pragma unroll
    for (int i = 0; i < N; i++) {
        const int xc = i * LR_THREADS + tid;
        const int best = (int)(k[i] & 0xffffu);
        const uint32_t cost = k[i] >> 16;
        const int x2 = xc + g.minX1 - best - g.minD;
        if (cost != 65535u && x2 >= 0 && x2 < W) atomicMin(&key2[x2], (cost << 16) | (uint32_t)(0xffff - xc));
    }
    __syncthreads();
    int16_t *out = raw + (size_t)row * W;
    uint32_t ka[N], kb[N];
    uint32_t ina = 0, inb = 0;             // bit i: the column that disparity i lands on lies inside the row
#pragma unroll
    for (int i = 0; i < N; i++) {
        const int x = i * LR_THREADS + tid;
        const int _x = x - (d1[i] >> 4), x_ = x - ((d1[i] + 15) >> 4);
        const bool a_in = _x >= 0 && _x < W, b_in = x_ >= 0 && x_ < W;
        ina |= (uint32_t)a_in << i;
        inb |= (uint32_t)b_in << i;
        ka[i] = key2[a_in ? _x : 0];
        kb[i] = key2[b_in ? x_ : 0];
    }
#pragma unroll
    for (int i = 0; i < N; i++) {
        const int x = i * LR_THREADS + tid;
        if (x >= W) continue;
        int val = g.invalid16;
        if (kw[i] != 0xffffffffu) {
            val = d1[i];
            if (val != g.invalid16) {
                const int _d = val >> 4, d_ = (val + 15) >> 4;
                const int va = ka[i] == 0xffffffffu ? g.invalid16 : (0xffff - (int)(ka[i] & 0xffffu)) + g.minX1 - (x - _d);
                const int vb = kb[i] == 0xffffffffu ? g.invalid16 : (0xffff - (int)(kb[i] & 0xffffu)) + g.minX1 - (x - d_);
                const bool c1 = ((ina >> i) & 1u) && va >= g.minD && abs(va - _d) > g.disp12MaxDiff;
                const bool c2 = ((inb >> i) & 1u) && vb >= g.minD && abs(vb - d_) > g.disp12MaxDiff;
                if (c1 && c2) val = g.invalid16;
            }
        }
        out[x] = (int16_t)val;
    }
}

// ---------------------------------------------------------------------------------------------
// K5: 3x3 median of int16 with replicated border
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void srt(int &a, int &b) { const int t = min(a, b); b = max(a, b); a = t; }

// (Both median kernels also put the word at `zero`, if given, to zero: the counter of the speckle filter's kernels, which
// run next on the stream -- a separate clear would be one more dependent launch in the chain.)
__global__ void __launch_bounds__(256) k_median3(const int16_t *__restrict__ src, int W, int H, int16_t *__restrict__ dst, int *__restrict__ zero)
{
    if (zero && (blockIdx.x | blockIdx.y | blockIdx.z | threadIdx.x) == 0) *zero = 0;
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y;
    if (x >= W) return;
    const int16_t *s = src + (size_t)blockIdx.z * W * H;
    const int xm = max(x - 1, 0), xp = min(x + 1, W - 1), ym = max(y - 1, 0), yp = min(y + 1, H - 1);
    int p0 = s[(size_t)ym * W + xm], p1 = s[(size_t)ym * W + x], p2 = s[(size_t)ym * W + xp];
    int p3 = s[(size_t)y * W + xm], p4 = s[(size_t)y * W + x], p5 = s[(size_t)y * W + xp];
    int p6 = s[(size_t)yp * W + xm], p7 = s[(size_t)yp * W + x], p8 = s[(size_t)yp * W + xp];
    srt(p1, p2); srt(p4, p5); srt(p7, p8); srt(p0, p1); srt(p3, p4); srt(p6, p7);
    srt(p1, p2); srt(p4, p5); srt(p7, p8); srt(p0, p3); srt(p5, p8); srt(p4, p7);
    srt(p3, p6); srt(p1, p4); srt(p2, p5); srt(p4, p7); srt(p4, p2); srt(p6, p4);
    srt(p4, p2);
    dst[(size_t)blockIdx.z * W * H + (size_t)y * W + x] = (int16_t)p4;
}

// Even widths: one thread owns a pixel pair (one 32-bit word) of MEDIAN_ROWS consecutive rows and works on
// both pixels at once with packed 16-bit min/max.  Each input row's (left, centre, right) triple is
// sorted once and shared by the three output rows that use it; the median of nine is then
// med3(max of the lows, med3 of the middles, min of the highs).
constexpr int MEDIAN_ROWS = 4;

__device__ __forceinline__ uint32_t pk_med3(uint32_t a, uint32_t b, uint32_t c)
{
    return pk_max(pk_min(a, b), pk_min(pk_max(a, b), c));
}

__global__ void __launch_bounds__(256) k_median3_pairs(const int16_t *__restrict__ src, int W, int H, int16_t *__restrict__ dst, int *__restrict__ zero)
{
    if (zero && (blockIdx.x | blockIdx.y | blockIdx.z | threadIdx.x | threadIdx.y) == 0) *zero = 0;
    const int WP = W >> 1;
    const int xp = blockIdx.x * blockDim.x + threadIdx.x;
    if (xp >= WP) return;
    const int y0 = (blockIdx.y * blockDim.y + threadIdx.y) * MEDIAN_ROWS;
    if (y0 >= H) return;
    const uint32_t *s = reinterpret_cast<const uint32_t *>(src + (size_t)blockIdx.z * W * H);
    uint32_t *d = reinterpret_cast<uint32_t *>(dst + (size_t)blockIdx.z * W * H);
    uint32_t lo[MEDIAN_ROWS + 2], mi[MEDIAN_ROWS + 2], hi[MEDIAN_ROWS + 2];
#pragma unroll
    for (int r = 0; r < MEDIAN_ROWS + 2; r++) {
        const int y = min(max(y0 + r - 1, 0), H - 1);
        const uint32_t *p = s + (size_t)y * WP;
        const uint32_t m = p[xp];
        const uint32_t w0 = xp > 0 ? p[xp - 1] : (m << 16);      // replicated left border
        const uint32_t w2 = xp + 1 < WP ? p[xp + 1] : (m >> 16); // replicated right border
        uint32_t a = __builtin_amdgcn_alignbit(m, w0, 16);       // left neighbours of both pixels
        uint32_t b = m;
        uint32_t c = __builtin_amdgcn_alignbit(w2, m, 16);       // right neighbours
        uint32_t t;
        t = pk_min(a, b); b = pk_max(a, b); a = t;
        t = pk_min(b, c); c = pk_max(b, c); b = t;
        t = pk_min(a, b); b = pk_max(a, b); a = t;
        lo[r] = a; mi[r] = b; hi[r] = c;
    }
#pragma unroll
    for (int r = 0; r < MEDIAN_ROWS; r++) {
        if (y0 + r >= H) break;
        const uint32_t a = pk_max(pk_max(lo[r], lo[r + 1]), lo[r + 2]);
        const uint32_t b = pk_med3(mi[r], mi[r + 1], mi[r + 2]);
        const uint32_t c = pk_min(pk_min(hi[r], hi[r + 1]), hi[r + 2]);
        d[(size_t)(y0 + r) * WP + xp] = pk_med3(a, b, c);
    }
}

static void launch_median3(const int16_t *src, int W, int H, int B, int16_t *dst, hipStream_t s, int *zero = nullptr)
{
    if ((W & 1) == 0 && (((uintptr_t)src | (uintptr_t)dst) & 3) == 0)
        hipLaunchKernelGGL(k_median3_pairs, dim3(cdiv(W >> 1, 64), cdiv(H, 4 * MEDIAN_ROWS), B), dim3(64, 4), 0, s, src, W, H, dst, zero);
    else
        hipLaunchKernelGGL(k_median3, dim3(cdiv(W, 256), H, B), dim3(256), 0, s, src, W, H, dst, zero);
}

// ---------------------------------------------------------------------------------------------
// K6: speckle filter = connected components (4-neighbourhood, |a-b| <= maxDiff, pixels equal to
// newVal excluded); components of at most maxSize pixels are set to newVal.
// Only "is the component larger than maxSize" matters, so almost everything is decided inside a
// 64x32 tile in LDS:
//   - rows are split into runs with one ballot per row (no atomics, no pointer chasing); union-find
//     joins run heads of vertically adjacent rows, once per stretch of adjacent connected columns;
//   - a tile-level component that does not touch an edge shared with another tile is complete: it is
//     erased or kept on the spot.  One with more than maxSize pixels is kept whatever lies beyond;
//   - what remains (small and on a shared edge: "pending") gets global labels.  k_speckle_border
//     walks the shared edges only and joins pending components, or marks one as large when it
//     continues into a component already known to be large; k_speckle_sum adds the pixel counts of
//     the pending tile-level roots onto their final roots; k_speckle_apply erases what stayed small.
// ---------------------------------------------------------------------------------------------
constexpr int ST = 64;  // speckle tile width = wave width: one ballot describes one tile row
constexpr int STH = 32; // speckle tile height (32 rows keep the tile's LDS at 21 KB: six blocks per CU)

__device__ __forceinline__ int lds_find(volatile int *L, int i)
{
    int r = i, p;
    while ((p = L[r]) != r) r = p;
    return r;
}
__device__ __forceinline__ void lds_union(int *L, int a, int b)
{
    while (true) {
        a = lds_find(L, a);
        b = lds_find(L, b);
        if (a == b) return;
        if (a > b) { const int t = a; a = b; b = t; }
        const int old = atomicMin(L + b, a);
        if (old == b) return;
        b = old;
    }
}
__device__ __forceinline__ int uf_load(const int *L, int i) { return __hip_atomic_load(L + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ int uf_find(const int *L, int i)
{
    int r = i, p;
    while ((p = uf_load(L, r)) != r) r = p;
    return r;
}
__device__ __forceinline__ void uf_union(int *L, int a, int b)
{
    while (true) {
        a = uf_find(L, a);
        b = uf_find(L, b);
        if (a == b) return;
        if (a > b) { const int t = a; a = b; b = t; }
        const int old = atomicMin(L + b, a);
        if (old == b) return;
        b = old;
    }
}

// labels[i] = -1 for every pixel that is settled (excluded, erased, or part of a large component),
// else the global index of the pixel's tile-level root, which holds its own index.  tsize / fsize are
// written at pending roots only.  Pending roots are appended to pend_list (*pend_count entries).
__global__ void __launch_bounds__(256) k_speckle_tile(int16_t *__restrict__ disp, int W, int H, int newVal, int maxDiff, int maxSize,
                                                      int *__restrict__ labels, int *__restrict__ tsize, int *__restrict__ fsize,
                                                      int *__restrict__ pend_list, int *__restrict__ pend_count, int *__restrict__ tile_pending)
{
    __shared__ int lab[STH * ST];   // run head of a pixel; for run heads: union-find parent
    __shared__ int cnt[STH * ST];   // at roots: pixel count, bit 31 = touches a shared tile edge
    __shared__ short dv[STH * ST];
    __shared__ unsigned long long rowL[STH]; // bit x: pixel x joins its left neighbour
    __shared__ unsigned long long rowD[STH]; // bit x: pixel x joins the pixel below it
    __shared__ int plist[256];
    __shared__ int pn, pbase;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int tx0 = blockIdx.x * ST, ty0 = blockIdx.y * STH;
    const size_t img = (size_t)blockIdx.z * W * H;
    const int gx = tx0 + lane;
    if (tid == 0) pn = 0;
    const unsigned long long upto = (2ull << lane) - 1ull; // bits 0..lane

    // all 16 rows of this wave are requested before the first one is used: one memory latency, not sixteen
    int vrow[STH / 4];
#pragma unroll
    for (int rr = 0; rr < STH / 4; rr++) {
        const int gy = ty0 + wv * (STH / 4) + rr;
        vrow[rr] = (gx < W && gy < H) ? (int)disp[img + (size_t)gy * W + gx] : newVal;
    }
    // A tile whose rows are one run each and whose adjacent rows join somewhere is one component (most tiles of a dense
    // disparity map): when that component is larger than maxSize nothing in the tile can be erased and nothing is pending.
    const int wvalid = min(ST, W - tx0), hvalid = min(STH, H - ty0);
    const unsigned long long one_run = ((wvalid == 64) ? ~0ull : ((1ull << wvalid) - 1ull)) & ~1ull;
    bool plain = true;
#pragma unroll
    for (int rr = 0; rr < STH / 4; rr++) {
        const int y = wv * (STH / 4) + rr, idx = y * ST + lane;
        const int v = vrow[rr];
        const int vl = __shfl_up(v, 1);
        const bool valid = v != newVal;
        const bool joinL = valid && lane > 0 && vl != newVal && abs(v - vl) <= maxDiff;
        const unsigned long long mL = __ballot(joinL);
        if (lane == 0) rowL[y] = mL;
        if (y < hvalid) plain = plain && mL == one_run && (__ballot(valid) & 1ull);
        const int head = 63 - __clzll((long long)(~mL & upto)); // bit 0 of ~mL is always set
        dv[idx] = (short)v;
        lab[idx] = valid ? y * ST + head : -1;
        cnt[idx] = 0;
    }
    __syncthreads();
    for (int rr = 0; rr < STH / 4; rr++) {
        const int y = wv * (STH / 4) + rr, idx = y * ST + lane;
        if (y + 1 >= STH) break;
        const int v = dv[idx], vb = dv[idx + ST];
        const bool joinD = v != newVal && vb != newVal && abs(v - vb) <= maxDiff;
        const unsigned long long mD = __ballot(joinD);
        if (lane == 0) rowD[y] = mD;
        if (y + 1 < hvalid) plain = plain && mD != 0;
    }
    if (__syncthreads_and(plain) && wvalid * hvalid > maxSize) {
        // the labels of the shared edges say "settled" (the pass over the edges reads them), the tile has nothing pending
#pragma unroll
        for (int rr = 0; rr < STH / 4; rr++) {
            const int y = wv * (STH / 4) + rr, gy = ty0 + y;
            if (gx >= W || gy >= H) continue;
            const bool edge = (lane == 0 && tx0 > 0) || (lane == ST - 1 && gx + 1 < W) || (y == 0 && ty0 > 0) || (y == STH - 1 && gy + 1 < H);
            if (edge) labels[img + (size_t)gy * W + gx] = -1;
        }
        if (tid == 0) tile_pending[(blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x] = 0;
        return;
    }
    for (int rr = 0; rr < STH / 4; rr++) {
        const int y = wv * (STH / 4) + rr, idx = y * ST + lane;
        if (y + 1 >= STH) break;
        const unsigned long long mD = rowD[y];
        // the join at x is implied by the join at x-1 when both rows continue from x-1 to x
        const unsigned long long need = mD & ~((mD << 1) & rowL[y] & rowL[y + 1]);
        if ((need >> lane) & 1ull) lds_union(lab, lab[idx], lab[idx + ST]);
    }
    __syncthreads();
    for (int rr = 0; rr < STH / 4; rr++) {
        const int y = wv * (STH / 4) + rr, idx = y * ST + lane, gy = ty0 + y;
        const unsigned long long mL = rowL[y];
        const bool head = dv[idx] != newVal && !((mL >> lane) & 1ull);
        if (head) {
            const unsigned long long later = ~mL & ~upto; // run heads (or excluded pixels) right of this lane
            const int len = (later ? __ffsll((long long)later) - 1 : ST) - lane;
            const int root = lds_find(lab, idx);
            lab[idx] = root;
            atomicAdd(&cnt[root], len);
            const bool edge = (lane == 0 && tx0 > 0) || (lane + len == ST && tx0 + ST < W) || (y == 0 && ty0 > 0) || (y == STH - 1 && gy + 1 < H);
            if (edge) atomicOr(&cnt[root], (int)0x80000000);
        }
    }
    __syncthreads();
    int outv[STH / 4];
#pragma unroll
    for (int rr = 0; rr < STH / 4; rr++) {
        const int y = wv * (STH / 4) + rr, idx = y * ST + lane, gy = ty0 + y;
        outv[rr] = -1;
        if (gx >= W || gy >= H) continue;
        const size_t gi = img + (size_t)gy * W + gx;
        const int hd = lab[idx];
        if (hd >= 0) {
            const int root = lab[hd];
            const int c = cnt[root];
            const int n = c & 0x7fffffff;
            if (c >= 0) {
                if (n <= maxSize) disp[gi] = (int16_t)newVal;   // complete and small
            } else if (n <= maxSize) {                          // small so far, may continue in another tile
                const int groot = (int)(img + (size_t)(ty0 + (root >> 6)) * W + tx0 + (root & (ST - 1)));
                outv[rr] = groot;
                if (idx == root) {
                    tsize[gi] = n;
                    fsize[gi] = n;
                    plist[atomicAdd(&pn, 1)] = groot;
                }
            }
        }
    }
    __syncthreads();
    // Labels are read again only on the tile's shared edges (by the pass over the edges) and, where the tile has
    // pending components, by the pass that erases: a tile without any writes its edge pixels only.
    const bool pending = pn > 0;
#pragma unroll
    for (int rr = 0; rr < STH / 4; rr++) {
        const int y = wv * (STH / 4) + rr, gy = ty0 + y;
        if (gx >= W || gy >= H) continue;
        const bool edge = (lane == 0 && tx0 > 0) || (lane == ST - 1 && gx + 1 < W) || (y == 0 && ty0 > 0) || (y == STH - 1 && gy + 1 < H);
        if (pending || edge) labels[img + (size_t)gy * W + gx] = outv[rr];
    }
    if (tid == 0) {
        tile_pending[(blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x] = pending ? 1 : 0;
        if (pending) pbase = atomicAdd(pend_count, pn);
    }
    __syncthreads();
    if (tid < pn) pend_list[pbase + tid] = plist[tid];
}

// walks the pixel pairs that straddle an edge between two tiles
__global__ void __launch_bounds__(256) k_speckle_border(const int16_t *__restrict__ disp, int W, int H, int B, int newVal, int maxDiff, int maxSize,
                                                        int *labels, int *tsize, int *fsize)
{
    const int nv = (W - 1) / ST, nh = (H - 1) / STH; // interior tile edges
    const int per_img = nv * H + nh * W;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)per_img * B) return;
    const int b = (int)(i / per_img), k = (int)(i % per_img);
    int xa, ya, step;
    if (k < nv * H) { xa = (k / H + 1) * ST - 1; ya = k % H; step = 1; }
    else { const int k2 = k - nv * H; ya = (k2 / W + 1) * STH - 1; xa = k2 % W; step = W; }
    const size_t a = (size_t)b * W * H + (size_t)ya * W + xa, c = a + step;
    const int va = disp[a], vc = disp[c];
    if (va == newVal || vc == newVal || abs(va - vc) > maxDiff) return;
    const int la = labels[a], lc = labels[c];
    if (la < 0 && lc < 0) return;                    // two large components: nothing to decide
    if (la < 0 || lc < 0) {                          // a pending component continues into a large one
        const int t = la < 0 ? lc : la;
        atomicMax(tsize + t, maxSize + 1);
        atomicMax(fsize + t, maxSize + 1);
        return;
    }
    uf_union(labels, la, lc);
}

// every pending tile-level root adds its count to the final root of its component and points at it
__global__ void __launch_bounds__(256) k_speckle_sum(const int *__restrict__ pend_list, const int *__restrict__ pend_count, int *labels,
                                                     const int *__restrict__ tsize, int *fsize)
{
    const int n = *pend_count;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const int t = pend_list[i];
        const int r = uf_find(labels, t);
        if (r != t) {
            atomicAdd(fsize + r, tsize[t]);
            __hip_atomic_store(labels + t, r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// one block per tile; tiles without pending components have nothing to erase
__global__ void __launch_bounds__(256) k_speckle_apply(int16_t *disp, int W, int H, int newVal, int maxSize, const int *__restrict__ labels,
                                                       const int *__restrict__ fsize, const int *__restrict__ tile_pending)
{
    if (!tile_pending[(blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x]) return;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int gx = blockIdx.x * ST + lane;
    if (gx >= W) return;
    const size_t img = (size_t)blockIdx.z * W * H;
    for (int rr = 0; rr < STH / 4; rr++) {
        const int gy = blockIdx.y * STH + wv * (STH / 4) + rr;
        if (gy >= H) break;
        const size_t i = img + (size_t)gy * W + gx;
        const int t = labels[i]; // tile-level root of a pending pixel
        if (t < 0) continue;
        int r = t, p;
        while ((p = labels[r]) != r) r = p; // at most a few hops: k_speckle_sum pointed tile roots at final roots
        if (fsize[r] <= maxSize) disp[i] = (int16_t)newVal;
    }
}

// Fills the first `n` 16-bit entries of every frame's slice (stride in entries) with `v`.
__global__ void __launch_bounds__(256) k_fill16(uint16_t *base, size_t frame_stride, size_t n, uint16_t v)
{
    uint16_t *o = base + (size_t)blockIdx.y * frame_stride;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) o[i] = v;
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
int sgbm_make_geom(const l3d_sgbm_params *p, int W, int H, SgbmGeom *g)
{
    std::memset(g, 0, sizeof(*g));
    g->W = W; g->H = H;
    g->minD = p->minDisparity;
    g->D = p->numDisparities;
    const int maxD = g->minD + g->D;
    g->SW2 = g->SH2 = p->blockSize > 0 ? p->blockSize / 2 : 1;
    g->P1 = p->P1 > 0 ? p->P1 : 2;
    g->P2 = std::max(p->P2 > 0 ? p->P2 : 5, g->P1 + 1);
    g->uniq = p->uniquenessRatio >= 0 ? p->uniquenessRatio : 10;
    g->disp12MaxDiff = p->disp12MaxDiff > 0 ? p->disp12MaxDiff : 1;
    g->ftzero = std::max(p->preFilterCap, 15) | 1;
    g->minX1 = std::max(maxD, 0);
    g->maxX1 = W + std::min(g->minD, 0);
    g->W1 = g->maxX1 - g->minX1;
    g->invalid16 = (g->minD - 1) * 16;
    g->mode = p->mode;
    g->stripe_sz = (H + 3) / 4;
    g->stripe_overlap = (g->SH2 + 1) + (int)std::ceil(0.1 * g->stripe_sz);
    if (p->mode != 2) {
        // the 5- and 8-path modes aggregate over the whole image: one stripe (the other three are empty)
        g->stripe_sz = H;
        g->stripe_overlap = 0;
    }
    g->speckleWindowSize = p->speckleWindowSize;
    g->speckleRange = p->speckleRange;
    // no saturation anywhere in the top->down recurrence: 0 <= C <= 32767 and P2 <= 32767
    const long long max_c = (long long)(2 * g->SW2 + 1) * (2 * g->SH2 + 1) * (2 * g->ftzero + 63);
    g->tonly = g->P2 <= 32767 && max_c <= 32767 && p->mode == 2;
    if (const char *ov = std::getenv("L3D_SGBM_TONLY"))
        if (std::atoi(ov) == 0) g->tonly = 0;
    g->uform = 0;   // decided below, once the lane-group shape (padding disparities or not) is known
    // 8 lanes per pixel halve the wave steps per pixel and amortise the per-step overhead; both kernels gain from
    // that up to 96 disparities (measured: slower again at 128) -- the horizontal one only in the path-only scheme, where it is bound by vector
    // instructions; with both volumes it is bound by memory stalls and needs the wave count of 16 lanes per row.
    // (4 lanes per row, 16 rows per wave, measured the same as 8 in the horizontal kernel at 64 disparities: not built.)
    // L3D_SGBM_LPPV / L3D_SGBM_LPPH = 8|16 override the choice (experiments).
    auto pick = [&](const char *name, int dflt) {
        int lpp = dflt;
        if (const char *ov = std::getenv(name)) {
            const int v = std::atoi(ov);
            if ((v == 8 || v == 16) && g->D % (2 * v) == 0 && g->D / (2 * v) <= 8) lpp = v;
        }
        return lpp;
    };
    // A lane holds an even number of disparities times NP: 16 lanes take multiples of 32 disparities, 8 lanes any
    // multiple of 16 up to 128.  An odd multiple of 16 above 128 runs on 16 lanes with 16 padding disparities.
    // So does 112 with blockSize 19 or 21, whose ring of box rows does not fit the LDS with 8 lanes per pixel.
    const bool odd16 = g->D % 32 != 0;
    const bool pad = odd16 && (g->D > 128 || vertical_lds_words(8, g->D / 16, g->SW2, g->SH2, g->D) * sizeof(uint32_t) > 160 * 1024);
    const bool only8 = odd16 && !pad;
    g->Dp = pad ? g->D + 16 : g->D;
    if (g->Dp != g->D) {
        g->LPPv = g->LPPh = 16;
    } else {
        g->LPPv = pick("L3D_SGBM_LPPV", (g->D <= 96 || only8) ? 8 : 16);
        g->LPPh = pick("L3D_SGBM_LPPH", ((g->D <= 96 && g->tonly) || only8) ? 8 : 16);
    }
    g->NPv = g->Dp / (2 * g->LPPv);
    g->NPh = g->Dp / (2 * g->LPPh);
    g->HR = g->H + 4;
    // the unsigned form of the recurrence: every path cost lies in [-P2, max_c], so with P2 added it is non-negative;
    // previous minimum + P2 <= max_c + P2 and the sum of two paths <= 2 * max_c must not reach the saturation value of
    // the signed form, and 3 * P2 (the offset of the three summed paths) must be a 16-bit value
    g->uform = g->tonly && g->Dp == g->D && max_c + g->P2 <= 32767 && 2 * max_c <= 32767 && 3LL * g->P2 <= 32767;
    if (const char *ov = std::getenv("L3D_SGBM_UFORM"))
        if (std::atoi(ov) == 0) g->uform = 0;
    return 0;
}

} // namespace l3d

using namespace l3d;

// The vertical kernel has variants with the ring of box-sum rows in registers for 96 disparities on 8 lanes and 128 on
// 16 lanes, block sizes 7, 9 and 11 (the reference's own default for frames from 400 px of width is 96 / 7).
static bool vertical_ring_in_regs(const SgbmGeom &g)
{
    const bool shape = g.Dp == g.D && ((g.NPv == 6 && g.LPPv == 8) || (g.NPv == 4 && g.LPPv == 16));
    return shape && g.SH2 >= 3 && g.SH2 <= 5 && g.SW2 == g.SH2;
}
// threads per block of the vertical kernel (see its NT parameter)
static int vertical_threads(const SgbmGeom &g)
{
    if (const char *ov = std::getenv("L3D_SGBM_VERTICAL_THREADS"))
        if (std::atoi(ov) == 256) return 256;
    // (8 lanes per pixel at block size 7, the one other variant that fits 128 registers: 2.03 -> 2.09 ms per 128 pairs of
    // 640x360 with 96 disparities -- 544 cost columns are 17 blocks of 32 but 8.5 of 64 -- so it keeps 256)
    return (g.ring_regs && g.LPPv == 16) ? 512 : 256;
}

struct l3d_sgbm {
    SgbmGeom g;
    int device = 0, max_batch = 0, last_batch = 0;
    hipStream_t stream = nullptr;
    // A batch runs as two halves on two streams of the handle's own, forked from and joined to the caller's
    // stream by events: the scan kernels of one half overlap the tails and the memory-bound phases of the other
    // (measured 7 % faster at 256 frames).  Off with L3D_SGBM_SPLIT=0, below kSplitMinBatch, and while profiling.
    hipStream_t sub[2] = {nullptr, nullptr};
    hipEvent_t ev_fork = nullptr, ev_join[2] = {nullptr, nullptr};
    bool split = true;
    uint8_t *in_left = nullptr, *in_right = nullptr; // staging for the host entry point
    int16_t *out_disp = nullptr;
    uint2 *lrec = nullptr;
    uint8_t *rrev = nullptr;
    uint32_t *Cvol = nullptr, *Tvol = nullptr, *Mvol = nullptr, *Lvol = nullptr, *wta = nullptr;
    int16_t *dsub = nullptr, *raw = nullptr, *med = nullptr;
    int *labels = nullptr, *sizes = nullptr, *tsizes = nullptr, *pend = nullptr, *tflags = nullptr;
    size_t lds_vertical = 0;
    // disparity -> depth table of the fused depth entry point (65536 floats) and the Q it was built for
    float *depth_lut = nullptr;
    double lut_Q[17] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, -1.0};
    // optional per-kernel timing: a ring of event sets, one set per compute call
    static constexpr int NK = 6, RING = 32;
    bool prof = false;
    hipEvent_t ev[RING][NK + 1] = {};
    int ev_head = 0, ev_pending = 0;
    double prof_ms[NK] = {};
    long prof_calls = 0;
};

static const char *const k_sgbm_kernel_names[l3d_sgbm::NK] = {"k_prefilter", "k_cost_vertical", "k_horizontal", "k_lrcheck", "k_median3", "k_speckle_*"};

// folds the oldest pending event set into the totals (waits for it to complete)
static int sgbm_prof_drain_one(l3d_sgbm *h)
{
    const int slot = (h->ev_head - h->ev_pending + 2 * l3d_sgbm::RING) % l3d_sgbm::RING;
    L3D_HIP_CHECK(hipEventSynchronize(h->ev[slot][l3d_sgbm::NK]));
    for (int k = 0; k < l3d_sgbm::NK; k++) {
        float ms = 0;
        L3D_HIP_CHECK(hipEventElapsedTime(&ms, h->ev[slot][k], h->ev[slot][k + 1]));
        h->prof_ms[k] += ms;
    }
    h->prof_calls++;
    h->ev_pending--;
    return L3D_OK;
}

extern "C" int l3d_sgbm_geometry(const l3d_sgbm_params *params, int width, int height, int *out8)
{
    L3D_REQUIRE(params && out8, "l3d_sgbm_geometry: null argument");
    SgbmGeom g;
    sgbm_make_geom(params, width, height, &g);
    out8[0] = g.minX1; out8[1] = g.W1; out8[2] = g.stripe_sz; out8[3] = g.stripe_overlap;
    out8[4] = g.invalid16; out8[5] = g.ftzero; out8[6] = g.P1; out8[7] = g.P2;
    return L3D_OK;
}

extern "C" void l3d_sgbm_destroy(l3d_sgbm *h)
{
    if (!h) return;
    DeviceGuard dg(h->device);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    void *bufs[] = {h->in_left, h->in_right, h->out_disp, h->lrec, h->rrev, h->Cvol, h->Tvol, h->Mvol, h->Lvol,
                    h->wta, h->dsub, h->raw, h->med, h->labels, h->sizes, h->tsizes, h->pend, h->tflags, h->depth_lut};
    for (void *p : bufs)
        if (p) (void)hipFree(p);
    for (auto &set : h->ev)
        for (hipEvent_t e : set)
            if (e) (void)hipEventDestroy(e);
    for (hipStream_t st : h->sub)
        if (st) { (void)hipStreamSynchronize(st); (void)hipStreamDestroy(st); }
    if (h->ev_fork) (void)hipEventDestroy(h->ev_fork);
    for (hipEvent_t e : h->ev_join)
        if (e) (void)hipEventDestroy(e);
    if (h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
}

extern "C" int l3d_sgbm_create(const l3d_sgbm_params *p, int width, int height, int max_batch, int device, l3d_sgbm **out)
{
    L3D_REQUIRE(p && out, "l3d_sgbm_create: null argument");
    *out = nullptr;
    L3D_REQUIRE(width >= 3 && height >= 1 && max_batch >= 1, "l3d_sgbm_create: bad size %dx%d batch %d", width, height, max_batch);
    if (width > 16384) {
        // the left-right check keeps one 4-byte entry per image column in LDS (64 KiB without opting in to more)
        set_error("l3d_sgbm_create: width %d exceeds 16384", width);
        return L3D_ERR_UNSUPPORTED;
    }
    if (p->speckleWindowSize > 0 && (unsigned long long)max_batch * (unsigned long long)width * (unsigned long long)height >= (1ull << 31)) {
        // the speckle filter labels pixels with 32-bit indices over the whole batch
        set_error("l3d_sgbm_create: %d frames of %dx%d exceed the 2^31 pixels the speckle filter can label; use a smaller max_batch",
                  max_batch, width, height);
        return L3D_ERR_UNSUPPORTED;
    }
    if (p->mode < 0 || p->mode > 2) {
        set_error("l3d_sgbm_create: mode must be 0 (STEREO_SGBM_MODE_SGBM), 1 (MODE_HH) or 2 (MODE_SGBM_3WAY), got %d", p->mode);
        return L3D_ERR_UNSUPPORTED;
    }
    if (p->numDisparities <= 0 || p->numDisparities % 16 != 0 || p->numDisparities > 256) {
        set_error("l3d_sgbm_create: numDisparities must be a multiple of 16 in [16, 256], got %d", p->numDisparities);
        return L3D_ERR_UNSUPPORTED;
    }
    if (p->blockSize > 21) {
        set_error("l3d_sgbm_create: blockSize %d > 21 is not supported", p->blockSize);
        return L3D_ERR_UNSUPPORTED;
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        set_error("l3d_sgbm_create: no HIP device available");
        return L3D_ERR_NO_DEVICE;
    }
    L3D_REQUIRE(device >= 0 && device < ndev, "l3d_sgbm_create: device %d out of range (%d devices)", device, ndev);
    l3d_sgbm *h = new (std::nothrow) l3d_sgbm();
    L3D_REQUIRE(h, "l3d_sgbm_create: out of host memory");
    sgbm_make_geom(p, width, height, &h->g);
    h->device = device;
    h->max_batch = max_batch;
    // A handle for single frames (a live camera) has too few image rows to fill the chip either way; there 16 lanes
    // per row (twice the waves, a shorter step per column) give the lower latency: 0.87 -> 0.76 ms per 640x360 pair.
    if (max_batch <= 4 && h->g.Dp == h->g.D && h->g.D % 32 == 0 && h->g.LPPh == 8 && !std::getenv("L3D_SGBM_LPPH")) {
        h->g.LPPh = 16;
        h->g.NPh = h->g.Dp / 32;
    }
    // The same holds for the vertical kernel, whose blocks walk down their stripe one row after the other: with 16 lanes
    // per pixel a row is half as many instructions per wave (one frame through the reference's frame loop at 64 / 5
    // and at 96 / 7: +6 to +9 % frames per second).
    if (max_batch <= 4 && h->g.Dp == h->g.D && h->g.D % 32 == 0 && h->g.LPPv == 8 && !std::getenv("L3D_SGBM_LPPV")) {
        h->g.LPPv = 16;
        h->g.NPv = h->g.Dp / 32;
    }
    const SgbmGeom &g = h->g;
    if (g.W1 > 0 && g.stripe_overlap > g.stripe_sz) {
        delete h;
        set_error("l3d_sgbm_create: height %d too small for blockSize %d (4-stripe scheme needs overlap %d <= stripe %d)",
                  height, p->blockSize, g.stripe_overlap, g.stripe_sz);
        return L3D_ERR_UNSUPPORTED;
    }
    h->g.ring_regs = vertical_ring_in_regs(g) && !(std::getenv("L3D_SGBM_RING_REGS") && std::atoi(std::getenv("L3D_SGBM_RING_REGS")) == 0);
    const int vnt = vertical_threads(g);
    h->lds_vertical = vertical_lds_words(g.LPPv, g.NPv, g.SW2, g.SH2, g.Dp, g.ring_regs != 0, vnt) * sizeof(uint32_t);
    if (vertical_stage_roles(vnt / g.LPPv + 2 * g.SW2, g.Dp) > vnt * kStageLoadsMax || (size_t)6 * width * height >= (1ull << 31)) {
        delete h;
        set_error("l3d_sgbm_create: image %dx%d with %d disparities exceeds the input staging of the vertical kernel", width, height, g.D);
        return L3D_ERR_UNSUPPORTED;
    }
    if (h->lds_vertical > 160 * 1024) {
        delete h;
        set_error("l3d_sgbm_create: blockSize %d with numDisparities %d needs %zu B of LDS (> 160 KiB)", p->blockSize, g.D, h->lds_vertical);
        return L3D_ERR_UNSUPPORTED;
    }
    DeviceGuard dg(device);
    const size_t B = (size_t)max_batch, px = (size_t)width * height;
    const size_t vol = g.W1 > 0 ? B * height * (size_t)g.W1 * g.Dp * sizeof(int16_t) : 0;
    const size_t cols = g.W1 > 0 ? B * height * (size_t)g.W1 : 0;
    hipError_t e = hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking);
    for (int i = 0; i < 2 && e == hipSuccess; i++) e = hipStreamCreateWithFlags(&h->sub[i], hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming);
    for (int i = 0; i < 2 && e == hipSuccess; i++) e = hipEventCreateWithFlags(&h->ev_join[i], hipEventDisableTiming);
    // What the split buys is one half's short filter kernels (left-right check, median, speckle filter) running beside
    // the other half's scan kernels.  A matcher set up for the disparity filter (no left-right check, no speckle filter:
    // stereo.createDisparityWLSFilter / createRightMatcher) has almost no tail, and its two halves only get in each
    // other's way: 1280x720, 96 disparities, block size 9, 128 pairs with right matcher and WLS, two pipelines in flight:
    // 49.4 ms a step with the split, 46.9 ms without.
    h->split = g.disp12MaxDiff <= g.D + 1 || p->speckleWindowSize > 0;
    if (const char *ov = std::getenv("L3D_SGBM_SPLIT")) h->split = std::atoi(ov) != 0;
    // every buffer is cleared once here: the first write to fresh device memory pays for mapping its
    // pages, and that belongs to set-up, not to the first compute call
    auto alloc = [&](void **ptr, size_t bytes) {
        if (e == hipSuccess && bytes) e = hipMalloc(ptr, bytes);
        if (e == hipSuccess && bytes) e = hipMemsetAsync(*ptr, 0, bytes, h->stream);
    };
    alloc((void **)&h->in_left, B * px);
    alloc((void **)&h->in_right, B * px);
    alloc((void **)&h->out_disp, B * px * 2);
    alloc((void **)&h->lrec, B * px * sizeof(uint2));
    alloc((void **)&h->rrev, B * px * 6 + 1024); // the row staging reads whole words past the last needed byte
    if (g.tonly) {
        // path-only layout: no aggregated-cost volume, H + 4 rows per frame, the per-column path minimum beside it
        alloc((void **)&h->Tvol, g.W1 > 0 ? B * g.HR * (size_t)g.W1 * g.Dp * sizeof(int16_t) : 0);
        alloc((void **)&h->Mvol, g.W1 > 0 ? B * g.HR * (size_t)g.W1 * sizeof(uint32_t) : 0);
    } else {
        alloc((void **)&h->Cvol, vol);
        alloc((void **)&h->Tvol, vol);
    }
    if (g.mode == 2) alloc((void **)&h->Lvol, g.W1 > 0 ? B * height * (size_t)cdiv(g.W1, 4) * g.Dp * sizeof(int16_t) : 0);
    alloc((void **)&h->wta, cols * 4);
    alloc((void **)&h->dsub, cols * 2);
    alloc((void **)&h->raw, B * px * 2);
    alloc((void **)&h->med, B * px * 2);
    alloc((void **)&h->depth_lut, 65536 * sizeof(float));
    if (p->speckleWindowSize > 0) {
        alloc((void **)&h->labels, B * px * 4);
        alloc((void **)&h->sizes, B * px * 4);
        alloc((void **)&h->tsizes, B * px * 4);
        alloc((void **)&h->pend, (B * cdiv(width, ST) * cdiv(height, STH) * 256 + 2) * sizeof(int)); // per part of a batch: [0] = count, then the list
        alloc((void **)&h->tflags, B * cdiv(width, ST) * cdiv(height, STH) * sizeof(int));           // per tile: holds pending components
    }
    if (e == hipSuccess && g.uform && g.W1 > 0) {
        // the row above a frame's first row holds the state a path starts from: zero costs, i.e. P2 in the offset form
        const size_t row = (size_t)g.W1 * g.Dp;
        hipLaunchKernelGGL(k_fill16, dim3(64, (unsigned)B), dim3(256), 0, h->stream, (uint16_t *)h->Tvol, (size_t)g.HR * row, row, (uint16_t)g.P2);
        hipLaunchKernelGGL(k_fill16, dim3(8, (unsigned)B), dim3(256), 0, h->stream, (uint16_t *)h->Mvol, (size_t)g.HR * g.W1 * 2, (size_t)g.W1 * 2,
                           (uint16_t)g.P2);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    if (e != hipSuccess) {
        set_error("l3d_sgbm_create: allocation failed: %s", hipGetErrorString(e));
        l3d_sgbm_destroy(h);
        return L3D_ERR_HIP;
    }
    if (h->lds_vertical > 64 * 1024) {
        // opt in to more than 64 KiB of dynamic LDS per block
        const size_t lds = h->lds_vertical;
        e = hipErrorInvalidValue;
        dispatch_variant(g.NPv, g.LPPv, g.Dp != g.D, [&](auto np, auto lpp, auto pd) {
            constexpr int NP = decltype(np)::value, LPP = decltype(lpp)::value;
            constexpr bool PAD = decltype(pd)::value;
            e = hipFuncSetAttribute((const void *)k_cost_vertical<NP, LPP, PAD>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if constexpr (!PAD) {
                if (e == hipSuccess)
                    e = hipFuncSetAttribute((const void *)k_cost_vertical<NP, LPP, PAD, 0, -1, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            }
        });
        if (e != hipSuccess) {
            set_error("l3d_sgbm_create: cannot reserve %zu B of LDS: %s", h->lds_vertical, hipGetErrorString(e));
            l3d_sgbm_destroy(h);
            return L3D_ERR_HIP;
        }
    }
    *out = h;
    return L3D_OK;
}

template <int NP, int LPP, bool PAD, bool UF>
static void launch_vertical_form(l3d_sgbm *h, int B, hipStream_t s)
{
    const SgbmGeom &g = h->g;
    dim3 gv(cdiv(g.W1, 256 / LPP), 4, B);
    if constexpr (!PAD && NP == 4 && LPP == 16) {
        if (g.ring_regs && vertical_threads(g) == 512) {
            const dim3 gw(cdiv(g.W1, 512 / LPP), 4, B);
            switch (g.SH2) {
            case 3: hipLaunchKernelGGL((k_cost_vertical<NP, LPP, PAD, 3, -1, UF, 512>), gw, dim3(512), h->lds_vertical, s, g, h->lrec, h->rrev, h->Cvol, h->Tvol, h->Mvol); return;
            case 4: hipLaunchKernelGGL((k_cost_vertical<NP, LPP, PAD, 4, -1, UF, 512>), gw, dim3(512), h->lds_vertical, s, g, h->lrec, h->rrev, h->Cvol, h->Tvol, h->Mvol); return;
            case 5: hipLaunchKernelGGL((k_cost_vertical<NP, LPP, PAD, 5, -1, UF, 512>), gw, dim3(512), h->lds_vertical, s, g, h->lrec, h->rrev, h->Cvol, h->Tvol, h->Mvol); return;
            default: break;
            }
        }
    }
    if constexpr (!PAD && ((NP == 6 && LPP == 8) || (NP == 4 && LPP == 16))) {
        // the block sizes with a register-ring variant (see vertical_ring_in_regs)
        if (g.ring_regs) {
            switch (g.SH2) {
            case 3: hipLaunchKernelGGL((k_cost_vertical<NP, LPP, PAD, 3, -1, UF>), gv, dim3(256), h->lds_vertical, s, g, h->lrec, h->rrev, h->Cvol, h->Tvol, h->Mvol); return;
            case 4: hipLaunchKernelGGL((k_cost_vertical<NP, LPP, PAD, 4, -1, UF>), gv, dim3(256), h->lds_vertical, s, g, h->lrec, h->rrev, h->Cvol, h->Tvol, h->Mvol); return;
            case 5: hipLaunchKernelGGL((k_cost_vertical<NP, LPP, PAD, 5, -1, UF>), gv, dim3(256), h->lds_vertical, s, g, h->lrec, h->rrev, h->Cvol, h->Tvol, h->Mvol); return;
            default: break;
            }
        }
    }
    if constexpr (!PAD && ((NP == 4 && LPP == 8) || (NP == 6 && LPP == 8) || (NP == 4 && LPP == 16))) {
        // block size 5 (the reference's choice for small frames) at 64, 96 and 128 disparities: offsets known at compile time
        if (g.SW2 == 2 && g.SH2 == 2 && !std::getenv("L3D_SGBM_NO_BS2")) {
            hipLaunchKernelGGL((k_cost_vertical<NP, LPP, PAD, 0, 2, UF>), gv, dim3(256), h->lds_vertical, s, g, h->lrec, h->rrev, h->Cvol, h->Tvol, h->Mvol);
            return;
        }
    }
    hipLaunchKernelGGL((k_cost_vertical<NP, LPP, PAD, 0, -1, UF>), gv, dim3(256), h->lds_vertical, s, g, h->lrec, h->rrev, h->Cvol, h->Tvol, h->Mvol);
}

template <int NP, int LPP, bool PAD>
static void launch_vertical(l3d_sgbm *h, int B, hipStream_t s)
{
    if constexpr (!PAD) {
        if (h->g.uform) {
            launch_vertical_form<NP, LPP, PAD, true>(h, B, s);
            return;
        }
    }
    launch_vertical_form<NP, LPP, PAD, false>(h, B, s);
}

// 3x3 median of h->raw into d_disp, then the speckle filter in place: the tail of StereoSGBM::compute
template <typename Stamp>
static int sgbm_post(l3d_sgbm *h, int B, int16_t *d_disp, hipStream_t s, Stamp &stamp)
{
    const SgbmGeom &g = h->g;
    const bool speckle = g.speckleWindowSize > 0;
    launch_median3(h->raw, g.W, g.H, B, d_disp, s, speckle ? h->pend : nullptr);   // (and the counter of pending components to zero)
    stamp();
    if (speckle) {
        const int newVal = g.invalid16, maxDiff = 16 * g.speckleRange, maxSize = g.speckleWindowSize;
        const size_t pairs = (size_t)B * ((size_t)((g.W - 1) / ST) * g.H + (size_t)((g.H - 1) / STH) * g.W);
        hipLaunchKernelGGL(k_speckle_tile, dim3(cdiv(g.W, ST), cdiv(g.H, STH), B), dim3(256), 0, s, d_disp, g.W, g.H, newVal, maxDiff, maxSize,
                           h->labels, h->tsizes, h->sizes, h->pend + 1, h->pend, h->tflags);
        if (pairs)
            hipLaunchKernelGGL(k_speckle_border, dim3((unsigned)((pairs + 255) / 256)), dim3(256), 0, s, d_disp, g.W, g.H, B, newVal, maxDiff, maxSize,
                               h->labels, h->tsizes, h->sizes);
        hipLaunchKernelGGL(k_speckle_sum, dim3(256), dim3(256), 0, s, h->pend + 1, h->pend, h->labels, h->tsizes, h->sizes);
        hipLaunchKernelGGL(k_speckle_apply, dim3(cdiv(g.W, ST), cdiv(g.H, STH), B), dim3(256), 0, s, d_disp, g.W, g.H, newVal, maxSize, h->labels,
                           h->sizes, h->tflags);
    }
    return L3D_OK;
}

// The handle's buffers from frame `f0` on: every buffer is laid out frame by frame, so a part of a batch runs on the
// same kernels with advanced base pointers and leaves exactly what the whole batch would have left.
static l3d_sgbm sgbm_view(const l3d_sgbm *h, int f0, int part)
{
    l3d_sgbm v = *h;
    const SgbmGeom &g = h->g;
    const size_t f = (size_t)f0, px = (size_t)g.W * g.H, cols = (size_t)g.H * g.W1, colw = (size_t)g.Dp / 2;
    v.prof = false;
    v.lrec += f * px;
    v.rrev += f * 6 * px;
    if (g.tonly) {
        v.Tvol += f * g.HR * g.W1 * colw;
        v.Mvol += f * g.HR * g.W1;
    } else {
        v.Cvol += f * cols * colw;
        v.Tvol += f * cols * colw;
    }
    if (v.Lvol) v.Lvol += f * g.H * (size_t)cdiv(g.W1, 4) * colw;
    v.wta += f * cols;
    v.dsub += f * cols;
    v.raw += f * px;
    if (v.labels) {
        v.labels += f * px; v.sizes += f * px; v.tsizes += f * px;
        v.pend += f * cdiv(g.W, ST) * cdiv(g.H, STH) * 256 + part;   // each part: [0] = count, then its list
        v.tflags += f * cdiv(g.W, ST) * cdiv(g.H, STH);
    }
    return v;
}

// prefilter ... speckle filter of B frames on stream s; `stamp` marks the kernel boundaries for the per-kernel timing
// (channels = 3: the inputs are BGR frames, converted to gray inside the prefilter)
template <typename Stamp>
static int sgbm_run_part(l3d_sgbm *h, const uint8_t *d_left, const uint8_t *d_right, int channels, int B, int16_t *d_disp, hipStream_t s, Stamp &stamp)
{
    const SgbmGeom &g = h->g;
    stamp();
    const dim3 gp(cdiv(cdiv(g.H, PF_ROWS) * cdiv(cdiv(g.W, 4), PF_GROUPS), 4), 1, 2 * B);   // four waves (strips) per block
    if (channels == 3)
        hipLaunchKernelGGL(k_prefilter<3>, gp, dim3(256), 0, s, d_left, d_right, g.W, g.H, g.ftzero, h->lrec, h->rrev);
    else
        hipLaunchKernelGGL(k_prefilter<1>, gp, dim3(256), 0, s, d_left, d_right, g.W, g.H, g.ftzero, h->lrec, h->rrev);
    stamp();
    const bool have_v = dispatch_variant(g.NPv, g.LPPv, g.Dp != g.D, [&](auto np, auto lpp, auto pd) { launch_vertical<decltype(np)::value, decltype(lpp)::value, decltype(pd)::value>(h, B, s); });
    stamp();
    bool have_h = true;
    if (g.mode == 2)
        have_h = sgbm_launch_horizontal(g, B, h->Cvol, h->Tvol, h->Mvol, h->Lvol, h->wta, h->dsub, s);
    else if (have_v) {
        // the vertical kernel left the aggregated cost in Cvol (and its top->down path in Tvol, which the first path pass overwrites)
        const int rc = sgbm_paths_aggregate(g, g.mode, B, h->Cvol, h->Tvol, h->wta, h->dsub, s);
        if (rc) return rc;
    }
    if (!have_v || !have_h) {
        set_error("sgbm: no kernel variant for %d disparities", g.D);
        return L3D_ERR_UNSUPPORTED;
    }
    stamp();
    {
        // rows that fit into a few turns of one wave take the variant with all loads in flight (640 columns: 2 turns)
        const int turns = cdiv(std::max(g.W, g.W1), LR_ILP * LR_THREADS);
        const bool checks = g.disp12MaxDiff <= g.D + 1;   // otherwise k_lrcheck only copies
        const dim3 grid(B * g.H), block(LR_THREADS);
        const size_t lds = (size_t)g.W * 4;
        if (checks && turns <= 2) hipLaunchKernelGGL(k_lrcheck_row<2>, grid, block, lds, s, g, h->wta, h->dsub, h->raw);
        else if (checks && turns <= 4) hipLaunchKernelGGL(k_lrcheck_row<4>, grid, block, lds, s, g, h->wta, h->dsub, h->raw);
        else if (checks && turns <= 6) hipLaunchKernelGGL(k_lrcheck_row<6>, grid, block, lds, s, g, h->wta, h->dsub, h->raw);
        else hipLaunchKernelGGL(k_lrcheck, grid, block, lds, s, g, h->wta, h->dsub, h->raw);
    }
    stamp();
    {
        const int rc = sgbm_post(h, B, d_disp, s, stamp);
        if (rc) return rc;
    }
    stamp();
    return L3D_OK;
}

constexpr int kSplitMinBatch = 8;

// `before(f0, nb, stream)` / `after(f0, nb, stream)` enqueue what a caller wants around the matcher for frames
// f0 .. f0+nb-1 on the stream that part runs on (the fused depth entry point converts to gray before and reprojects
// after, so that these stages of one half overlap the other half's scan kernels too); they return a status.
template <typename Before, typename After>
static int sgbm_run(l3d_sgbm *h, const uint8_t *d_left, const uint8_t *d_right, int channels, int B, int16_t *d_disp, hipStream_t s,
                    Before &&before, After &&after)
{
    const SgbmGeom &g = h->g;
    const size_t px = (size_t)g.W * g.H;
    const size_t n = px * B;
    h->last_batch = B;
    if (g.W1 <= 0) {
        // no column has a full disparity range: the whole map is invalid
        // (median of a constant image and the speckle filter leave it unchanged)
        std::vector<int16_t> fill(n, (int16_t)g.invalid16);
        L3D_HIP_CHECK(hipMemcpyAsync(d_disp, fill.data(), n * 2, hipMemcpyHostToDevice, s));
        L3D_HIP_CHECK(hipStreamSynchronize(s));
        return L3D_OK;
    }
    if (h->split && !h->prof && B >= kSplitMinBatch) {
        // fork: both halves start when the caller's stream gets here; join: the caller's stream goes on after both
        L3D_HIP_CHECK(hipEventRecord(h->ev_fork, s));
        // (more parts on the two streams, or three and four streams, measured slower: 4 parts +0.7 %, 8 parts +5 %,
        // 3 streams +0.3 %, 4 streams +2 %)
        constexpr int nchunk = 2;
        auto no_stamp = []() {};
        for (int i = 0; i < 2; i++) L3D_HIP_CHECK(hipStreamWaitEvent(h->sub[i], h->ev_fork, 0));
        int rc = L3D_OK;
        for (int c = 0, f0 = 0; c < nchunk && rc == L3D_OK; c++) {
            const int nb = (B * (c + 1)) / nchunk - (B * c) / nchunk;
            l3d_sgbm v = sgbm_view(h, f0, c);
            rc = before(f0, nb, h->sub[c & 1]);
            if (rc == L3D_OK) rc = sgbm_run_part(&v, d_left + f0 * px * channels, d_right + f0 * px * channels, channels, nb, d_disp + f0 * px, h->sub[c & 1], no_stamp);
            if (rc == L3D_OK) rc = after(f0, nb, h->sub[c & 1]);
            f0 += nb;
        }
        // the join happens whatever the outcome: the caller's stream must not run ahead of work already enqueued
        for (int i = 0; i < 2; i++) L3D_HIP_CHECK(hipEventRecord(h->ev_join[i], h->sub[i]));
        for (int i = 0; i < 2; i++) L3D_HIP_CHECK(hipStreamWaitEvent(s, h->ev_join[i], 0));
        if (rc) return rc;
        L3D_HIP_CHECK(hipGetLastError());
        return L3D_OK;
    }
    int slot = -1;
    if (h->prof) {
        if (h->ev_pending == l3d_sgbm::RING) {
            int rc = sgbm_prof_drain_one(h);
            if (rc) return rc;
        }
        slot = h->ev_head;
        h->ev_head = (h->ev_head + 1) % l3d_sgbm::RING;
        h->ev_pending++;
    }
    int mark = 0;
    auto stamp = [&]() {
        if (slot >= 0) (void)hipEventRecord(h->ev[slot][mark++], s);
    };
    int rc = before(0, B, s);
    if (rc == L3D_OK) rc = sgbm_run_part(h, d_left, d_right, channels, B, d_disp, s, stamp);
    if (rc == L3D_OK) rc = after(0, B, s);
    if (rc) return rc;
    L3D_HIP_CHECK(hipGetLastError());
    return L3D_OK;
}

static int sgbm_run(l3d_sgbm *h, const uint8_t *d_left, const uint8_t *d_right, int B, int16_t *d_disp, hipStream_t s)
{
    auto nothing = [](int, int, hipStream_t) { return (int)L3D_OK; };
    return sgbm_run(h, d_left, d_right, 1, B, d_disp, s, nothing, nothing);
}

extern "C" int l3d_sgbm_compute_device(l3d_sgbm *h, const uint8_t *d_left, const uint8_t *d_right, int batch, int16_t *d_disp, void *stream)
{
    L3D_REQUIRE(h && d_left && d_right && d_disp, "l3d_sgbm_compute_device: null argument");
    L3D_REQUIRE(batch >= 1 && batch <= h->max_batch, "l3d_sgbm_compute_device: batch %d outside [1, %d]", batch, h->max_batch);
    DeviceGuard dg(h->device);
    hipStream_t s = (hipStream_t)stream;
    return sgbm_run(h, d_left, d_right, batch, d_disp, s);
}

extern "C" int l3d_sgbm_compute_depth_device(l3d_sgbm *h, const uint8_t *d_left_bgr, const uint8_t *d_right_bgr, int batch,
                                             const double *Q16_host, int16_t *d_disp, float *d_depth, void *stream)
{
    L3D_REQUIRE(h && d_left_bgr && d_right_bgr && d_depth, "l3d_sgbm_compute_depth_device: null argument");
    L3D_REQUIRE(batch >= 1 && batch <= h->max_batch, "l3d_sgbm_compute_depth_device: batch %d outside [1, %d]", batch, h->max_batch);
    DeviceGuard dg(h->device);
    const SgbmGeom &g = h->g;
    const size_t px = (size_t)g.W * g.H;
    int16_t *disp = d_disp ? d_disp : h->out_disp;
    // Z by table look-up where it depends on the disparity alone (the table is rebuilt on the caller's stream, ahead of
    // the fork, only when Q changes), else the general kernel
    const bool lut = depth_from_disp16_lut(disp, Q16_host, 0, d_depth, h->depth_lut, h->lut_Q, (hipStream_t)stream) == L3D_OK;
    auto depth = [&](int f0, int nb, hipStream_t st) {
        if (lut) return depth_from_disp16_lut(disp + f0 * px, Q16_host, px * nb, d_depth + f0 * px, h->depth_lut, h->lut_Q, st);
        return l3d_depth_from_disp16_device(disp + f0 * px, Q16_host, g.W, g.H, nb, d_depth + f0 * px, nullptr, h->device, st);
    };
    auto nothing = [](int, int, hipStream_t) { return (int)L3D_OK; };
    if (g.W1 <= 0) {
        // no column has a full disparity range: the matcher fills the map on the caller's stream, the rest follows there
        int rc = sgbm_run(h, d_left_bgr, d_right_bgr, 3, batch, disp, (hipStream_t)stream, nothing, nothing);
        if (rc == L3D_OK) rc = depth(0, batch, (hipStream_t)stream);
        return rc;
    }
    // The prefilter reads the colour frames and converts them on the way in (no gray plane is written); the
    // reprojection runs with each half of the batch on the handle's streams.
    return sgbm_run(h, d_left_bgr, d_right_bgr, 3, batch, disp, (hipStream_t)stream, nothing, depth);
}

extern "C" int l3d_sgbm_compute(l3d_sgbm *h, const uint8_t *left, const uint8_t *right, int batch, ptrdiff_t row_stride,
                                ptrdiff_t img_stride, int16_t *disp)
{
    L3D_REQUIRE(h && left && right && disp, "l3d_sgbm_compute: null argument");
    L3D_REQUIRE(batch >= 1 && batch <= h->max_batch, "l3d_sgbm_compute: batch %d outside [1, %d]", batch, h->max_batch);
    const SgbmGeom &g = h->g;
    L3D_REQUIRE(row_stride >= g.W, "l3d_sgbm_compute: row_stride %td < width %d", row_stride, g.W);
    DeviceGuard dg(h->device);
    hipStream_t s = h->stream;
    const size_t px = (size_t)g.W * g.H;
    for (int b = 0; b < batch; b++) {
        L3D_HIP_CHECK(hipMemcpy2DAsync(h->in_left + b * px, g.W, left + b * img_stride, row_stride, g.W, g.H, hipMemcpyHostToDevice, s));
        L3D_HIP_CHECK(hipMemcpy2DAsync(h->in_right + b * px, g.W, right + b * img_stride, row_stride, g.W, g.H, hipMemcpyHostToDevice, s));
    }
    int rc = sgbm_run(h, h->in_left, h->in_right, batch, h->out_disp, s);
    if (rc) return rc;
    L3D_HIP_CHECK(hipMemcpyAsync(disp, h->out_disp, px * batch * 2, hipMemcpyDeviceToHost, s));
    L3D_HIP_CHECK(hipStreamSynchronize(s));
    return L3D_OK;
}

extern "C" int l3d_sgbm_postfilter(l3d_sgbm *h, const int16_t *raw, int batch, int16_t *disp)
{
    L3D_REQUIRE(h && raw && disp, "l3d_sgbm_postfilter: null argument");
    L3D_REQUIRE(batch >= 1 && batch <= h->max_batch, "l3d_sgbm_postfilter: batch %d outside [1, %d]", batch, h->max_batch);
    DeviceGuard dg(h->device);
    const size_t bytes = (size_t)h->g.W * h->g.H * batch * sizeof(int16_t);
    L3D_HIP_CHECK(hipMemcpyAsync(h->raw, raw, bytes, hipMemcpyHostToDevice, h->stream));
    auto no_stamp = []() {};
    const int rc = sgbm_post(h, batch, h->out_disp, h->stream, no_stamp);
    if (rc) return rc;
    L3D_HIP_CHECK(hipGetLastError());
    L3D_HIP_CHECK(hipMemcpyAsync(disp, h->out_disp, bytes, hipMemcpyDeviceToHost, h->stream));
    L3D_HIP_CHECK(hipStreamSynchronize(h->stream));
    return L3D_OK;
}

// Host-side rebuild of the stage volumes [B][H][W1][D] from the device layout (stage-level checks only).
// Path-only layout: T rows are gathered out of the volume with its stripe padding rows; C = T - (what the
// recurrence added from the row above), in plain integers.  Columns are Dp wide on the device, D here.
static void stage_unpack(const SgbmGeom &g, std::vector<int16_t> &dev, int B, int stage, int16_t *dst)
{
    if (g.uform)   // the device volume holds T + P2
        for (int16_t &v : dev) v = (int16_t)(uint16_t)((uint16_t)v - (uint16_t)g.P2);
    const size_t col = (size_t)g.Dp, row = (size_t)g.W1 * col, ocol = (size_t)g.D, orow = (size_t)g.W1 * ocol;
    const int P1 = std::min(g.P1, 32767);
    for (int b = 0; b < B; b++)
        for (int y = 0; y < g.H; y++) {
            const size_t r = g.tonly ? (size_t)b * g.HR + y + y / g.stripe_sz + 1 : (size_t)b * g.H + y;
            const int16_t *T = dev.data() + r * row;
            int16_t *o = dst + ((size_t)b * g.H + y) * orow;
            for (int x = 0; x < g.W1; x++) {
                const int16_t *t = T + x * col;
                if (!g.tonly || stage == 1) { std::memcpy(o + x * ocol, t, ocol * sizeof(int16_t)); continue; }
                const int16_t *p = t - row; // the row above (a stripe's padding row above its first row)
                int minp = 32767;
                for (int d = 0; d < g.D; d++) minp = std::min(minp, (int)p[d]);
                const int mp2 = std::min(minp + g.P2, 32767);
                for (int d = 0; d < g.D; d++) {
                    const int l = d > 0 ? p[d - 1] : 32767, r2 = d < g.D - 1 ? p[d + 1] : 32767;
                    const int a = std::min(std::min(l, r2) + P1, 32767);
                    const int m = std::min(a, std::min((int)p[d], mp2));
                    o[x * ocol + d] = (int16_t)(t[d] - std::max(m - mp2, -32768));
                }
            }
        }
}

extern "C" int l3d_sgbm_fetch_stage(l3d_sgbm *h, int stage, void *host_dst, size_t bytes)
{
    L3D_REQUIRE(h && host_dst, "l3d_sgbm_fetch_stage: null argument");
    const SgbmGeom &g = h->g;
    const size_t B = (size_t)h->last_batch;
    const size_t vol = g.W1 > 0 ? B * g.H * (size_t)g.W1 * g.D * 2 : 0;
    const size_t img = B * g.W * g.H * 2;
    const void *src = nullptr;
    size_t avail = 0;
    switch (stage) {
    case 0: src = h->Cvol; avail = vol; break;
    case 1: src = h->Tvol; avail = vol; break;
    case 2: src = h->raw; avail = img; break;
    default: set_error("l3d_sgbm_fetch_stage: unknown stage %d", stage); return L3D_ERR_INVALID_ARG;
    }
    L3D_REQUIRE(bytes <= avail, "l3d_sgbm_fetch_stage: %zu bytes requested, stage holds %zu", bytes, avail);
    DeviceGuard dg(h->device);
    L3D_HIP_CHECK(hipStreamSynchronize(h->stream));
    L3D_HIP_CHECK(hipDeviceSynchronize());
    if ((g.tonly || g.Dp != g.D) && stage != 2) {
        L3D_REQUIRE(bytes == avail, "l3d_sgbm_fetch_stage: this volume layout returns whole stages (%zu bytes)", avail);
        std::vector<int16_t> dev(B * (g.tonly ? g.HR : g.H) * (size_t)g.W1 * g.Dp);
        L3D_HIP_CHECK(hipMemcpy(dev.data(), g.tonly ? h->Tvol : src, dev.size() * sizeof(int16_t), hipMemcpyDeviceToHost));
        stage_unpack(g, dev, (int)B, stage, (int16_t *)host_dst);
        return L3D_OK;
    }
    L3D_HIP_CHECK(hipMemcpy(host_dst, src, bytes, hipMemcpyDeviceToHost));
    return L3D_OK;
}

extern "C" int l3d_sgbm_profile(l3d_sgbm *h, int enable)
{
    L3D_REQUIRE(h, "l3d_sgbm_profile: null handle");
    DeviceGuard dg(h->device);
    while (h->ev_pending > 0) {
        int rc = sgbm_prof_drain_one(h);
        if (rc) return rc;
    }
    if (enable && !h->ev[0][0]) {
        for (auto &set : h->ev)
            for (hipEvent_t &e : set) L3D_HIP_CHECK(hipEventCreate(&e));
    }
    h->prof = enable != 0;
    for (double &v : h->prof_ms) v = 0;
    h->prof_calls = 0;
    return L3D_OK;
}

extern "C" int l3d_sgbm_profile_read(l3d_sgbm *h, double *total_ms, int n, long *calls)
{
    L3D_REQUIRE(h && total_ms && calls, "l3d_sgbm_profile_read: null argument");
    L3D_REQUIRE(n >= l3d_sgbm::NK, "l3d_sgbm_profile_read: need room for %d kernels", l3d_sgbm::NK);
    DeviceGuard dg(h->device);
    while (h->ev_pending > 0) {
        int rc = sgbm_prof_drain_one(h);
        if (rc) return rc;
    }
    for (int k = 0; k < l3d_sgbm::NK; k++) total_ms[k] = h->prof_ms[k];
    *calls = h->prof_calls;
    return L3D_OK;
}

extern "C" const char *l3d_sgbm_profile_kernel_name(int k)
{
    return (k >= 0 && k < l3d_sgbm::NK) ? k_sgbm_kernel_names[k] : "";
}

extern "C" int l3d_sgbm_profile_kernel_count(void) { return l3d_sgbm::NK; }
