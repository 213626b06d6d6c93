// Device-side building blocks of the semi-global matcher shared by its translation units (sgbm.hip: the 3-path mode
// of the reference; sgbm_paths.hip: the 5- and 8-path modes): geometry, packed 16-bit helpers, the recurrence step,
// winner search and sub-pixel fit, and the dispatch over the built lane-group shapes.
#pragma once
#include "common.h"
#include <type_traits>

namespace l3d {

typedef short v2s __attribute__((ext_vector_type(2)));
typedef unsigned short v2u __attribute__((ext_vector_type(2)));

struct SgbmGeom {
    int W, H, minD, D, SW2, SH2, P1, P2, uniq, disp12MaxDiff, ftzero;
    int minX1, maxX1, W1, invalid16;
    int stripe_sz, stripe_overlap;
    int speckleWindowSize, speckleRange;
    // lanes that share one pixel's disparity vector (8 or 16) and packed int16 pairs per lane
    // (D = 2 * NP * LPP), chosen separately for the vertical and the horizontal kernel: the volumes
    // between them are plain [column][disparity] arrays either way
    int LPPv, NPv, LPPh, NPh;
    int Dp;   // disparities per volume column = 2 * NP * LPP: D, or D + 16 when D is an odd multiple of 16 above 128 (the tail is padding)
    // tonly: the aggregated-cost volume is not stored.  The vertical kernel writes the top->down path volume
    // only, with the per-column minimum beside it, into HR = H + 4 rows per frame: row y of stripe n sits at
    // y + n + 1 and the row before a stripe's first own row holds that stripe's path state entering it (zeros
    // for stripe 0), so "the previous row" of every image row is always the row above it.  The horizontal
    // kernel gets C back as T - (what the recurrence added), which is exact as long as nothing saturates.
    int tonly, HR;
    // vertical kernel: the ring of box-sum rows lives in registers (a kernel variant built for this block size)
    // instead of LDS
    int ring_regs;
    // Path costs are kept with P2 added (so they are non-negative) and the recurrence runs in its unsigned form
    // (sgm_step_u), which needs one instruction less per packed word.  Set when nothing can saturate: path-only layout,
    // no padding disparities, max cost + P2 <= 32767, 2 * max cost <= 32767 and 3 * P2 <= 32767.
    int uform;
    // 2 = the 3-path mode (four stripes); 0 / 1 = the 5- / 8-path modes: one stripe over the whole image, both volumes
    // stored (the second one serves as the sum volume), the paths run by sgbm_paths.hip
    int mode;
};

// ---------------------------------------------------------------------------------------------
// device helpers
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t as_u32(v2s a) { return __builtin_bit_cast(uint32_t, a); }
__device__ __forceinline__ v2s as_v2s(uint32_t a) { return __builtin_bit_cast(v2s, a); }
__device__ __forceinline__ uint32_t as_u32u(v2u a) { return __builtin_bit_cast(uint32_t, a); }
__device__ __forceinline__ v2u as_v2u(uint32_t a) { return __builtin_bit_cast(v2u, a); }

__device__ __forceinline__ uint32_t pk_min(uint32_t a, uint32_t b) { return as_u32(__builtin_elementwise_min(as_v2s(a), as_v2s(b))); }
__device__ __forceinline__ uint32_t pk_max(uint32_t a, uint32_t b) { return as_u32(__builtin_elementwise_max(as_v2s(a), as_v2s(b))); }
__device__ __forceinline__ uint32_t pk_adds(uint32_t a, uint32_t b) { return as_u32(__builtin_elementwise_add_sat(as_v2s(a), as_v2s(b))); }
__device__ __forceinline__ uint32_t pk_subs(uint32_t a, uint32_t b) { return as_u32(__builtin_elementwise_sub_sat(as_v2s(a), as_v2s(b))); }
__device__ __forceinline__ uint32_t pk_add(uint32_t a, uint32_t b) { return as_u32(as_v2s(a) + as_v2s(b)); }
__device__ __forceinline__ uint32_t pk_sub(uint32_t a, uint32_t b) { return as_u32(as_v2s(a) - as_v2s(b)); }
__device__ __forceinline__ uint32_t pk_mul(uint32_t a, int s) { v2s m = {(short)s, (short)s}; return as_u32(as_v2s(a) * m); }
__device__ __forceinline__ uint32_t pk_splat(int s) { return ((uint32_t)s & 0xffffu) * 0x10001u; }
// unsigned 16-bit packed helpers for the Birchfield-Tomasi cost
__device__ __forceinline__ uint32_t pku_subs(uint32_t a, uint32_t b) { return as_u32u(__builtin_elementwise_sub_sat(as_v2u(a), as_v2u(b))); }
__device__ __forceinline__ uint32_t pku_max(uint32_t a, uint32_t b) { return as_u32u(__builtin_elementwise_max(as_v2u(a), as_v2u(b))); }
__device__ __forceinline__ uint32_t pku_min(uint32_t a, uint32_t b) { return as_u32u(__builtin_elementwise_min(as_v2u(a), as_v2u(b))); }

// DPP moves inside a row of 16 lanes.  Lanes without a source keep `fill`.
template <int CTRL>
__device__ __forceinline__ uint32_t dpp_row(uint32_t fill, uint32_t src)
{
    return (uint32_t)__builtin_amdgcn_update_dpp((int)fill, (int)src, CTRL, 0xf, 0xf, false);
}
// The same move when the value of lanes without a source does not matter (they read 0): no destination to initialise
template <int CTRL>
__device__ __forceinline__ uint32_t dpp_row_any(uint32_t src)
{
    return (uint32_t)__builtin_amdgcn_mov_dpp((int)src, CTRL, 0xf, 0xf, true);
}
#define DPP_ROW_SHL1 0x101        // lane i <- lane i+1
#define DPP_ROW_SHR1 0x111        // lane i <- lane i-1
#define DPP_QUAD_XOR1 0xB1        // quad_perm [1,0,3,2]: lane i <- lane i^1
#define DPP_QUAD_XOR2 0x4E        // quad_perm [2,3,0,1]: lane i <- lane i^2
#define DPP_ROW_HALF_MIRROR 0x141 // lane i <- lane 7-i inside each half row
#define DPP_ROW_MIRROR 0x140      // lane i <- lane 15-i
#define DPP_ROW_ROR8 0x128        // lane i <- lane i^8 (rotation by half a row)

// A pixel's disparity vector is spread over a group of LPP consecutive lanes (LPP = 4, 8 or 16,
// groups aligned inside a DPP row).  All-lanes minimum over one group; every move has a source
// for every lane, so `fill` is never used: passing the identity of min lets the compiler fold
// each move into one v_min_*_dpp.  After the two quad steps a quad is uniform, so the mirrored
// moves deliver the other quad's / other half's value.
template <int LPP>
__device__ __forceinline__ uint32_t group_min_u32(uint32_t v)
{
    v = min(v, dpp_row<DPP_QUAD_XOR1>(0xffffffffu, v));
    v = min(v, dpp_row<DPP_QUAD_XOR2>(0xffffffffu, v));
    if (LPP >= 8) v = min(v, dpp_row<DPP_ROW_HALF_MIRROR>(0xffffffffu, v));
    if (LPP >= 16) v = min(v, dpp_row<DPP_ROW_MIRROR>(0xffffffffu, v));
    return v;
}
template <int LPP>
__device__ __forceinline__ int group_min_i32(int v)
{
    v = min(v, (int)dpp_row<DPP_QUAD_XOR1>(0x7fffffffu, (uint32_t)v));
    v = min(v, (int)dpp_row<DPP_QUAD_XOR2>(0x7fffffffu, (uint32_t)v));
    if (LPP >= 8) v = min(v, (int)dpp_row<DPP_ROW_HALF_MIRROR>(0x7fffffffu, (uint32_t)v));
    if (LPP >= 16) v = min(v, (int)dpp_row<DPP_ROW_MIRROR>(0x7fffffffu, (uint32_t)v));
    return v;
}
// Minimum over both int16 halves of `v` and over the lanes of the group, returned in both
// halves of the result.  A word with two equal halves orders as int32 exactly like its halves
// order as int16, and 32-bit v_min_i32 can carry a DPP modifier (packed 16-bit ops cannot).
template <int LPP>
__device__ __forceinline__ uint32_t group_min_splat(uint32_t v)
{
    int m = (int)pk_min(v, __builtin_amdgcn_alignbit(v, v, 16));
    m = min(m, (int)dpp_row<DPP_QUAD_XOR1>(0x7fffffffu, (uint32_t)m));
    m = min(m, (int)dpp_row<DPP_QUAD_XOR2>(0x7fffffffu, (uint32_t)m));
    if (LPP >= 8) m = min(m, (int)dpp_row<DPP_ROW_HALF_MIRROR>(0x7fffffffu, (uint32_t)m));
    if (LPP >= 16) m = min(m, (int)dpp_row<DPP_ROW_MIRROR>(0x7fffffffu, (uint32_t)m));
    return (uint32_t)m;
}

// f(integral_constant<K>), f(integral_constant<K+1>), ... up to N - 1, stopping at the first call that returns false:
// a loop whose counter is a compile-time constant inside the body.
template <int K, int N, typename F>
__device__ __forceinline__ void unroll_while(F &&f)
{
    if constexpr (K < N) {
        if (f(std::integral_constant<int, K>{})) unroll_while<K + 1, N>(f);
    }
}

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also waits for every
// outstanding global load and store (vmcnt(0)), which would drain the row prefetch and expose
// the latency of the volume stores at every row; the tiles exchanged here live in LDS.
__device__ __forceinline__ void lds_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

template <int NP>
struct alignas(NP == 1 ? 4 : (NP == 2 ? 8 : (NP % 4 == 0 ? 16 : (NP % 2 == 0 ? 8 : 4)))) Pack {
    uint32_t v[NP];
};

typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
// Views for memory that is also accessed through another type (the LDS records of summed costs are written as packed
// words and read as single 16-bit entries, and the other way round): may_alias makes those accesses well defined, so
// their order does not hang on a compiler barrier between them.
typedef u32x4 __attribute__((may_alias)) u32x4_any;
template <int NP>
using PackAny __attribute__((may_alias)) = Pack<NP>;
template <int NP>
__device__ __forceinline__ Pack<NP> buf_load(BufRsrc r, uint32_t voff, uint32_t soff)
{
    Pack<NP> o;
    int p = 0;
#pragma unroll
    for (; p + 4 <= NP; p += 4) {
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, voff + 4 * p, soff, 0);
        o.v[p] = v.x; o.v[p + 1] = v.y; o.v[p + 2] = v.z; o.v[p + 3] = v.w;
    }
    if (p + 2 <= NP) {
        const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(r, voff + 4 * p, soff, 0);
        o.v[p] = v.x; o.v[p + 1] = v.y;
        p += 2;
    }
    if (p < NP) o.v[p] = __builtin_amdgcn_raw_buffer_load_b32(r, voff + 4 * p, soff, 0);
    return o;
}
template <int NP>
__device__ __forceinline__ void buf_store(const Pack<NP> &o, BufRsrc r, uint32_t voff, uint32_t soff)
{
    int p = 0;
#pragma unroll
    for (; p + 4 <= NP; p += 4) {
        const u32x4 v = {o.v[p], o.v[p + 1], o.v[p + 2], o.v[p + 3]};
        __builtin_amdgcn_raw_buffer_store_b128(v, r, voff + 4 * p, soff, 0);
    }
    if (p + 2 <= NP) {
        const u32x2 v = {o.v[p], o.v[p + 1]};
        __builtin_amdgcn_raw_buffer_store_b64(v, r, voff + 4 * p, soff, 0);
        p += 2;
    }
    if (p < NP) __builtin_amdgcn_raw_buffer_store_b32(o.v[p], r, voff + 4 * p, soff, 0);
}

// Penalties of the recurrence as packed int16 splats.  prevMin + P2 saturates at 32767 in the
// recurrence; P2 itself may exceed 32767 (P2 = 96*blockSize^2 does from blockSize 19 on), so it
// is applied as up to three saturating additions of non-negative parts, which equals
// min(prevMin + P2, 32767) exactly for every int16 prevMin.
struct SgmPenalty {
    uint32_t P1v, P2a, P2b, P2c;
};
__device__ __forceinline__ SgmPenalty make_penalty(const SgbmGeom &g)
{
    SgmPenalty k;
    k.P1v = pk_splat(min(g.P1, 32767));
    const int a = min(g.P2, 32767), b = min(g.P2 - a, 32767), c = min(g.P2 - a - b, 32767);
    k.P2a = pk_splat(a); k.P2b = pk_splat(b); k.P2c = pk_splat(c);
    return k;
}

// One step of the SGM recurrence on a disparity vector spread over a lane group:
//   out[d] = C[d] + min(prev[d], prev[d-1]+P1, prev[d+1]+P1, prevMin+P2) - (prevMin+P2)
// in saturating int16 arithmetic.  `st` holds prev on entry and out on return; prevMin and the
// returned min_d out carry the value in both halves.  `first` / `last` flag the group's end
// lanes: inside a 16-lane DPP row a shift would otherwise pull in the neighbouring group's value.
//
// PAD variants: the lane group holds 16 more disparities than the matcher has.  Those entries are pinned to
// 32767 in every state, which is what the definition reads beyond the last disparity, so they never win a
// minimum and the real entries see exactly the border they should.
template <int NP, bool PAD>
struct PadFlags {
    bool on[NP];
    // real_words = packed pairs of real disparities in a column
    __device__ __forceinline__ void init(int lane_in_group, int real_words)
    {
#pragma unroll
        for (int p = 0; p < NP; p++) on[p] = PAD && lane_in_group * NP + p >= real_words;
    }
    __device__ __forceinline__ uint32_t initial(int p) const { return (PAD && on[p]) ? 0x7fff7fffu : 0u; }
};

// BIGP2 = false: the caller knows that P2 <= 32767 (one addition applies it).
template <int NP, int LPP, bool PAD, bool BIGP2 = true>
__device__ __forceinline__ uint32_t sgm_step(const uint32_t (&c)[NP], uint32_t (&st)[NP], uint32_t prevMin, const SgmPenalty &k, bool first,
                                             bool last, const PadFlags<NP, PAD> &pad)
{
    const uint32_t BIG = 0x7fff7fffu;
    uint32_t mp2v = pk_adds(prevMin, k.P2a);
    if (BIGP2 && k.P2b) {
        // kernel-uniform and rare (P2 > 32767): the empty statement keeps this a branch instead of two
        // additions and a select on every step
        asm volatile("" ::: "memory");
        mp2v = pk_adds(pk_adds(mp2v, k.P2b), k.P2c);
    }
    // lanes without a source keep the first operand: BIG at the ends of a 16-lane row; with narrower groups the
    // end lanes are overridden below anyway, so the move needs no initialised destination
    uint32_t below = LPP < 16 ? dpp_row_any<DPP_ROW_SHR1>(st[NP - 1]) : dpp_row<DPP_ROW_SHR1>(BIG, st[NP - 1]); // lane-1's top pair
    uint32_t above = LPP < 16 ? dpp_row_any<DPP_ROW_SHL1>(st[0]) : dpp_row<DPP_ROW_SHL1>(BIG, st[0]);           // lane+1's bottom pair
    if (LPP < 16) {
        below = first ? BIG : below;
        above = last ? BIG : above;
    }
    uint32_t out[NP];
    uint32_t mn = BIG;
#pragma unroll
    for (int p = 0; p < NP; p++) {
        const uint32_t lo_src = (p > 0) ? st[p - 1] : below;
        const uint32_t hi_src = (p < NP - 1) ? st[p + 1] : above;
        const uint32_t dm1 = __builtin_amdgcn_alignbit(st[p], lo_src, 16); // (prev[d-1]) for both halves
        const uint32_t dp1 = __builtin_amdgcn_alignbit(hi_src, st[p], 16); // (prev[d+1])
        const uint32_t a = pk_adds(pk_min(dm1, dp1), k.P1v);
        const uint32_t m = pk_min(a, pk_min(st[p], mp2v));
        out[p] = pk_adds(c[p], pk_subs(m, mp2v));
        if (PAD) out[p] = pad.on[p] ? BIG : out[p];
        mn = pk_min(mn, out[p]);
    }
#pragma unroll
    for (int p = 0; p < NP; p++) st[p] = out[p];
    return group_min_splat<LPP>(mn);
}

// The same recurrence on costs that carry an offset of P2 (state' = state + P2 >= 0, c' = c + P2, prevMin' = prevMin + P2),
// valid when no operation of the signed form can saturate (SgbmGeom::uform).  With m = prevMin' + P2:
//   out'[d] = c'[d] - w[d],   w[d] = m - min(st'[d], st'[d-1] + P1, st'[d+1] + P1, m)
//                                  = max(e[d], max(e[d-1], e[d+1]) -sat P1),   e[d] = m -sat st'[d]
// in unsigned saturating arithmetic: the subtraction that clamps at zero does the work of one of the minima.  A missing
// neighbour contributes e = 0, which is what a DPP move delivers to a lane without a source.
template <int NP, int LPP>
__device__ __forceinline__ void sgm_excess(const uint32_t (&st)[NP], uint32_t m, uint32_t P1v, bool first, bool last, uint32_t (&w)[NP])
{
    uint32_t e[NP];
#pragma unroll
    for (int p = 0; p < NP; p++) e[p] = pku_subs(m, st[p]);
    uint32_t below = dpp_row_any<DPP_ROW_SHR1>(e[NP - 1]); // lane-1's top pair
    uint32_t above = dpp_row_any<DPP_ROW_SHL1>(e[0]);      // lane+1's bottom pair
    if (LPP < 16) {
        below = first ? 0u : below;
        above = last ? 0u : above;
    }
#pragma unroll
    for (int p = 0; p < NP; p++) {
        const uint32_t lo_src = (p > 0) ? e[p - 1] : below;
        const uint32_t hi_src = (p < NP - 1) ? e[p + 1] : above;
        const uint32_t dm1 = __builtin_amdgcn_alignbit(e[p], lo_src, 16);
        const uint32_t dp1 = __builtin_amdgcn_alignbit(hi_src, e[p], 16);
        w[p] = pku_max(e[p], pku_subs(pku_max(dm1, dp1), P1v));
    }
}
// Minimum over both uint16 halves of `v` and over the lanes of the group, in both halves of the result.
template <int LPP>
__device__ __forceinline__ uint32_t group_min_splat_u(uint32_t v)
{
    return group_min_u32<LPP>(pku_min(v, __builtin_amdgcn_alignbit(v, v, 16)));
}
template <int NP, int LPP>
__device__ __forceinline__ uint32_t sgm_step_u(const uint32_t (&c)[NP], uint32_t (&st)[NP], uint32_t prevMin, const SgmPenalty &k, bool first, bool last)
{
    uint32_t w[NP];
    sgm_excess<NP, LPP>(st, pk_add(prevMin, k.P2a), k.P1v, first, last, w);
    uint32_t mn = 0xffffffffu;
#pragma unroll
    for (int p = 0; p < NP; p++) {
        st[p] = pk_sub(c[p], w[p]);
        mn = pku_min(mn, st[p]);
    }
    return group_min_splat_u<LPP>(mn);
}

// Winner-take-all and uniqueness test for one column from the summed costs S, on all lanes of the
// group.  Returns the group-uniform key (biased min S << 16 | argmin), or 0xFFFFFFFF when the
// uniqueness test rejects the column.
// `window` is a table in LDS (uniq_window_fill): entry r + 2, r = best - d0 clamped to -2 .. 2*NP+1, holds for each of
// the lane's 2*NP disparities 32767 if it lies in best-1 .. best+1 and -32768 otherwise.
template <int NP>
__device__ __forceinline__ void uniq_window_fill(uint32_t *window, int tid, int nthreads)
{
    for (int i = tid; i < (2 * NP + 4) * NP; i += nthreads) {
        const int r = i / NP - 2, p = i - (i / NP) * NP;
        const uint32_t lo = abs(2 * p - r) <= 1 ? 0x7fffu : 0x8000u, hi = abs(2 * p + 1 - r) <= 1 ? 0x7fffu : 0x8000u;
        window[i] = lo | (hi << 16);
    }
}

// The winner alone: the group-uniform key (biased min S << 16 | argmin); the first minimum wins.
template <int NP, int LPP>
__device__ __forceinline__ uint32_t wta_min_key(const uint32_t (&S)[NP], int d0)
{
    // per disparity a signed 32-bit key (S << 16 | d): ordering by S, then by d; the disparities of a word sit side by
    // side in dpair[p], so each key is one byte permute
    int skey = 0x7fffffff;
#pragma unroll
    for (int p = 0; p < NP; p++) {
        const uint32_t dpair = (uint32_t)(d0 + 2 * p) * 0x10001u + 0x10000u;
        const int k0 = (int)__builtin_amdgcn_perm(S[p], dpair, 0x05040100u);   // S low half  : disparity d0 + 2p
        const int k1 = (int)__builtin_amdgcn_perm(S[p], dpair, 0x07060302u);   // S high half : disparity d0 + 2p + 1
        skey = min(skey, min(k0, k1));
    }
    skey = group_min_i32<LPP>(skey);
    return (uint32_t)skey ^ 0x80000000u;   // cost biased by 32768: the unsigned form the later stages order by
}

template <int NP, int LPP>
__device__ __forceinline__ uint32_t wta_key(const SgbmGeom &g, const uint32_t (&S)[NP], int d0, const uint32_t *window)
{
    const uint32_t key = wta_min_key<NP, LPP>(S, d0);
    if (g.uniq <= 0) return key;
    const int best = (int)(key & 0xffffu);
    const int minS = (int)(key >> 16) - 32768;
    const int lim = minS * 100, f = 100 - g.uniq;
    bool reject;
    if (f > 0) {
        // some d outside best-1..best+1 has S[d]*f < lim  <=>  the minimum of S outside that window has:
        // entries inside the window are raised to 32767 by a packed max with the table row of this lane
        const int r = min(max(best - d0, -2), 2 * NP + 1);
        const Pack<NP> in_window = *reinterpret_cast<const Pack<NP> *>(window + (r + 2) * NP);
        uint32_t acc = 0x7fff7fffu;
#pragma unroll
        for (int p = 0; p < NP; p++) acc = pk_min(acc, pk_max(S[p], in_window.v[p]));
        const int m2 = (int)(short)(group_min_splat<LPP>(acc) & 0xffffu);
        reject = __mul24(m2, f) < lim;
    } else {
        bool bad = false;
#pragma unroll
        for (int p = 0; p < NP; p++) {
            const v2s s2 = as_v2s(S[p]);
            const int da = d0 + 2 * p, db = da + 1;
            bad |= ((int)s2.x * f < lim) && (abs(da - best) > 1);
            bad |= ((int)s2.y * f < lim) && (abs(db - best) > 1);
        }
        const int grp = (int)(threadIdx.x & 63) / LPP;
        const unsigned long long bal = __ballot(bad);
        reject = ((bal >> (grp * LPP)) & ((1ull << LPP) - 1ull)) != 0;
    }
    return reject ? 0xffffffffu : key;
}

// Sub-pixel disparity (x16) from the winning cost sc at `best` and its two neighbours:
// OpenCV's integer parabola fit with the truncating division done in float plus an exact fix-up
// (numerator and denominator are below 2^24 and the quotient is at most 8 in magnitude).
__device__ __forceinline__ int subpixel_disp(const SgbmGeom &g, int best, int sm, int sc, int sp)
{
    int dd = best * 16;
    if (0 < best && best < g.D - 1) {
        const int denom2 = max(sm + sp - 2 * sc, 1);
        const int num = (sm - sp) * 16 + denom2, den = denom2 * 2;
        const int a = abs(num);
        int q = (int)((float)a * __builtin_amdgcn_rcpf((float)den));
        int r = a - q * den;
        if (r < 0) { q--; r += den; }
        if (r >= den) q++;
        dd += num < 0 ? -q : q;
    }
    return dd + g.minD * 16;
}

} // namespace l3d

// Calls f(integral_constant<NP>, integral_constant<LPP>) for the built kernel variant; false if there is none.
template <typename F>
static inline bool dispatch_variant(int NP, int LPP, bool pad, F &&f)
{
#define L3D_VARIANT_PAD(np, lpp, pd)                                                                                     \
    if (NP == np && LPP == lpp && pad == pd) {                                                                           \
        f(std::integral_constant<int, np>{}, std::integral_constant<int, lpp>{}, std::integral_constant<bool, pd>{});    \
        return true;                                                                                                     \
    }
#define L3D_VARIANT(np, lpp) L3D_VARIANT_PAD(np, lpp, false)
    L3D_VARIANT(1, 8) L3D_VARIANT(2, 8) L3D_VARIANT(3, 8) L3D_VARIANT(4, 8)
    L3D_VARIANT(5, 8) L3D_VARIANT(6, 8) L3D_VARIANT(7, 8) L3D_VARIANT(8, 8)
    L3D_VARIANT(1, 16) L3D_VARIANT(2, 16) L3D_VARIANT(3, 16) L3D_VARIANT(4, 16)
    L3D_VARIANT(5, 16) L3D_VARIANT(6, 16) L3D_VARIANT(7, 16) L3D_VARIANT(8, 16)
    // 144, 176, 208, 240 disparities (and 112 with the largest blocks)
    L3D_VARIANT_PAD(4, 16, true) L3D_VARIANT_PAD(5, 16, true) L3D_VARIANT_PAD(6, 16, true) L3D_VARIANT_PAD(7, 16, true) L3D_VARIANT_PAD(8, 16, true)
#undef L3D_VARIANT
#undef L3D_VARIANT_PAD
    return false;
}


// The 5- and 8-path modes (sgbm_paths.hip): path passes over the aggregated-cost volume into the sum volume, then
// winner / uniqueness / sub-pixel per column into wta and dsub.  mode: 0 = MODE_SGBM, 1 = MODE_HH.
int sgbm_paths_aggregate(const l3d::SgbmGeom &g, int mode, int B, const uint32_t *Cvol, uint32_t *Svol, uint32_t *wta, int16_t *dsub,
                         hipStream_t s);

// The horizontal kernel of the 3-path mode (sgbm_horizontal.hip); false when no variant is built for the shape.
bool sgbm_launch_horizontal(const l3d::SgbmGeom &g, int B, const uint32_t *Cvol, const uint32_t *Tvol, const uint32_t *Mvol, uint32_t *ckpt,
                            uint32_t *wta, int16_t *dsub, hipStream_t s);
