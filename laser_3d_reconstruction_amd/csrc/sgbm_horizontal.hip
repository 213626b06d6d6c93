// The horizontal kernel of the 3-path matcher (left->right and right->left paths, sum with the top->down path, winner /
// uniqueness / sub-pixel), in a translation unit of its own: it is the one kernel here that gains from the compiler's
// max-ILP scheduling strategy (two thirds fewer wait states between dependent packed 16-bit operations), which costs
// the vertical kernel 2 %, and the strategy is chosen per translation unit (build.sh).
#include "sgbm_dev.h"

namespace l3d {

// ---------------------------------------------------------------------------------------------
// K3: left->right and right->left paths, sum of the three paths, WTA / uniqueness / sub-pixel.
// One lane group per image row; a wave covers 64/LPP consecutive rows.
// The left->right costs are needed in right->left order.  Instead of spilling them to HBM the
// first sweep stores only its state every SEG columns (checkpoints); the second sweep walks the
// segments right to left, re-runs the left->right recurrence inside the segment from its
// checkpoint into registers, then runs the right->left recurrence over the same columns.  All
// loads are issued one segment ahead of their use.
// ---------------------------------------------------------------------------------------------
template <int NP, bool TONLY>
struct SegLen {
    // TONLY prefetches three operands per column.  The longer segments fit the registers only when the next segment is
    // requested late, a few columns at a time (LateFetch below): 16 columns with 3 or 4 words a lane, 8 with 5 or 6.
    // (3 words a lane = 96 disparities on 16 lanes, the shape of single-frame handles: 3.27 -> 2.97 ms per 128 pairs
    // against 4-column segments; batches of that shape run on 8 lanes with 6 words and 8-column segments, 2.62 ms.)
    static constexpr int value = ((NP == 4 || NP == 3) && TONLY) ? 16 : ((NP <= 2 && !TONLY) || ((NP == 5 || NP == 6) && TONLY)) ? 8 : 4;
};

// Whether the second sweep requests the next segment from inside the current one, in two halves (after the left -> right
// costs of the current segment are rebuilt, and half way through its right -> left steps), instead of whole and ahead of it:
// the operands of two whole segments at 13 registers a column do not fit beside the segment's left -> right costs.
template <int NP, bool TONLY>
struct LateFetch {
    static constexpr bool value = TONLY && ((NP > 4 && SegLen<NP, TONLY>::value == 8) || SegLen<NP, TONLY>::value == 16);
};

// The aggregated cost of a column from the path volume: C = T - (min(P[d], P[d-1]+P1, P[d+1]+P1, minP+P2) - (minP+P2)),
// P = the path costs of the row above, minP their minimum (both halves).  Exact when the recurrence never saturated.
// UF: the volume holds the path costs with P2 added and the result is the aggregated cost with P2 added (sgm_step_u).
template <int NP, int LPP, bool UF>
__device__ __forceinline__ void cost_from_paths(const uint32_t (&T)[NP], const uint32_t (&P)[NP], uint32_t minP, const SgmPenalty &k, bool first,
                                                bool last, uint32_t (&C)[NP])
{
    if constexpr (UF) {
        uint32_t w[NP];
        sgm_excess<NP, LPP>(P, pk_add(minP, k.P2a), k.P1v, first, last, w);
#pragma unroll
        for (int p = 0; p < NP; p++) C[p] = pk_add(T[p], w[p]);
        return;
    }
    const uint32_t BIG = 0x7fff7fffu;
    const uint32_t mp2v = pk_adds(minP, k.P2a);
    uint32_t below = LPP < 16 ? dpp_row_any<DPP_ROW_SHR1>(P[NP - 1]) : dpp_row<DPP_ROW_SHR1>(BIG, P[NP - 1]);
    uint32_t above = LPP < 16 ? dpp_row_any<DPP_ROW_SHL1>(P[0]) : dpp_row<DPP_ROW_SHL1>(BIG, P[0]);
    if (LPP < 16) {
        below = first ? BIG : below;
        above = last ? BIG : above;
    }
#pragma unroll
    for (int p = 0; p < NP; p++) {
        const uint32_t lo_src = (p > 0) ? P[p - 1] : below;
        const uint32_t hi_src = (p < NP - 1) ? P[p + 1] : above;
        const uint32_t dm1 = __builtin_amdgcn_alignbit(P[p], lo_src, 16);
        const uint32_t dp1 = __builtin_amdgcn_alignbit(hi_src, P[p], 16);
        const uint32_t a = pk_adds(pk_min(dm1, dp1), k.P1v);
        const uint32_t m = pk_min(a, pk_min(P[p], mp2v));
        C[p] = pk_sub(T[p], pk_subs(m, mp2v));
    }
}

// What the horizontal kernel fetches per column: the aggregated cost, or (TONLY) the path costs of the row and of
// the row above plus the latter's minimum, from which the aggregated cost is rebuilt at the point of use.
template <int NP, bool TONLY>
struct ColIn;
template <int NP>
struct ColIn<NP, false> {
    Pack<NP> c;
};
template <int NP>
struct ColIn<NP, true> {
    Pack<NP> t, p;
    uint32_t m;
};

template <int NP, int LPP, bool TONLY, bool PAD, bool UF>
__global__ void __launch_bounds__(64) k_horizontal(SgbmGeom g, int nrows_total, const uint32_t *__restrict__ Cvol,
                                                   const uint32_t *__restrict__ Tvol, const uint32_t *__restrict__ Mvol,
                                                   uint32_t *__restrict__ ckpt, uint32_t *__restrict__ wta, int16_t *__restrict__ dsub)
{
    constexpr int SEG = SegLen<NP, TONLY>::value;
    constexpr bool LATE = LateFetch<NP, TONLY>::value;
    constexpr int RPW = 64 / LPP; // image rows per wave
    constexpr int CW = NP * LPP;  // u32 per volume column
    constexpr int D = 2 * CW;
    constexpr int REC = D + 8;    // int16 per staged column: the D summed costs and a gap, so best-1 / best+1 stay inside the array
    // summed costs and winner keys of one segment, parked for the step that runs once per segment: uniqueness test and
    // sub-pixel fit of the segment's RPW * SEG (row, column) pairs, LPI lanes to a pair.  (Per column, on the lanes that
    // hold the sums, the uniqueness test costs a table read, two packed operations per word and a reduction over the
    // group in which every lane computes the same value: 25 instructions per column against 8 here.)
    // the per-segment step takes at most 8 columns at a time: a 16-column segment runs it twice
    constexpr int SSEG = SEG < 8 ? SEG : 8;
    __shared__ __align__(16) int16_t sS[8 + RPW * SSEG * REC];
    __shared__ uint32_t sK[RPW * SSEG];
    // results of OSEG segments per row, so that they leave as 64-byte (keys) and 32-byte (disparities) runs:
    // 16- and 8-byte stores scattered over rows reached HBM as partial lines (+40 % write traffic)
    constexpr int OSEG = 16 / SEG > 0 ? 16 / SEG : 1, OCOL = OSEG * SEG;
    __shared__ uint32_t oK[RPW * OCOL];
    __shared__ int16_t oD[RPW * OCOL];
    constexpr int ITEMS = RPW * SSEG, LPI = 64 / ITEMS;  // pairs per step, lanes per pair
    constexpr int WPL = CW / LPI;                        // words of a pair's summed costs that one lane scans
    static_assert(LPI >= 1 && LPI <= 4 && LPI * ITEMS == 64 && WPL % 4 == 0, "per-segment step: lane split");
    const int tid = threadIdx.x, grp = tid / LPP, l = tid & (LPP - 1);
    const bool first = l == 0, last = l == LPP - 1;
    // Vertically adjacent row groups share a row of the volume: they run on one XCD.  (Putting them into one block
    // of 2 or 4 waves shares that row through the L1 as well and moves 6% fewer bytes, but measured 7-10% slower.)
    const int row0 = xcd_contiguous(blockIdx.x, gridDim.x) * RPW;
    // groups past the last row redo the last row (their lanes are needed for the per-segment step) and store nothing
    const bool row_ok = row0 + grp < nrows_total;
    const int grp_ld = min(grp, nrows_total - 1 - row0);
    const int W1 = g.W1;
    const int nseg = (W1 + SEG - 1) / SEG;
    constexpr size_t col_bytes = (size_t)CW * sizeof(uint32_t);
    // block-uniform bases plus a fixed per-lane byte offset (row inside the block, lane inside the column)
    // volume row of an image row (index over batch * H): plain, or with the stripe padding of the path-only layout
    auto vol_row = [&](int row) -> size_t {
        if (!TONLY) return (size_t)row;
        const int b = row / g.H, y = row - b * g.H;
        return (size_t)b * g.HR + y + y / g.stripe_sz + 1;
    };
    const size_t vrow0 = vol_row(row0) - (TONLY ? 1 : 0);       // uniform base row: one above the first row when the row above is read too
    const size_t vrel = vol_row(row0 + grp_ld) - vrow0;         // this lane group's row relative to it
    // Every access is a buffer access: a block-uniform descriptor (base of the block's rows), a uniform byte offset
    // of the column in a scalar register and a per-lane offset that never changes -- no vector ALU work on addresses.
    const BufRsrc Cr = make_rsrc(reinterpret_cast<const char *>(Cvol) + (TONLY ? 0 : vrow0 * W1 * col_bytes));
    const BufRsrc Tr = make_rsrc(reinterpret_cast<const char *>(Tvol) + vrow0 * W1 * col_bytes);
    const BufRsrc Mr = make_rsrc(reinterpret_cast<const char *>(Mvol) + (TONLY ? vrow0 * W1 * sizeof(uint32_t) : 0));
    const BufRsrc Kr = make_rsrc(reinterpret_cast<const char *>(ckpt) + (size_t)row0 * nseg * col_bytes);
    const uint32_t voff = (uint32_t)((vrel * W1 * CW + l * NP) * sizeof(uint32_t));
    const uint32_t poff = voff - (uint32_t)(TONLY ? W1 * col_bytes : 0);       // the row above
    const uint32_t moff = (uint32_t)((vrel - (TONLY ? 1 : 0)) * W1 * sizeof(uint32_t));
    const uint32_t koff = (uint32_t)(((size_t)grp_ld * nseg * CW + l * NP) * sizeof(uint32_t));
    const SgmPenalty pen = make_penalty(g);
    const int d0 = l * 2 * NP; // first disparity index held by this lane
    auto fetch = [&](int x) __attribute__((always_inline)) {
        const uint32_t xc = (uint32_t)min(max(x, 0), W1 - 1);
        ColIn<NP, TONLY> r;
        if constexpr (TONLY) {
            r.t = buf_load<NP>(Tr, voff, xc * (uint32_t)col_bytes);
            r.p = buf_load<NP>(Tr, poff, xc * (uint32_t)col_bytes);
            r.m = __builtin_amdgcn_raw_buffer_load_b32(Mr, moff, xc * (uint32_t)sizeof(uint32_t), 0);
        } else
            r.c = buf_load<NP>(Cr, voff, xc * (uint32_t)col_bytes);
        return r;
    };
    auto ldT = [&](int x) __attribute__((always_inline)) { return buf_load<NP>(Tr, voff, (uint32_t)min(max(x, 0), W1 - 1) * (uint32_t)col_bytes); };
    // aggregated cost of a fetched column, left in place (TONLY: rebuilt into the slot of the row above's costs)
    auto cost = [&](ColIn<NP, TONLY> &r) __attribute__((always_inline)) -> const uint32_t(&)[NP] {
        if constexpr (TONLY) {
            uint32_t c[NP];
            cost_from_paths<NP, LPP, UF>(r.t.v, r.p.v, r.m, pen, first, last, c);
#pragma unroll
            for (int p = 0; p < NP; p++) r.p.v[p] = c[p];
            return r.p.v;
        } else
            return r.c.v;
    };
    PadFlags<NP, PAD> pad;
    pad.init(l, g.D / 2);
    // one step of a horizontal path
    auto path_step = [&](const uint32_t(&c)[NP], uint32_t(&stt)[NP], uint32_t pm) __attribute__((always_inline)) -> uint32_t {
        if constexpr (UF) return sgm_step_u<NP, LPP>(c, stt, pm, pen, first, last);
        else return sgm_step<NP, LPP, PAD, !TONLY>(c, stt, pm, pen, first, last, pad);
    };
    // role of this lane in the per-segment step: part e_h of column e_u of row e_grp
    const int e_item = tid / LPI, e_h = tid % LPI;
    const int e_u = e_item % SSEG, e_grp = e_item / SSEG;
    const bool e_on = e_h == 0 && row0 + e_grp < nrows_total;
    const int uniq_f = 100 - g.uniq;
    uint32_t *wblk = wta + (size_t)row0 * W1;
    int16_t *dblk = dsub + (size_t)row0 * W1;
    const int rows_here = min(RPW, nrows_total - row0);

    // segment being consumed (A) and segment in flight (B)
    struct Seg {
        ColIn<NP, TONLY> c[SEG];
        Pack<NP> t[TONLY ? 1 : SEG]; // !TONLY, sweep 2: the row's path costs
        Pack<NP> ck;                 // sweep 2: left -> right state at the segment start
    };
    Seg A, B;

    // ---- sweep 1: left -> right, keep the state at every segment start ----
    uint32_t st[NP];
#pragma unroll
    for (int p = 0; p < NP; p++) st[p] = UF ? pen.P2a : pad.initial(p);   // a path starts from zero costs (UF: + P2)
    uint32_t mn = UF ? pen.P2a : 0u;
    // The columns pass through a ring of 2 * SEG register slots: the slot of a column is refilled, right after
    // its use, with the column 2 * SEG further on, so every load has the time of 2 * SEG steps to arrive.  (This
    // sweep has a third of the arithmetic of the second one per column: with whole segments fetched one segment
    // ahead it waited for memory.)
    {
        constexpr int RING = LATE ? SEG : 2 * SEG;
        constexpr int BURST = 4;   // measured 1 / 2 / 4 / 8: 3.61 / 3.59 / 3.58 / 3.58 ms at 640x360, 64 disparities, 256 frames
        ColIn<NP, TONLY> ring[RING];
#pragma unroll
        for (int u = 0; u < RING; u++) ring[u] = fetch(u);
        for (int x0 = 0; x0 < W1; x0 += RING) {
#pragma unroll
            for (int u = 0; u < RING; u++) {
                const int x = x0 + u;
                if (x < W1) {
                    if (u % SEG == 0) {
                        Pack<NP> o;
#pragma unroll
                        for (int p = 0; p < NP; p++) o.v[p] = st[p];
                        if (row_ok) buf_store<NP>(o, Kr, koff, (uint32_t)(x / SEG) * (uint32_t)col_bytes);
                    }
                    mn = path_step(cost(ring[u]), st, mn);
                }
                // the slots are refilled BURST at a time: requests for adjacent columns of a row reach memory together
                if ((u + 1) % BURST == 0) {
#pragma unroll
                    for (int k = u + 1 - BURST; k <= u; k++) ring[k] = fetch(x0 + k + RING);
                }
            }
        }
    }

    // ---- sweep 2: segments right -> left ----
    uint32_t rst[NP];
#pragma unroll
    for (int p = 0; p < NP; p++) rst[p] = UF ? pen.P2a : pad.initial(p);
    uint32_t rmn = UF ? pen.P2a : 0u;
    // UF: the left -> right costs enter the sum with 3 * P2 taken off, which is the offset of the three summands together:
    // (R' + (L' - 3 P2)) + T' = R + L + T, the last addition saturating as in the definition (the first cannot)
    const uint32_t P2x3 = pk_splat(3 * g.P2);
    // columns u0 .. u1-1 of segment j (and its checkpoint with the first of them)
    auto fetch2_part = [&](Seg &s, int j, auto u0c, auto u1c) __attribute__((always_inline)) {
        constexpr int u0 = decltype(u0c)::value, u1 = decltype(u1c)::value;
        const int jc = max(j, 0);
#pragma unroll
        for (int u = u0; u < u1; u++) {
            s.c[u] = fetch(jc * SEG + u);
            if constexpr (!TONLY) s.t[u] = ldT(jc * SEG + u);
        }
        if (u0 == 0) s.ck = buf_load<NP>(Kr, koff, (uint32_t)jc * (uint32_t)col_bytes);
    };
    auto fetch2 = [&](Seg &s, int j) __attribute__((always_inline)) {
        fetch2_part(s, j, std::integral_constant<int, 0>{}, std::integral_constant<int, SEG>{});
    };
    // The step that runs once per (at most 8) finished columns of a segment: uniqueness test and parabola fit per (row, column).
    // half: which 8 columns of a 16-column segment.
    auto seg_step = [&](int j, int half) __attribute__((always_inline)) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        {
            const int xc = j * SEG + half * SSEG + e_u;
            const uint32_t key = sK[e_item];
            const int best = xc < W1 ? (int)(key & 0xffffu) : 0;   // columns past the row's end hold nothing: stay inside the record
            const int sc = (int)(key >> 16) - 32768;
            int16_t *rec = sS + 8 + e_item * REC;
            const int sm = rec[best - 1], sp = rec[best + 1];
            bool reject = false;
            if (g.uniq > 0) {
                // rejected when some d outside best-1 .. best+1 has S[d] * (100 - uniq) < min S * 100, i.e. when the smallest
                // such S[d] has (the largest, should 100 - uniq be negative): the three entries are overwritten with a
                // value that cannot win, then the record is scanned, each of the pair's lanes taking a part
                const bool lo = uniq_f > 0;
                const int16_t never = lo ? (int16_t)32767 : (int16_t)-32768;
                if (e_h == 0) { rec[best - 1] = never; rec[best] = never; rec[best + 1] = never; }
                // (the scan below reads the record through a may_alias vector type, so it stays behind the three stores)
                const u32x4_any *q = reinterpret_cast<const u32x4_any *>(rec + e_h * 2 * WPL);
                int m;
                if (lo) {
                    uint32_t acc = 0x7fff7fffu;
#pragma unroll
                    for (int k = 0; k < WPL / 4; k++) {
                        const u32x4 v = q[k];
                        acc = pk_min(acc, pk_min(pk_min(v.x, v.y), pk_min(v.z, v.w)));
                    }
                    m = (int)pk_min(acc, __builtin_amdgcn_alignbit(acc, acc, 16));
                    if (LPI >= 2) m = min(m, (int)dpp_row<DPP_QUAD_XOR1>(0x7fffffffu, (uint32_t)m));
                    if (LPI >= 4) m = min(m, (int)dpp_row<DPP_QUAD_XOR2>(0x7fffffffu, (uint32_t)m));
                } else {
                    uint32_t acc = 0x80008000u;
#pragma unroll
                    for (int k = 0; k < WPL / 4; k++) {
                        u32x4 v = q[k];
                        if (PAD) {   // the padding entries hold 32767: not part of the range
                            const int w0 = e_h * WPL + 4 * k, real = g.D / 2;
                            v.x = w0 < real ? v.x : 0x80008000u; v.y = w0 + 1 < real ? v.y : 0x80008000u;
                            v.z = w0 + 2 < real ? v.z : 0x80008000u; v.w = w0 + 3 < real ? v.w : 0x80008000u;
                        }
                        acc = pk_max(acc, pk_max(pk_max(v.x, v.y), pk_max(v.z, v.w)));
                    }
                    m = (int)pk_max(acc, __builtin_amdgcn_alignbit(acc, acc, 16));
                    if (LPI >= 2) m = max(m, (int)dpp_row<DPP_QUAD_XOR1>(0x80000000u, (uint32_t)m));
                    if (LPI >= 4) m = max(m, (int)dpp_row<DPP_QUAD_XOR2>(0x80000000u, (uint32_t)m));
                }
                reject = (int)(short)(m & 0xffff) * uniq_f < sc * 100;
            }
            if (e_on && xc < W1) {
                const int slot = e_grp * OCOL + (j % OSEG) * SEG + half * SSEG + e_u;
                oK[slot] = reject ? 0xffffffffu : key;
                oD[slot] = (int16_t)(reject ? 0 : subpixel_disp(g, best, sm, sc, sp));
            }
        }
        asm volatile("" ::: "memory");
    };
    // `nxt`, `jn`: the segment to request from inside this one (LATE)
    auto run2 = [&](Seg &s, int j, Seg &nxt, int jn) __attribute__((always_inline)) {
        // left -> right costs of this segment, recomputed from the checkpoint
        uint32_t fst[NP], mv = UF ? 0xffffffffu : 0x7fff7fffu;
#pragma unroll
        for (int p = 0; p < NP; p++) { fst[p] = s.ck.v[p]; mv = UF ? pku_min(mv, fst[p]) : pk_min(mv, fst[p]); }
        uint32_t fmn = UF ? group_min_splat_u<LPP>(mv) : group_min_splat<LPP>(mv); // the previous minimum is the minimum of the previous state
        uint32_t Lseg[SEG][NP];
#pragma unroll
        for (int u = 0; u < SEG; u++) {
            if (j * SEG + u < W1) fmn = path_step(cost(s.c[u]), fst, fmn);
#pragma unroll
            for (int p = 0; p < NP; p++) Lseg[u][p] = UF ? pk_sub(fst[p], P2x3) : fst[p];
        }
        // LATE: the next segment is requested four columns at a time while the registers of this one free up
        auto part = [&](auto k) __attribute__((always_inline)) {
            constexpr int u0 = 4 * decltype(k)::value;
            fetch2_part(nxt, jn, std::integral_constant<int, u0>{}, std::integral_constant<int, u0 + 4>{});
        };
        if constexpr (LATE && SEG == 8) part(std::integral_constant<int, 0>{});
#pragma unroll
        for (int u = SEG - 1; u >= 0; u--) {
            const int xc = j * SEG + u;
            if constexpr (LATE && SEG == 8) {
                if (u == 3) part(std::integral_constant<int, 1>{});
            }
            if constexpr (LATE && SEG == 16) {
                if (u == 11) part(std::integral_constant<int, 0>{});
                if (u == 8) part(std::integral_constant<int, 1>{});
                if (u == 5) part(std::integral_constant<int, 2>{});
                if (u == 2) part(std::integral_constant<int, 3>{});
            }
            if (xc < W1) {
                // (TONLY: cost() already ran on this column in the loop above and left C in place)
                const uint32_t(&cu)[NP] = [&]() -> const uint32_t(&)[NP] { if constexpr (TONLY) return s.c[u].p.v; else return s.c[u].c.v; }();
                const uint32_t(&tu)[NP] = [&]() -> const uint32_t(&)[NP] { if constexpr (TONLY) return s.c[u].t.v; else return s.t[u].v; }();
                rmn = path_step(cu, rst, rmn);
                Pack<NP> S;
#pragma unroll
                for (int p = 0; p < NP; p++) S.v[p] = pk_adds(UF ? pk_add(rst[p], Lseg[u][p]) : pk_adds(rst[p], Lseg[u][p]), tu[p]);
                const uint32_t key = wta_min_key<NP, LPP>(S.v, d0);
                *reinterpret_cast<PackAny<NP> *>(sS + 8 + (grp * SSEG + u % SSEG) * REC + d0) = S;
                if (first) sK[grp * SSEG + u % SSEG] = key;
            }
            if (u % SSEG == 0) seg_step(j, u / SSEG);
        }
        if (j % OSEG == 0) {
            // the columns j*SEG .. j*SEG+OCOL-1 of every row are complete
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int i = tid; i < RPW * OCOL; i += 64) {
                const int r = i / OCOL, c = i - r * OCOL, x = j * SEG + c;
                if (r < rows_here && x < W1) {
                    wblk[(size_t)r * W1 + x] = oK[i];
                    dblk[(size_t)r * W1 + x] = oD[i];
                }
            }
        }
        asm volatile("" ::: "memory");
    };
    fetch2(A, nseg - 1);
    int j = nseg - 1;
    for (; j >= 1; j -= 2) {
        // two segments per turn with the buffers swapping roles: no copy of the segment in flight
        if constexpr (!LATE) fetch2(B, j - 1);
        run2(A, j, B, j - 1);
        if constexpr (!LATE) fetch2(A, j - 2);
        run2(B, j - 1, A, j - 2);
    }
    if (j == 0) run2(A, 0, B, -1);
}

} // namespace l3d

using namespace l3d;

// false when no kernel variant is built for the lane-group shape
bool sgbm_launch_horizontal(const SgbmGeom &g, int B, const uint32_t *Cvol, const uint32_t *Tvol, const uint32_t *Mvol, uint32_t *ckpt,
                            uint32_t *wta, int16_t *dsub, hipStream_t s)
{
    const int nrows = B * g.H;
    return dispatch_variant(g.NPh, g.LPPh, g.Dp != g.D, [&](auto np, auto lpp, auto pd) {
        constexpr int NP = decltype(np)::value, LPP = decltype(lpp)::value;
        constexpr bool PAD = decltype(pd)::value;
        const dim3 grid(cdiv(nrows, 64 / LPP));
        if constexpr (!PAD) {
            if (g.tonly && g.uform) {
                hipLaunchKernelGGL((k_horizontal<NP, LPP, true, PAD, true>), grid, dim3(64), 0, s, g, nrows, Cvol, Tvol, Mvol, ckpt, wta, dsub);
                return;
            }
        }
        if (g.tonly)
            hipLaunchKernelGGL((k_horizontal<NP, LPP, true, PAD, false>), grid, dim3(64), 0, s, g, nrows, Cvol, Tvol, Mvol, ckpt, wta, dsub);
        else
            hipLaunchKernelGGL((k_horizontal<NP, LPP, false, PAD, false>), grid, dim3(64), 0, s, g, nrows, Cvol, Tvol, Mvol, ckpt, wta, dsub);
    });
}
