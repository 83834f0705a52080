// bb_lane.h — the fast barcode stage with ONE LANE PER FLANK HIT (searcher.rs:267-337 on one word per lane).
//
// k_barcode_pfx gives a hit's N barcodes to N lanes: everything a hit owns is built once per block iteration in LDS
// (per-column records, the walks through the shared rows, the top-2 cells) and the twelve waves of a block meet at two
// barriers per iteration — 41 % of that kernel's wave time is parked (profiles/valu_nbd96.json), at three waves per
// SIMD where and/or/add issue at 70 % of their rate.  Here a lane keeps ITS hit (window codes, carry-in bits of the
// shared rows, trailing-row masks) in registers and walks through the group's barcodes one after the other:
//   * the barcode index is wave-uniform, so a column's Eq word is one LDS read at  barcode * 64 + code * 4  (four
//     distinct words in four banks: a broadcast), the address being the column's code byte (kept as code << 2) added
//     to a scalar;
//   * the top-2 of both candidate sets are running maxima in registers — no cross-lane reduction, no atomics;
//   * no barrier after the tables are loaded, no per-iteration table builds; the bound needs no walk through the shared
//     rows (at most P Match columns ending where the cursor enters row P: see the trip's comment);
//   * the shared rows are the lane's own business: their DP runs once per hit in the prologue (carry-in masks in registers,
//     no prefix record read), and the final trip recomputes their move planes into registers for the winner's walk;
//   * the path of the best-bounded barcode is not carried along (14 registers and as many selects per barcode) but
//     recomputed in one extra trip of the same body with a per-lane barcode index — a second instantiation of the body
//     AFTER the barcode loop, so that what only it needs (the walk's planes, the exact replay's f64 state) does not raise
//     the register pressure of the loop;
//   * that final trip also decides (rows_decide: exact score of the winner's path against the runner-up's bound).
// Output = k_barcode_pfx<.., FAST = true> + k_rows': the hit's row, or its index on the fallback list of the exact kernel.
#pragma once
#include <type_traits>
#include "bb_k_bar_common.h"

#ifndef BB_LANE_NOHOIST
#define BB_LANE_NOHOIST 1
#endif

// The winner's walk through the shared rows, columns LO+1 .. HI (1-based): their move planes recomputed into HI - LO registers (the DP
// from column 1: a dozen instructions per column), then one step per column from HI down, for the lanes whose cursor is in that column.
template <uint32_t PRIO, int LO, int HI, int NW>
__device__ __forceinline__ void shared_rows_walk(const uint32_t* s_eqt, const uint32_t (&cw)[NW], int P, uint32_t pm, int wmax, uint32_t& bh, int32_t& col,
                                                 int32_t& ntext, uint32_t& dgh, uint32_t& mth, uint32_t (&pl_w)[2], uint32_t (&ph_w)[2]) {
    uint32_t shw[HI - LO];
    {
        uint32_t pv = P >= 32 ? 0xFFFFFFFFu : (1u << P) - 1u, mv = 0u;
#pragma unroll
        for (int c = 0; c < HI; ++c) {
            uint32_t w = 0u;
            if (c < wmax) {  // wave-uniform
                uint32_t hp, hm;
                shared_rows_column<PRIO>(PRIO, s_eqt[(cw[c >> 2] >> (8 * (c & 3) + 2)) & 0xFu] & 0xFFFFu, P, pv, mv, hp, hm, w);
            }
            if (c >= LO) shw[c - LO] = w;
        }
    }
#pragma unroll
    for (int c = HI; c > LO; --c) {
        if (c <= wmax) {
            const bool on = bh != 0u && col == c;
            const uint32_t w = shw[c - 1 - LO];
            const uint32_t Lr = w & 0xFFFFu, Hr = w >> 16;
            const uint32_t Dr = Lr & Hr;
            const uint32_t nb = on ? (((Dr + bh) & ~Dr) & pm) : 0u;
            const bool has = nb != 0u, lo = (Lr & nb) != 0u, hi = (Hr & nb) != 0u;
            pl_w[(c - 1) >> 5] |= lo ? 1u << ((c - 1) & 31) : 0u;
            ph_w[(c - 1) >> 5] |= hi ? 1u << ((c - 1) & 31) : 0u;
            const bool consume = has & !hi;
            dgh |= consume ? nb : 0u;
            mth |= (has && !lo && !hi) ? nb : 0u;   // rows matched (the debug build's check of the NM masks)
            if (on) {
                bh = consume ? ((nb << 1) & pm) : nb;
                ntext += has ? 1 : 0;
                col -= has ? 1 : 0;
                if (col < 1) bh = 0u;
            }
        }
    }
}

// The bound runs over the pattern's rows.  With one exponent for every row without a Match — min(eS, eD) — it costs 8 instructions per
// column of a walk and 8 table steps; where that exponent is 0 but Sub or Del rows do decay (lodhi=3:0.5:1110: the rows without a Match
// would take no time at all and a third of the hits stayed undecided) the rows go by class (lodhi_bound_table_entry4: 9 instructions per
// column, 16 table steps — 8 % of the stage, which is why e.g. 2211, whose single exponent decides 98 % of the hits, keeps the cheap form).
__device__ __forceinline__ bool lane_rows4(uint32_t e4) {
    const uint32_t eS = (e4 >> 8) & 0xFFu, eD = e4 >> 24;
    return eS != eD && (eS == 0u || eD == 0u);
}

// waves per SIMD the register budget is set for: the move planes are 2 x CW registers — 48 columns fit three waves (<= 168 VGPRs), 64 two
// PRIO: the class of the policy's traceback order (bb_prio.h): the move planes are one v_bitop3 each with the class's truth tables
#ifndef BB_LANE_LOWSKIP
#define BB_LANE_LOWSKIP 12   // the column groups from here down are walked only while a lane's cursor needs them
#endif
template <int CW, bool TAIL, uint32_t PRIO, bool NM>
__global__ __launch_bounds__(256, CW <= 48 ? 3 : 2) void k_barcode_lane(const uint8_t* __restrict__ tables, const bb_group_dev* __restrict__ groups, uint32_t g,
                                                      uint32_t strand, const bb_hit* __restrict__ hits, const uint32_t* __restrict__ hit_meta,
                                                      const uint32_t* __restrict__ hit_list, const uint32_t* __restrict__ list_cnt,
                                                      uint32_t n_hits_all, bb_rowtmp* __restrict__ rows, double min_score, double min_score_diff,
                                                      double margin, uint32_t* __restrict__ fb_lists, uint32_t list_stride, uint32_t* __restrict__ fb_cnt) {
    constexpr bool use_nm = NM;   // groups with large flank budgets: the bound by the walk's own Match columns (below)
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const uint32_t n_list = hit_list ? list_cnt[g] : n_hits_all;
    if (blockIdx.x * 256u >= n_list) return;
    const bb_group_dev& G = groups[g];
    const int N = G.n_seqs, m = G.m_bar, P = G.pfx[strand], T = TAIL ? G.tail[strand] : 0;
    const int32_t k1 = G.k1, k2 = G.k2;
    const int pol_lm = G.pol_lm;
    const bool tie_last = G.pol_tie_last != 0;
    // LDS: [Peq: N x 16 words, barcode-major][bound table 256 x 32 B][trailing-row planes]
    uint32_t* s_peq = reinterpret_cast<uint32_t*>(smem);
    size_t o = ((size_t)N * 64 + 31) & ~(size_t)31;
    bb_lb_entry* s_lb = reinterpret_cast<bb_lb_entry*>(smem + o);
    o += 256 * sizeof(bb_lb_entry);
    o = (o + 15) & ~(size_t)15;
    unsigned long long* s_tail = reinterpret_cast<unsigned long long*>(smem + o);  // [t][lo|hi][thread]
    o += (size_t)T * 2 * 256 * 8;
    uint4* s_hdr = reinterpret_cast<uint4*>(smem + o);   // [piece][thread]: the hits' 32-byte headers, kept for the final trip (read again from their
    o += 2 * 256 * sizeof(uint4);                        // records ~100 trips later they came from HBM a second time: ~0.1 KB per hit)
    // NM (groups with large flank budgets): per entry column, WHICH of the P shared rows the walk from there matches — [column][lane] 16-bit
    // masks (bit q <-> row q + 1).  The bound of a barcode then grants the shared rows exactly those Matches where its path enters row P
    // instead of all P: with k = 20 the flank hits are mostly chance hits whose pad rows match badly, and the all-P assumption left 8 x as
    // many hits undecided.
    uint16_t* s_nm = reinterpret_cast<uint16_t*>(smem + o);
    __shared__ uint32_t s_eqt[16];   // Peq of the leading shared rows per base set; trailing rows matched per base set in bits 16..
    if (threadIdx.x < 16u) {
        const uint32_t e = reinterpret_cast<const uint32_t*>(tables + G.off_peq_pfx[strand])[threadIdx.x];
        const uint32_t tl = (tables + G.off_tail_lut[strand])[threadIdx.x];
        s_eqt[threadIdx.x] = (e & 0xFFFFu) | (tl << 16);
    }
    {
        const uint32_t* gp = reinterpret_cast<const uint32_t*>(tables + G.off_peq_sub[strand]);  // [code][barcode]
        for (int i = threadIdx.x; i < 16 * N; i += 256) {
            const int code = i / N, p = i - code * N;
            s_peq[p * 16 + code] = gp[i];
        }
        // the bound runs over the pattern's ROWS (below): a row without a Match is a Sub or a Del, so the table's other-op exponent is min(eS, eD)
        const uint32_t e4 = (uint32_t)G.pol_lodhi_exp, e_rows = (e4 & 0xFFFFu) | ((e4 >> 24) << 16);
        if (lane_rows4(e4)) {   // the table by row classes (walks collect the rows left diagonally as well)
            for (uint32_t i = threadIdx.x; i < 256u; i += 256u) { bb_lb_entry e; lodhi_bound_table_entry4(i, e4, e); lb_put(s_lb, i, e); }
        } else {
            for (uint32_t i = threadIdx.x; i < 256u; i += 256u) { bb_lb_entry e; lodhi_bound_table_entry(i, e_rows, e); lb_put(s_lb, i, e); }
        }
    }
    const bool rows4 = lane_rows4((uint32_t)G.pol_lodhi_exp);   // wave-uniform
    // Which hit a lane takes: the block's 256 hits, those with windows of at most CW - 4 columns first.  A wave walks as many
    // column groups as its widest window needs; 99 % of the windows of SQK-NBD114-96 are 44 columns wide, but one 45-column window
    // among a wave's 64 costs all of them a twelfth group, forward and back.  Sorted, three waves in four skip it.
    __shared__ uint32_t s_perm[256];
    __shared__ uint32_t s_wcnt[8];
    uint32_t hit_idx;
    bool exists;
    {
        const uint32_t li = blockIdx.x * 256u + threadIdx.x;
        const bool ex0 = li < n_list;
        const uint32_t h0i = hit_list ? (ex0 ? hit_list[li] : 0u) : (ex0 ? li : 0u);
        const bool narrow = ex0 && ((hit_meta[h0i] >> 16) & 0xFFu) <= (uint32_t)(CW - BB_CG);   // the window's width from the hit's meta word, not its record
        const bool wide = ex0 && !narrow;
        const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
        const unsigned long long bn = __ballot(narrow), bw = __ballot(wide);
        if (lane == 0) { s_wcnt[wv] = (uint32_t)__popcll(bn); s_wcnt[4 + wv] = (uint32_t)__popcll(bw); }
        s_perm[threadIdx.x] = 0xFFFFFFFFu;
        __syncthreads();
        uint32_t base_n = 0u, base_w = 0u, tot_n = 0u;
#pragma unroll
        for (uint32_t q = 0; q < 4u; ++q) { base_n += q < wv ? s_wcnt[q] : 0u; base_w += q < wv ? s_wcnt[4 + q] : 0u; tot_n += s_wcnt[q]; }
        const unsigned long long below = (1ull << lane) - 1ull;
        if (narrow) s_perm[base_n + (uint32_t)__popcll(bn & below)] = h0i;
        if (wide) s_perm[tot_n + base_w + (uint32_t)__popcll(bw & below)] = h0i;
        __syncthreads();
        hit_idx = s_perm[threadIdx.x];
        exists = hit_idx != 0xFFFFFFFFu;
        if (!exists) hit_idx = 0u;
    }
    // ---- the lane's hit: header, window codes (as LDS byte offsets of their Peq word), carry-in bits, trailing rows ----
    uint32_t cw[CW / 4];
    int32_t wn;
    bool active;
    {
        const uint4* hp4 = reinterpret_cast<const uint4*>(hits + hit_idx);
        const uint4 h0 = hp4[0], h1 = hp4[1];
        s_hdr[threadIdx.x] = h0; s_hdr[256u + threadIdx.x] = h1;   // read back by the same lane only: no barrier
        const bool valid = (h1.z & 0xFFu) != 0u;
        active = exists && valid;
        if (exists && !valid) rows[hit_idx].row._pad[0] = 0;
        wn = active ? (int32_t)(h1.x - h0.w) : 0;   // we - ws
#pragma unroll
        for (int q = 0; q < CW / 16; ++q) {
            const uint4 w = hp4[2 + q];
            cw[4 * q] = (w.x & 0x0F0F0F0Fu) << 2; cw[4 * q + 1] = (w.y & 0x0F0F0F0Fu) << 2;
            cw[4 * q + 2] = (w.z & 0x0F0F0F0Fu) << 2; cw[4 * q + 3] = (w.w & 0x0F0F0F0Fu) << 2;
        }
    }
    uint32_t hpw[2] = {0u, 0u}, hmw[2] = {0u, 0u};
    unsigned long long TE[BB_MAX_TAIL];
#pragma unroll
    for (int q = 0; q < BB_MAX_TAIL; ++q) TE[q] = 0ull;
    int wmax = wn;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) wmax = max(wmax, __shfl_xor(wmax, d, 64));
    wmax = __builtin_amdgcn_readfirstlane(wmax);
    __syncthreads();  // the tables; the only barrier of the kernel
    // The leading shared rows once per hit, by the lane itself (k_bar_prefix's recurrence: 12 instructions per column against the 96 x 40
    // of the barcode loop): their carry-in masks and the trailing rows' match masks stay in registers, and no prefix record is read.
    {
        uint32_t pv = P ? (P >= 32 ? 0xFFFFFFFFu : (1u << P) - 1u) : 0u, mv = 0u;
#pragma unroll
        for (int c = 0; c < CW; ++c) {
            if (c < wmax) {  // wave-uniform
                const uint32_t e = s_eqt[(cw[c >> 2] >> (8 * (c & 3) + 2)) & 0xFu];
                const bool in = c < wn;
                if (TAIL) {
#pragma unroll
                    for (int q = 0; q < BB_MAX_TAIL; ++q) TE[q] |= (in && ((e >> (16 + q)) & 1u)) ? 1ull << c : 0ull;
                }
                if (P > 0) {
                    uint32_t hp, hm, shw;
                    shared_rows_column<PRIO>(PRIO, e & 0xFFFFu, P, pv, mv, hp, hm, shw);
                    hpw[c >> 5] |= in ? hp << (c & 31) : 0u;
                    hmw[c >> 5] |= in ? hm << (c & 31) : 0u;
                }
            }
        }
    }

    if (NM && P > 0) {  // wave-uniform
        // W[r] = the shared rows matched by the walk from (row r, column c) up to row 0 (bit q <-> row q + 1):
        //   Match: (W[r-1] of column c-1) | bit r-1;  Sub: W[r-1] of column c-1;  Ins: W[r] of column c-1;  Del: W[r-1] of column c.
        // A rolled loop over the columns (the hit's window codes re-read from its record, the shared rows' step again: a dozen instructions),
        // the rows unrolled with their masks in registers.
        uint32_t pv = P >= 32 ? 0xFFFFFFFFu : (1u << P) - 1u, mv = 0u;
        uint32_t Wp[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) Wp[r] = 0u;
        const uint8_t* win = reinterpret_cast<const uint8_t*>(hits + hit_idx) + 32;
        const int nmc = min(wmax, BB_LANE_NM_COLS);
#pragma unroll 1
        for (int c = 0; c < nmc; ++c) {
            const uint32_t code = active && c < wn ? (uint32_t)win[c] & 0xFu : 0u;
            uint32_t hp, hm, shw;
            shared_rows_column<PRIO>(PRIO, s_eqt[code] & 0xFFFFu, P, pv, mv, hp, hm, shw);
            uint32_t above = 0u, diag = 0u, out = 0u;
#pragma unroll
            for (int r = 1; r <= 16; ++r) {
                if (r <= P) {  // wave-uniform
                    const uint32_t l = (shw >> (P - r)) & 1u, hh = (shw >> (16 + P - r)) & 1u;   // the cell's move: row r <-> bit P - r of each plane
                    const uint32_t left = Wp[r - 1];
                    const uint32_t v = hh ? (l ? above : left) : (diag | (l ? 0u : 1u << (r - 1)));
                    diag = left;          // W[r][c-1] is the next row's diagonal
                    Wp[r - 1] = v;        // becomes W[r][c]
                    above = v;
                    if (r == P) out = v;
                }
            }
            s_nm[c * 256 + threadIdx.x] = (uint16_t)out;   // the walk that enters row P in column c + 1
        }
    }
    // eight or more shared rows granted as Matches (no NM masks, one exponent for the rows without a Match): the bound's first table step
    // is the same for every barcode of every hit
    const bool skip0 = !TAIL && !(use_nm && P > 0) && P >= 8;   // wave-uniform (not with trailing rows: that instantiation has no three registers to spare)
    float lb_sc0 = 0.0f, lb_u10 = 0.0f, lb_u20 = 0.0f;   // (read from the table in every trip instead, the three registers cost more than they hold: 8.96 against 8.88 ms)
    if (skip0) { if (rows4) lodhi_bound_first_rows4(s_lb, lb_sc0, lb_u10, lb_u20); else lodhi_bound_first_byte(s_lb, lb_sc0, lb_u10, lb_u20); }
    const uint8_t* s_peq_b = reinterpret_cast<const uint8_t*>(s_peq);
    // running top-2 of the two candidate sets (searcher.rs:303-328): bound bits + 1 (0 = empty), first maximum's barcode
    uint32_t b1A = 0u, b2A = 0u, pA = 0u, b1B = 0u, b2B = 0u, pB = 0u;
    uint32_t ptop = 0u;
    bool want = false;  // the final trip: this lane has a winner to trace again
    // one trip: barcode `it` for every lane, or (last) each lane's own winner again
    auto trip = [&](const int it, const bool last) __attribute__((always_inline)) {
        const uint32_t pbase = (last ? ptop : (uint32_t)it) * 64u;  // byte offset of the barcode's 16 Peq words
#if BB_LANE_NOHOIST
        // The per-column fields of these words (48 Peq offsets, 96 carry-in bits) are the same in every trip and the compiler keeps them
        // all in registers across the loop (256 VGPRs, two waves per SIMD).  Opaque to it, they are extracted where they are used:
        // three half-rate extractions more per column for ~130 registers less.
        asm volatile("" : "+v"(hpw[0]), "+v"(hpw[1]), "+v"(hmw[0]), "+v"(hmw[1]));
#pragma unroll
        for (int q = 0; q < CW / 4; ++q) asm volatile("" : "+v"(cw[q]));
#endif
        const bool on = last ? want : active;
        // ---- forward pass on the lane's own rows (one word), carry-in from the shared rows ----
        uint32_t L0[CW], H0[CW];
        int32_t best_cost = 0x7FFFFFFF, best_pos = -1;
        unsigned long long Pm, Mm;
        {
            uint32_t pv = 0xFFFFFFFFu, mv = 0u;
            uint32_t upr[2] = {0u, 0u}, dnr[2] = {0u, 0u};
            // the Eq words of a column group are requested a group ahead (before the wave-uniform branch that would otherwise keep
            // the scheduler from hoisting them): with two waves per SIMD an exposed LDS round trip halves the issue rate
            uint32_t eqn[BB_CG];
#pragma unroll
            for (int q = 0; q < BB_CG; ++q) eqn[q] = *reinterpret_cast<const uint32_t*>(s_peq_b + (pbase + ((cw[q >> 2] >> (8 * (q & 3))) & 0xFFu)));
#pragma unroll
            for (int c0 = 0; c0 < CW; c0 += BB_CG) {
                uint32_t eqc[BB_CG];
#pragma unroll
                for (int q = 0; q < BB_CG; ++q) eqc[q] = eqn[q];
                if (c0 + BB_CG < CW) {
#pragma unroll
                    for (int q = 0; q < BB_CG; ++q) {
                        const int c = c0 + BB_CG + q;
                        eqn[q] = *reinterpret_cast<const uint32_t*>(s_peq_b + (pbase + ((cw[c >> 2] >> (8 * (c & 3))) & 0xFFu)));
                    }
                }
                if (c0 < wmax) {  // wave-uniform
#pragma unroll
                    for (int c = c0; c < c0 + BB_CG; ++c) {
                        const uint32_t eq = eqc[c - c0];
                        const uint32_t hp = (hpw[c >> 5] >> (c & 31)) & 1u, hm = (hmw[c >> 5] >> (c & 31)) & 1u;
                        const uint32_t x = bitop3<0xC8>(eq, pv, hm);                       // (eq | hm) & pv
                        const uint32_t t = bitop3<BB_TT_XOR_OR>(x + pv, pv, eq);           // ((x + pv) ^ pv) | eq
                        const uint32_t d0 = bitop3<0xFE>(t, hm, mv);                       // t | hm | mv
                        const uint32_t ph = bitop3<BB_TT_OR_NOR>(mv, d0, pv), mh = bitop3<0xC0>(pv, d0, 0u);
                        uint32_t l, hh;
                        if constexpr (!bb_prio_needs_pvn(PRIO)) move_planes<PRIO>(d0, eq, ph, 0u, l, hh);  // Del last (the default): planes of (d0, eq, ph)
                        const unsigned long long tp = shl1_64(((unsigned long long)upr[c >> 5] << 32) | ph);
                        const unsigned long long tm = shl1_64(((unsigned long long)dnr[c >> 5] << 32) | mh);
                        upr[c >> 5] = (uint32_t)(tp >> 32); dnr[c >> 5] = (uint32_t)(tm >> 32);
                        const uint32_t nph = bitop3<0x01>((uint32_t)tp, hp, d0);           // ~(phs | d0)
                        mv = bitop3<0xA8>((uint32_t)tp, hp, d0);                           // phs & d0
                        pv = bitop3<0xFE>(nph, (uint32_t)tm, hm);                          // mhs | ~(d0 | phs)
                        if constexpr (bb_prio_needs_pvn(PRIO)) move_planes<PRIO>(d0, eq, ph, pv, l, hh);   // orders that test Del: its bit is the new column's vertical +1
                        L0[c] = __brev(l); H0[c] = __brev(hh);  // row P+1 <-> bit 31, row P+32 <-> bit 0
                    }
                }
            }
            const int pc = min(CW, ((wmax + BB_CG - 1) / BB_CG) * BB_CG);
            const int n0 = min(pc, 32), n1 = pc - n0;
            uint32_t up[2], dn[2];
            up[0] = n0 ? __brev(upr[0]) >> (32 - n0) : 0u; dn[0] = n0 ? __brev(dnr[0]) >> (32 - n0) : 0u;
            up[1] = n1 ? __brev(upr[1]) >> (32 - n1) : 0u; dn[1] = n1 ? __brev(dnr[1]) >> (32 - n1) : 0u;
            const unsigned long long wmask = wn >= 64 ? ~0ull : ((1ull << wn) - 1ull);
            Pm = (((unsigned long long)up[1] << 32) | up[0]) & wmask;   // horizontal deltas of row P+32
            Mm = (((unsigned long long)dn[1] << 32) | dn[0]) & wmask;
            if constexpr (TAIL) {  // the trailing shared rows, row-wise (see k_barcode_pfx)
#pragma unroll
                for (int t = 0; t < BB_MAX_TAIL; ++t) {
                    if (t < T) {
                        const unsigned long long Eq = TE[t];
                        const unsigned long long D0 = (((Eq & Pm) + Pm) ^ Pm) | Eq | Mm;
                        const unsigned long long Pvv = Mm | ~(D0 | Pm), Mvv = Pm & D0;
                        const unsigned long long Pvs = (Pvv << 1) | 1ull, Mvs = Mvv << 1;
                        const unsigned long long Ph = Mvs | ~(D0 | Pvs), Mh = Pvs & D0;
                        unsigned long long tl, th;  // row-wise: Ins tests the new row's horizontal +1 (Ph), Del the vertical +1 between the two rows (Pvv)
                        move_planes_any64<PRIO>(PRIO, D0, Eq, Ph, Pvv, tl, th);
                        s_tail[(size_t)(2 * t) * 256u + threadIdx.x] = tl;
                        s_tail[(size_t)(2 * t + 1) * 256u + threadIdx.x] = th;
                        Pm = Ph & wmask; Mm = Mh & wmask;
                    }
                }
            }
            pick_minimum(Pm, Mm, wn, m, on, pol_lm, tie_last, best_cost, best_pos);
        }
        const bool cand = on && best_pos >= 0 && best_cost <= k2;
        if (!__any(cand)) return;
        // ---- traceback, phase 0: the trailing shared rows ----
        unsigned long long plo = 0ull, phi = 0ull;
        uint32_t b = 0u, dg = 0u;
        int32_t c_ent = best_pos;
        int32_t tr = cand ? T - 1 : -1;
        uint32_t dgt = 0u, mtt = 0u;   // trailing rows left by a diagonal move / matched
        while (TAIL && __any(tr >= 0 && c_ent >= 1)) {
            const bool onn = tr >= 0 && c_ent >= 1;
            const int rr = onn ? tr : 0, sh = onn ? c_ent - 1 : 0;
            const unsigned long long l64 = s_tail[(size_t)(2 * rr) * 256u + threadIdx.x], h64 = s_tail[(size_t)(2 * rr + 1) * 256u + threadIdx.x];
            const uint32_t lo = (uint32_t)(l64 >> sh) & 1u, hi = (uint32_t)(h64 >> sh) & 1u;
            const bool del = onn && (lo & hi) != 0u, text = onn && !del, diag = onn && hi == 0u;
            plo |= text ? (unsigned long long)lo << sh : 0ull;
            phi |= text ? (unsigned long long)hi << sh : 0ull;
            dgt |= diag ? 1u << rr : 0u;
            mtt |= (text && (lo | hi) == 0u) ? 1u << rr : 0u;
            tr -= (del || diag) ? 1 : 0;
            c_ent -= text ? 1 : 0;
        }
        const unsigned long long smask = (cand && tr < 0 && c_ent >= 1) ? 1ull << (c_ent - 1) : 0ull;
        const uint32_t sm_w[2] = {(uint32_t)smask, (uint32_t)(smask >> 32)};
        // ---- phase 1: the lane's own rows, one-hot cursor (see k_barcode_pfx) ----
        uint32_t pl_acc[2] = {0u, 0u}, ph_acc[2] = {0u, 0u};
        // The lowest column groups only while some lane's cursor is still inside its rows (or yet to enter): the leading shared rows hold the
        // window's first ~P columns, so the last two or three groups are usually nobody's — a sixth of the walk.
        int rem = 0;   // columns left out at the low end (wave-uniform)
        uint32_t ncol = 0u, mrow = 0u;
        auto walk = [&](auto r4) __attribute__((always_inline)) {   // r4: a non-final walk that also collects the rows left diagonally (eS != eD)
            constexpr bool R4 = decltype(r4)::value;
#pragma unroll
            for (int c0 = CW; c0 >= BB_CG; c0 -= BB_CG) {
                if (c0 - (BB_CG - 1) <= wmax && rem == 0) {  // wave-uniform
                    if (c0 <= BB_LANE_LOWSKIP) {
                        if (!__any((b | (sm_w[0] & ((2u << (c0 - 1)) - 1u))) != 0u)) rem = c0;
                    }
                    if (rem == 0) {
                        // a lane's cursor enters the walk in ONE column (its end position): only the one or two groups that hold some lane's
                        // entry column extract the entry bit and add it in (two half-rate instructions per column)
                        auto group = [&](auto inj) __attribute__((always_inline)) {
                            constexpr bool INJ = decltype(inj)::value;
#pragma unroll
                            for (int c = c0; c > c0 - BB_CG; --c) {
                                const uint32_t Lr = L0[c - 1], Hr = H0[c - 1];
                                const uint32_t Dr = Lr & Hr;
                                const uint32_t sum = INJ ? Dr + b + ((sm_w[(c - 1) >> 5] >> ((c - 1) & 31)) & 1u) : Dr + b;
                                const uint32_t nb = bitop3<0x0C>(Dr, sum, 0u);  // ~Dr & sum
                                const uint32_t cm = bitop3<0x0C>(Hr, nb, 0u);  // ~Hr & nb
                                if (last) {   // the winner's walk: both planes of the path and its diagonal rows (rows_decide replays them)
                                    const uint32_t tl = Lr & nb, th = Hr & nb;
                                    pl_acc[(c - 1) >> 5] = __builtin_amdgcn_alignbit(pl_acc[(c - 1) >> 5], 0u - tl, 31);
                                    ph_acc[(c - 1) >> 5] = __builtin_amdgcn_alignbit(ph_acc[(c - 1) >> 5], 0u - th, 31);
                                    dg |= cm;
                                } else {      // every other walk: the bound wants the matched ROWS only (and, NM, the number of text columns)
                                    mrow |= bitop3<0x10>(nb, Lr, Hr);  // nb & ~Lr & ~Hr: the cursor's cell is a Match
                                    if (R4) dg |= cm;                  // Match or Sub: what is left of the path's rows are Dels
                                    if (NM) asm("v_bcnt_u32_b32 %0, %1, %0" : "+v"(ncol) : "v"(nb));   // ncol += popcount(nb): a text op per column the cursor is in
                                }
                                b = nb + cm;
                            }
                        };
                        const uint32_t gmask = ((1u << BB_CG) - 1u) << ((c0 - BB_CG) & 31);   // a group's columns lie in one word (BB_CG divides 32)
                        if (__any((sm_w[(c0 - 1) >> 5] & gmask) != 0u)) group(std::true_type{}); else group(std::false_type{});
                    }
                }
            }
        };
        if (!last && rows4) walk(std::true_type{}); else walk(std::false_type{});
        if (rem) { pl_acc[0] <<= rem; ph_acc[0] <<= rem; }   // rem <= BB_LANE_LOWSKIP < 32: the columns left out are all in word 0
        if (last) {
            plo |= ((unsigned long long)pl_acc[1] << 32) | pl_acc[0];
            phi |= ((unsigned long long)ph_acc[1] << 32) | ph_acc[0];
            int32_t ntext = cand ? __popc(dg) + __popc(dgt) + __popcll(phi & ~plo) : 0;
            const int32_t cx = cand ? best_pos - ntext : 0;
            // ---- phase 2: the shared rows (row r <-> bit P - r), walked on the lane's own prefix record ----
            uint32_t dgh = 0u, mth = 0u;
            if (P > 0) {
                // the shared rows' move planes again, into registers (the lane rows' planes are dead by now), and the walk column by
                // column from the top: a lane takes part from the column its cursor entered row P in.  Half the window at a time — 24
                // registers instead of 48 keep the kernel at three waves per SIMD without spills — and the upper half only if some
                // lane's cursor enters there (a winner's does not: its lane rows have consumed ~32 columns by then).
                const uint32_t pm = (1u << P) - 1u;  // P <= 16
                uint32_t bh = (cand && cx >= 1) ? 1u : 0u;
                int32_t col = cx;
                uint32_t pl_w[2] = {0u, 0u}, ph_w[2] = {0u, 0u};
                constexpr int HALF = CW / 2;
                if (__any(bh != 0u && col > HALF)) shared_rows_walk<PRIO, HALF, CW>(s_eqt, cw, P, pm, wmax, bh, col, ntext, dgh, mth, pl_w, ph_w);
                shared_rows_walk<PRIO, 0, HALF>(s_eqt, cw, P, pm, wmax, bh, col, ntext, dgh, mth, pl_w, ph_w);
                plo |= ((unsigned long long)pl_w[1] << 32) | pl_w[0];
                phi |= ((unsigned long long)ph_w[1] << 32) | ph_w[0];
            }
            const int32_t tstart = cand ? best_pos - ntext : 0;
            {   // what k_rows would do with the record, here (cand: always, for a lane with `want` — the same barcode was a candidate in its
                // own trip): the exact score of the winner's path, the decision by the runner-up's bound, the row or the fallback list
                const bool pass2 = b2A == 0u && k1 < k2;
                const uint32_t sx = pass2 ? b2B : b2A;
                bb_winrec W;
                W.plo = plo; W.phi = phi;
                W.diagrow = ((unsigned long long)__brev(dg) << P) | (P ? (unsigned long long)(__brev(dgh) >> (32 - P)) : 0ull) | ((unsigned long long)dgt << (P + 32));
                W.ub_second = sx ? (double)__uint_as_float(sx - 1u) / G.perfect : -1.0;   // -1: no other candidate
                W.tstart = (uint8_t)tstart; W.best_pos = (uint8_t)best_pos; W.top = (uint16_t)ptop;
                W.flags = 0; W.marker = 2; W._pad[0] = W._pad[1] = 0;
#ifdef BB_LANE_CHECK_NM
                // debug build only: the stored mask of the walk entering at cx against the rows the walk just done matched; a difference wrecks
                // the record, so that any parity test fails on it
                if (use_nm && P > 0 && cand && cx >= 1 && cx <= BB_LANE_NM_COLS) {
                    const uint32_t wm = (uint32_t)s_nm[(cx - 1) * 256 + threadIdx.x];
                    if (wm != (__brev(mth) >> (32 - P))) W.plo = ~0ull;
                }
#endif
                int bmax = cand ? best_pos : 0;
#pragma unroll
                for (int d = 32; d >= 1; d >>= 1) bmax = max(bmax, __shfl_xor(bmax, d, 64));
                bmax = __builtin_amdgcn_readfirstlane(bmax);
                const uint4 h0 = s_hdr[threadIdx.x], h1 = s_hdr[256u + threadIdx.x];   // (lanes without a candidate: ignored by rows_decide)
                rows_decide(cand, W, h0, h1, hit_idx, bmax, groups, rows, min_score, min_score_diff, margin, fb_lists, list_stride, fb_cnt);
                if (want && !cand) {  // cannot happen — the winner was a candidate in its own trip; should an edit ever break that, the hit goes to
                                      // the exact kernel instead of keeping whatever an earlier batch left in its slot
                    const uint32_t slot = 4u * g + (wn > 48 ? 2u : 0u) + strand;
                    const uint32_t at = atomicAdd(&fb_cnt[slot], 1u);
                    fb_lists[(size_t)slot * list_stride + at] = hit_idx;
                    rows[hit_idx].row._pad[0] = 0;
                }
            }
            return;
        }
        // The bound without the walk through the shared rows.  The cursor enters row P in column cx; whatever the walk does there, it
        // consumes at most P rows by a Match, in distinct columns <= cx.  The kernel's weights decay with the span of a subsequence
        // (lambda < 1 per column) and the bound's with the span in text columns, so of all placements of at most min(P, cx) Match
        // columns left of the cursor the contiguous run ending at cx gives every subsequence its shortest span: the bound of THAT
        // placement is an upper bound of the exact score — and equals the bound of the true path whenever the shared rows match
        // without gaps (the usual case: they are the flank the hit was found with).  Only the winner's walk is ever done (final trip).
        // The bound, over the pattern's rows.  The exact score runs over the path's op string; dropping ops from it (here: the Ins ops, which
        // have no row) shortens spans and can only raise it, as can turning ops into Matches.  So: the lane rows' and trailing rows' Match
        // bits as the walk found them, and the leading shared rows — not walked except by the winner — all P granted as Matches, or (NM)
        // exactly those the walk from the entry column matches.  [Through round 4 the bound ran over text columns: two planes of the path
        // collected per column and the entry column computed for every barcode — 12.5 instructions per column of the walk against 8.]
        unsigned long long rm = (unsigned long long)__brev(mrow) << P;   // row P + 1 <-> bit 31 of the lane's word
        if (TAIL) rm |= (unsigned long long)mtt << (P + 32);
        unsigned long long lead = low64(P), lead_any = low64(P);   // the shared rows matched / not known to be Dels
        if (use_nm && P > 0) {  // wave-uniform
            const int32_t cxq = cand ? c_ent - (int32_t)ncol : 0;   // the entry column: text columns left of the trailing rows' and the lane rows' ops
            lead = cxq >= 1 ? (cxq <= BB_LANE_NM_COLS ? (unsigned long long)s_nm[(cxq - 1) * 256 + threadIdx.x] : lead) : 0ull;
            lead_any = cxq >= 1 ? lead_any : 0ull;
        }
        float ubf;
        if (rows4) {  // wave-uniform: Sub and Del rows decay differently — the rows by class (bb_k_bar_common.h: Match = both masks, Sub = the
                      // second only, Del = neither; a shared row the walk did not match, or no walk is known for, = the first only: the
                      // smaller exponent.  With an entry column left of the window the shared rows are Dels: neither)
            unsigned long long bm = (unsigned long long)__brev(dg) << P;
            if (TAIL) bm |= (unsigned long long)dgt << (P + 32);
            const unsigned long long am = rm | lead_any;
            ubf = lodhi_bound_mask4<64>(cand ? am : 0ull, cand ? (bm | lead) : 0ull, m, s_lb, skip0, lb_sc0, lb_u10, lb_u20);
        } else
            ubf = lodhi_bound_mask<64>(cand ? (rm | lead) : 0ull, m, s_lb, skip0, lb_sc0, lb_u10, lb_u20);  // all bytes: no branches between the table reads
        const uint32_t v = __float_as_uint(ubf) + 1u;
        if (cand) {  // first maximum wins: strictly greater replaces
            if (v > b1B) { b2B = b1B; b1B = v; pB = (uint32_t)it; } else if (v > b2B) b2B = v;
            if (best_cost <= k1) { if (v > b1A) { b2A = b1A; b1A = v; pA = (uint32_t)it; } else if (v > b2A) b2A = v; }
        }
    };
#pragma unroll 1
    for (int it = 0; it < N; ++it) trip(it, false);
    {
        const bool pass2 = b2A == 0u && k1 < k2;
        const uint32_t mx = pass2 ? b1B : b1A;
        ptop = pass2 ? pB : pA;
        want = active && mx != 0u;
        if (__any(want)) trip(N, true);
    }
    if (active && !want) {  // no candidate at all: flank-only row (searcher.rs:353-362)
        const uint4 h0 = s_hdr[threadIdx.x], h1 = s_hdr[256u + threadIdx.x];
        bb_rowtmp R;
        bb_row& r = R.row;
        r.read_idx = h0.x; r.read_len = h1.w;
        r.rel_dist_to_end = rel_dist_to_end((int64_t)h0.y, (int64_t)h1.w);
        r.read_start_flank = h0.y; r.read_end_flank = h0.z;
        r.flank_cost = (int16_t)(h1.y & 0xFFFFu); r.group_idx = (uint8_t)((h1.y >> 16) & 0xFFu); r.strand = (uint8_t)(h1.y >> 24);
        r._pad[0] = 1; r._pad[1] = r._pad[2] = 0;
        r.read_start_bar = h0.y; r.read_end_bar = h0.z;
        r.bar_start = 0; r.bar_end = 0;
        r.match_type = (uint8_t)(G.type == BB_FTAG ? BB_FFLANK : BB_RFLANK);
        r.barcode_cost = (int16_t)G.m_bar; r.barcode_idx = -1;
        rows[hit_idx] = R;
    }
}
