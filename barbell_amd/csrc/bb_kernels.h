// bb_kernels.h — hand-written HIP kernels (gfx950 / CDNA4, wave64) of the annotate hot path.
//
// Pipeline per batch, all integer bit-twiddling, no MFMA:
//   flank scan     sassy's search of the N-masked flank over the whole read, both strands (searcher.rs:438): either
//                  k_flank_filter (15 or 31 rows of the flank, both strands in one pass over the text, flags per 16 bytes)
//                  + k_flank_verify (the full-height scan around flagged columns only), or k_flank_scan2 (the full-height
//                  scan of every column, one lane per (read, strand)).
//                  Local-minimum ends <= k are the raw flank hits.
//   scan           exclusive scan of per-(read,group,strand) hit counts -> deterministic slots.
//   k_flank_trace  one lane per flank hit: (m+k)-column DP with move bits, traceback,
//                  get_matching_region + window padding (cigar_parse.rs:71-82, searcher.rs:453-456).
//   barcode stage  one lane per (flank hit, barcode) (searcher.rs:267-426): k_bar_prefix (shared leading rows, once per
//                  hit), k_barcode_pfx (one Myers word per lane; fast variant: score bounds, top-2 per hit), k_rows (exact
//                  score of the best-bounded path, decision), the exact variant for undecided hits; k_barcode_reg /
//                  k_barcode for geometries outside the row split.
//   k_collapse     one lane per read: collapse_overlapping_matches (interval.rs:4-79).
//   scan + k_emit  compaction of surviving rows in read order + per-barcode histogram.
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>
#include <type_traits>

#include "../../include/barbell_amd_filter.h"
#include "../../include/barbell_amd_inspect.h"
#include "bb_common.h"
#include "bb_synth.h"

// ------------------------------------------------------------------------------------------------
// Myers / Hyyro column step on a W-word (32-bit) vertical bit-vector.  Row r (1-based) <-> bit r-1.
// pv/mv: vertical +1/-1 deltas of the previous column, updated in place to the new column.
// d0: diagonal-zero vector, ph/mh: horizontal deltas (before the shift), all for the new column.
// Top boundary row is all zero (text is free: D[0][i] = 0), so the horizontal carry-in is 0.
// ------------------------------------------------------------------------------------------------
// gfx950 three-input boolean: result bit = TT[(a << 2) | (b << 1) | c].  The compiler finds some of these on its
// own but leaves e.g. pv = mhs | ~(d0 | phs) as or + not + or; spelled out they are one instruction each.
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
template <int TT>
__device__ __forceinline__ uint32_t bitop3(uint32_t a, uint32_t b, uint32_t c) { return __builtin_amdgcn_bitop3_b32(a, b, c, TT); }
#define BB_TT_XOR_OR 0xBE    /* (a ^ b) | c   */
#define BB_TT_OR_NOR 0xF1    /* a | ~(b | c)  */

// 64-bit shift by one in ONE instruction (v_lshlrev_b64, half rate like v_lshlrev_b32 / v_alignbit_b32 — measured in
// profiles/valu_ceiling.json — but it does both words); left to itself the compiler splits it into lshl + alignbit
__device__ __forceinline__ unsigned long long shl1_64(unsigned long long x) {
    unsigned long long r;
    asm("v_lshlrev_b64 %0, 1, %1" : "=v"(r) : "v"(x));
    return r;
}
// 64-bit add in ONE instruction without the carry flag (the compiler's v_add_co / v_addc pair needs a wait state between
// its halves on gfx950 and both are half rate)
__device__ __forceinline__ unsigned long long add_64(unsigned long long x, unsigned long long y) {
    unsigned long long r;
    asm("v_lshl_add_u64 %0, %1, 0, %2" : "=v"(r) : "v"(x), "v"(y));
    return r;
}
// x << 1 as x + x: v_add_u32 issues at the full rate, v_lshlrev_b32 at half of it (profiles/valu_ceiling.json); written as inline
// assembly because the compiler canonicalises x + x back into the shift
__device__ __forceinline__ uint32_t shl1_32(uint32_t x) {
    uint32_t r;
    asm("v_add_u32 %0, %1, %1" : "=v"(r) : "v"(x));
    return r;
}
#ifndef BB_MYERS64
#define BB_MYERS64 1  // two-word step: carry chain as one 64-bit add (v_lshl_add_u64), the two shifts as v_lshlrev_b64
#endif
template <int W>
__device__ __forceinline__ void myers_step(uint32_t (&pv)[W], uint32_t (&mv)[W], const uint32_t (&eq)[W],
                                           uint32_t (&d0)[W], uint32_t (&ph)[W], uint32_t (&mh)[W]) {
    if constexpr (W == 2 && BB_MYERS64) {
        const unsigned long long x = ((unsigned long long)(eq[1] & pv[1]) << 32) | (eq[0] & pv[0]);
        const unsigned long long s = add_64(x, ((unsigned long long)pv[1] << 32) | pv[0]);
        d0[0] = bitop3<BB_TT_XOR_OR>((uint32_t)s, pv[0], eq[0]) | mv[0];
        d0[1] = bitop3<BB_TT_XOR_OR>((uint32_t)(s >> 32), pv[1], eq[1]) | mv[1];
        ph[0] = bitop3<BB_TT_OR_NOR>(mv[0], d0[0], pv[0]); ph[1] = bitop3<BB_TT_OR_NOR>(mv[1], d0[1], pv[1]);
        mh[0] = pv[0] & d0[0]; mh[1] = pv[1] & d0[1];
        const unsigned long long phs = shl1_64(((unsigned long long)ph[1] << 32) | ph[0]);
        const unsigned long long mhs = shl1_64(((unsigned long long)mh[1] << 32) | mh[0]);
        pv[0] = bitop3<BB_TT_OR_NOR>((uint32_t)mhs, d0[0], (uint32_t)phs); pv[1] = bitop3<BB_TT_OR_NOR>((uint32_t)(mhs >> 32), d0[1], (uint32_t)(phs >> 32));
        mv[0] = (uint32_t)phs & d0[0]; mv[1] = (uint32_t)(phs >> 32) & d0[1];
        return;
    }
    uint32_t phs[W], mhs[W];
    if constexpr (W >= 3 && W <= 4 && BB_MYERS64) {
        // words in pairs: 64-bit adds, the carry out of a pair from the majority function of its high words' bit 31
        // (x, pv, ~sum), the shifts as 64-bit shifts with one v_alignbit across the pair boundary
        uint32_t carry = 0;
#pragma unroll
        for (int w = 0; w < W; w += 2) {
            if (w + 1 < W) {
                const uint32_t x0 = eq[w] & pv[w], x1 = eq[w + 1] & pv[w + 1];
                unsigned long long sum = add_64(((unsigned long long)x1 << 32) | x0, ((unsigned long long)pv[w + 1] << 32) | pv[w]);
                if (w) sum = add_64(sum, (unsigned long long)carry);  // carry of the pair below (0/1)
                d0[w] = bitop3<BB_TT_XOR_OR>((uint32_t)sum, pv[w], eq[w]) | mv[w];
                d0[w + 1] = bitop3<BB_TT_XOR_OR>((uint32_t)(sum >> 32), pv[w + 1], eq[w + 1]) | mv[w + 1];
                carry = bitop3<0xD4>(x1, pv[w + 1], (uint32_t)(sum >> 32)) >> 31;  // (x & pv) | ((x | pv) & ~sum)
            } else {
                const uint32_t x = eq[w] & pv[w];
                const uint32_t sum = x + pv[w] + carry;
                d0[w] = bitop3<BB_TT_XOR_OR>(sum, pv[w], eq[w]) | mv[w];
            }
        }
#pragma unroll
        for (int w = 0; w < W; ++w) {
            ph[w] = bitop3<BB_TT_OR_NOR>(mv[w], d0[w], pv[w]);
            mh[w] = pv[w] & d0[w];
        }
#pragma unroll
        for (int w = 0; w < W; w += 2) {
            if (w + 1 < W) {
                const unsigned long long p2 = shl1_64(((unsigned long long)ph[w + 1] << 32) | ph[w]);
                const unsigned long long m2 = shl1_64(((unsigned long long)mh[w + 1] << 32) | mh[w]);
                phs[w] = (uint32_t)p2 | (w ? (ph[w - 1] >> 31) : 0u); phs[w + 1] = (uint32_t)(p2 >> 32);
                mhs[w] = (uint32_t)m2 | (w ? (mh[w - 1] >> 31) : 0u); mhs[w + 1] = (uint32_t)(m2 >> 32);
            } else {
                phs[w] = __builtin_amdgcn_alignbit(ph[w], ph[w - 1], 31);
                mhs[w] = __builtin_amdgcn_alignbit(mh[w], mh[w - 1], 31);
            }
        }
    } else {
    uint32_t carry = 0;
#pragma unroll
    for (int w = 0; w < W; ++w) {
        uint32_t x = eq[w] & pv[w];
        uint64_t s = (uint64_t)x + (uint64_t)pv[w] + (uint64_t)carry;
        carry = (uint32_t)(s >> 32);
        d0[w] = bitop3<BB_TT_XOR_OR>((uint32_t)s, pv[w], eq[w]) | mv[w];
    }
#pragma unroll
    for (int w = 0; w < W; ++w) {
        ph[w] = bitop3<BB_TT_OR_NOR>(mv[w], d0[w], pv[w]);
        mh[w] = pv[w] & d0[w];
    }
#pragma unroll
    for (int w = W - 1; w >= 0; --w) {
        phs[w] = (ph[w] << 1) | (w ? (ph[w - 1] >> 31) : 0u);
        mhs[w] = (mh[w] << 1) | (w ? (mh[w - 1] >> 31) : 0u);
    }
    }
#pragma unroll
    for (int w = 0; w < W; ++w) {
        pv[w] = bitop3<BB_TT_OR_NOR>(mhs[w], d0[w], phs[w]);
        mv[w] = phs[w] & d0[w];
    }
}

// Move bits of the traceback preference (oracle [H3]): at cell (row, column) with cost g
//   Match if diagonal-zero and characters match      (d0 & eq)
//   Ins   else if D[j][i-1] == g-1                    (ph)
//   Sub   else if D[j-1][i-1] == g-1                  (~d0)
//   Del   otherwise
// encoded as 2 bits per cell: 0 Match, 1 Sub, 2 Ins, 3 Del  ->  lo = Sub|Del, hi = Ins|Del.
template <int W>
__device__ __forceinline__ void move_bits(const uint32_t (&eq)[W], const uint32_t (&d0)[W], const uint32_t (&ph)[W],
                                          uint32_t (&lo)[W], uint32_t (&hi)[W]) {
#pragma unroll
    for (int w = 0; w < W; ++w) {  // both planes are three-input functions of (d0, eq, ph)
        lo[w] = bitop3<0x15>(d0[w], eq[w], ph[w]);  // ~((d0 & eq) | ph)
        hi[w] = bitop3<0x3A>(d0[w], eq[w], ph[w]);  // (ph & ~(d0 & eq)) | (lo & d0)
    }
}

// The same planes for any preference order (policy [H3]): prio holds the four ops, first choice in bits 0-1; an op is
// applicable at a cell iff  Match: d0 & eq,  Sub: ~d0 (diagonal is g-1),  Ins: ph (left is g-1),  Del: pvn, the NEW
// column's vertical +1 delta (above is g-1).  Used by the kernels that honour every policy (k_flank_trace, k_barcode);
// the default order takes the two-instruction form above.
template <int W>
__device__ __forceinline__ void move_bits_prio(uint32_t prio, const uint32_t (&eq)[W], const uint32_t (&d0)[W], const uint32_t (&ph)[W],
                                               const uint32_t (&pvn)[W], uint32_t (&lo)[W], uint32_t (&hi)[W]) {
    if (prio == (uint32_t)BB_PRIO_DEFAULT) { move_bits<W>(eq, d0, ph, lo, hi); return; }  // wave-uniform
#pragma unroll
    for (int w = 0; w < W; ++w) {
        const uint32_t vM = d0[w] & eq[w], vS = ~d0[w], vI = ph[w], vD = pvn[w];
        uint32_t taken = 0u, sS = 0u, sI = 0u, sD = 0u;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const uint32_t op = (prio >> (2 * q)) & 3u;
            const uint32_t v = (op == BB_OP_MATCH ? vM : op == BB_OP_SUB ? vS : op == BB_OP_INS ? vI : vD) & ~taken;
            taken |= v;
            sS |= op == BB_OP_SUB ? v : 0u; sI |= op == BB_OP_INS ? v : 0u; sD |= op == BB_OP_DEL ? v : 0u;
        }
        lo[w] = sS | sD; hi[w] = sI | sD;
    }
}

template <int W>
__device__ __forceinline__ uint32_t get_bit(const uint32_t (&v)[W], int bit) {
    uint32_t word = v[0];
#pragma unroll
    for (int w = 1; w < W; ++w) word = (bit >> 5) == w ? v[w] : word;
    return (word >> (bit & 31)) & 1u;
}

template <int W, int S>
__device__ __forceinline__ void load_eq(const uint32_t* tab, uint32_t c, uint32_t (&eq)[W]) {
    if constexpr (S == 2) {
        uint2 v = *reinterpret_cast<const uint2*>(tab + c * 2);
        eq[0] = v.x;
        if constexpr (W > 1) eq[1] = v.y;
    } else if constexpr (S == 4) {
        uint4 v = *reinterpret_cast<const uint4*>(tab + c * 4);
        eq[0] = v.x;
        if constexpr (W > 1) eq[1] = v.y;
        if constexpr (W > 2) eq[2] = v.z;
        if constexpr (W > 3) eq[3] = v.w;
    } else {
        const uint4 v = *reinterpret_cast<const uint4*>(tab + c * 8), u = *reinterpret_cast<const uint4*>(tab + c * 8 + 4);
        eq[0] = v.x; eq[1] = v.y; eq[2] = v.z; eq[3] = v.w;
        eq[4] = u.x;
        if constexpr (W > 5) eq[5] = u.y;
        if constexpr (W > 6) eq[6] = u.z;
        if constexpr (W > 7) eq[7] = u.w;
    }
}

// streaming local-minimum rule (policy [H1], include/barbell_amd_policy.h); evaluated lazily: only steps that touch the
// <= k zone matter, and entering the zone from above is a strict decrease, so `dec` and `cand` (the position of the
// last strict decrease: the left end of the plateau in progress) are always fresh when they are read.
struct lm_lane {
    int32_t prev;
    uint32_t dec;
    uint32_t nrep;
    uint32_t cand;
};

__device__ __forceinline__ void emit_hit(bb_hit_raw* hits, uint32_t cap, uint32_t* count, uint32_t read, uint32_t e,
                                         int32_t cost, uint32_t g, uint32_t strand, uint32_t ordinal) {
    uint32_t slot = atomicAdd(count, 1u);
    if (slot < cap) {
        bb_hit_raw h;
        h.read_idx = read; h.e = e; h.cost = (int16_t)cost; h.group = (uint8_t)g; h.strand = (uint8_t)strand; h.ordinal = ordinal;
        hits[slot] = h;
    }
}

// ------------------------------------------------------------------------------------------------
// k_flank_scan2: the production scan.  One lane = one (read, strand); grid.y = strand, so a block
// needs one strand's Peq table.  Reads are streamed from HBM in whole, 128-byte-aligned lines:
// each lane's next line is copied global->LDS with eight 16-byte LDS-DMA loads
// (global_load_lds_dwordx4: per-lane source address, wave-linear LDS destination, no VGPR staging),
// then consumed 16 bytes at a time with conflict-free ds_read_b128.  Every line of the batch is
// therefore requested from HBM exactly once per strand (round 1's first scan kernel, with per-lane 16-byte loads,
// re-fetched each line ~7x: profiles/r01_v1_pmc.txt).  The partial first/last line of a read is
// walked with byte loads.  The reverse-complement strand walks lines and bytes downwards.
// ------------------------------------------------------------------------------------------------
// Hits found by a lane are kept in registers (up to 4) and written once at the end of the read:
// a returning global atomic inside the column loop would park the whole wave for a memory round
// trip every time any lane reports (the dominant stall of the first version: profiles/r01_v2_pmc.txt).
struct hit_buf {
    uint32_t e0, e1, e2, e3;
    uint32_t costs;  // 4 x 8 bit
};
// lm_left / lm_strict: wave-uniform flags of the policy's rule (BB_LM_PLATEAU_LEFT / BB_LM_STRICT), in scope at every use
#define BB_LM_STEP_BUF(ST, CUR, IDX)                                                            \
    do {                                                                                        \
        int32_t cur_ = (CUR);                                                                   \
        if (min(cur_, ST.prev) <= kk) {                                                         \
            if (cur_ > ST.prev) {                                                               \
                if (ST.dec && ST.prev <= kk) {                                                  \
                    const uint32_t e_ = lm_left ? ST.cand : (IDX)-1u, k_ = ST.nrep;             \
                    if (k_ < 4u) {                                                              \
                        hb.e0 = k_ == 0u ? e_ : hb.e0; hb.e1 = k_ == 1u ? e_ : hb.e1;           \
                        hb.e2 = k_ == 2u ? e_ : hb.e2; hb.e3 = k_ == 3u ? e_ : hb.e3;           \
                        hb.costs |= ((uint32_t)ST.prev & 0xFFu) << (8u * k_);                   \
                    } else {                                                                    \
                        emit_hit(hits, hit_cap, hit_count, read, e_, ST.prev, g, (uint32_t)STRAND, k_); \
                    }                                                                           \
                    ST.nrep = k_ + 1u;                                                          \
                }                                                                               \
                ST.dec = 0;                                                                     \
            } else if (cur_ < ST.prev) {                                                        \
                ST.dec = 1; ST.cand = (IDX);                                                    \
            } else if (lm_strict) {                                                             \
                ST.dec = 0;                                                                     \
            }                                                                                   \
        }                                                                                       \
        ST.prev = cur_;                                                                         \
    } while (0)

#ifndef BB_VERIFY_CHUNKS
#define BB_VERIFY_CHUNKS 2   // 16-byte text loads per lane and round in k_flank_verify (4: 2.60 -> 1.96 GB of HBM traffic per step, but 4.67 -> 4.84 ms: lanes with short intervals idle through the longer rounds)
#endif
#define BB_VERIFY_FLW 12u     // flag words per lane cached in LDS by k_flank_verify (reads up to ~5.5 kb; longer ones read theirs from HBM)
#define BB_VERIFY_STAGE 128u  // hit records per wave in k_flank_verify's LDS staging area (a round with more goes out directly)
// wave-wide: the staged records go out with one atomic and 16-byte stores of consecutive lanes
__device__ __forceinline__ void stage_flush(const bb_hit_raw* stage, uint32_t fill, bb_hit_raw* __restrict__ hits, uint32_t hit_cap,
                                            uint32_t* __restrict__ hit_count) {
    const uint32_t lane = threadIdx.x & 63u;
    if (fill == 0u) return;
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    uint32_t base = 0u;
    if (lane == 0u) base = atomicAdd(hit_count, fill);
    base = (uint32_t)__shfl((int)base, 0, 64);
    for (uint32_t i = lane; i < fill; i += 64u)
        if (base + i < hit_cap) hits[base + i] = stage[i];
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
}
// End of a (read, strand) scan, shared by the streaming scan and the windowed verification: the right-overhang
// positions after the last column, the pending local minimum, the count, and the flush of the buffered hits.
template <int W, int STRAND>
__device__ __forceinline__ void scan_finish(bool live, uint32_t n, int m, int32_t kk, int32_t sc, uint32_t (&pv)[W], uint32_t (&mv)[W],
                                            uint32_t idx, lm_lane& st, hit_buf& hb, const int32_t* __restrict__ ovh, uint32_t read,
                                            uint32_t g, uint32_t n_groups, uint32_t* __restrict__ cnt, bb_hit_raw* __restrict__ hits,
                                            uint32_t hit_cap, uint32_t* __restrict__ hit_count, int pol_lm, bool at_end = true, int ovh_steps = 0x7FFFFFFF,
                                            bb_hit_raw* stage = nullptr, uint32_t* stage_fill = nullptr) {
    const uint32_t lane = threadIdx.x & 63u;
    const int TB = (m - 1) & 31;
    const bool lm_left = pol_lm == BB_LM_PLATEAU_LEFT, lm_strict = pol_lm == BB_LM_STRICT;
    // right overhang (oracle [H4]): C[n+o] = D[m-o][n] + floor(alpha*o), o = 1..m
    if (live) {
        int32_t d = sc;
        // positions beyond the last o with floor(alpha * o) <= k cost more than k: the first of them closes a pending minimum,
        // the rest change nothing (ovh_steps = that o + 1, capped at m)
        for (int o = 1; at_end && o <= m && o <= ovh_steps; ++o) {
            d -= (int32_t)((pv[W - 1] >> TB) & 1u) - (int32_t)((mv[W - 1] >> TB) & 1u);
#pragma unroll
            for (int w = W - 1; w >= 0; --w) {
                pv[w] = (pv[w] << 1) | (w ? (pv[w - 1] >> 31) : 0u);
                mv[w] = (mv[w] << 1) | (w ? (mv[w - 1] >> 31) : 0u);
            }
            ++idx;
            BB_LM_STEP_BUF(st, d + ovh[o], idx);
        }
        if (at_end && st.dec && st.prev <= kk) {
            const uint32_t e_ = lm_left ? st.cand : n + (uint32_t)m, k_ = st.nrep;
            if (k_ < 4u) {
                hb.e0 = k_ == 0u ? e_ : hb.e0; hb.e1 = k_ == 1u ? e_ : hb.e1;
                hb.e2 = k_ == 2u ? e_ : hb.e2; hb.e3 = k_ == 3u ? e_ : hb.e3;
                hb.costs |= ((uint32_t)st.prev & 0xFFu) << (8u * k_);
            } else {
                emit_hit(hits, hit_cap, hit_count, read, e_, st.prev, g, (uint32_t)STRAND, k_);
            }
            st.nrep = k_ + 1u;
        }
        cnt[((uint64_t)read * n_groups + g) * 2 + STRAND] = st.nrep;
    }
    // flush the buffered hits: one atomic per wave
    {
        const uint32_t mine = live ? min(st.nrep, 4u) : 0u;
        uint32_t incl = mine;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const uint32_t y = (uint32_t)__shfl_up((int)incl, d, 64); if ((int)lane >= d) incl += y; }
        const uint32_t total = (uint32_t)__shfl((int)incl, 63, 64);
        uint32_t base = 0;
        if (stage && total <= BB_VERIFY_STAGE) {
            // the wave's LDS staging area (BB_VERIFY_STAGE records): filled item by item, written out with one atomic when the
            // next item's hits would not fit (and by the caller at the end)
            uint32_t fill = *stage_fill;
            if (fill + total > BB_VERIFY_STAGE) { stage_flush(stage, fill, hits, hit_cap, hit_count); fill = 0u; }
            const uint32_t es[4] = {hb.e0, hb.e1, hb.e2, hb.e3};
#pragma unroll
            for (uint32_t k = 0; k < 4; ++k) {
                if (k < mine) {
                    bb_hit_raw h;
                    h.read_idx = read; h.e = es[k]; h.cost = (int16_t)((hb.costs >> (8u * k)) & 0xFFu);
                    h.group = (uint8_t)g; h.strand = (uint8_t)STRAND; h.ordinal = k;
                    stage[fill + incl - mine + k] = h;
                }
            }
            *stage_fill = fill + total;
        } else if (total) {
            if (lane == 0) base = atomicAdd(hit_count, total);
            base = (uint32_t)__shfl((int)base, 0, 64);
            uint32_t slot = base + incl - mine;
            const uint32_t es[4] = {hb.e0, hb.e1, hb.e2, hb.e3};
#pragma unroll
            for (uint32_t k = 0; k < 4; ++k) {
                if (k < mine && slot + k < hit_cap) {
                    bb_hit_raw h;
                    h.read_idx = read; h.e = es[k]; h.cost = (int16_t)((hb.costs >> (8u * k)) & 0xFFu);
                    h.group = (uint8_t)g; h.strand = (uint8_t)STRAND; h.ordinal = k;
                    hits[slot + k] = h;
                }
            }
        }
    }
}

// 2: line-aligned pieces, the partial first/last line of a read predicated (production); 1: pieces start at the read's
// own first byte (0.2 ms faster, but consecutive pieces share a 64-byte sector and half of the second requests miss
// L2: 25.1 instead of 16.5 GB per 2 M reads); 0: line-aligned pieces, partial lines walked with per-lane byte loops
#ifndef BB_SCAN_UNALIGNED
#define BB_SCAN_UNALIGNED 2
#endif
#ifndef BB_SCAN_LQ
#define BB_SCAN_LQ 8u  // 16-byte pieces per streamed line: 8 = 128-byte lines (8 KB of LDS per wave), 4 = 64-byte lines
#endif
template <int W, int STRAND>
__device__ __forceinline__ void flank_scan_lane(const uint8_t* __restrict__ bases, const uint64_t* __restrict__ offsets,
                                                uint32_t n_reads, const uint8_t* __restrict__ tables, int32_t kk, int m, int32_t score0,
                                                uint32_t off_pv0, uint32_t off_ovh, int ovh_steps, int pol_lm,
                                                uint32_t g, uint32_t n_groups, uint32_t* __restrict__ cnt,
                                                bb_hit_raw* __restrict__ hits, uint32_t hit_cap, uint32_t* __restrict__ hit_count,
                                                const uint32_t* s_peq, uint4* s_line /* this wave's [BB_SCAN_LQ][64] */) {
    constexpr int S = (W <= 2 ? 2 : (W <= 4 ? 4 : 8));
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t read = blockIdx.x * 256u + threadIdx.x;
    const bool live = read < n_reads;
    const uint64_t off = live ? offsets[read] : 0ull;
    const uint32_t n = live ? (uint32_t)(offsets[read + 1] - off) : 0u;
    const uint8_t* rb = bases + off;
    const int TB = (m - 1) & 31;
    const uint32_t* pv0 = reinterpret_cast<const uint32_t*>(tables + off_pv0);
    const int32_t* ovh = reinterpret_cast<const int32_t*>(tables + off_ovh);

    uint32_t pv[W], mv[W];
#pragma unroll
    for (int w = 0; w < W; ++w) { pv[w] = pv0[w]; mv[w] = 0; }
    int32_t sc = score0;
    lm_lane st = {score0, 1u, 0u, 0u};
    hit_buf hb = {0u, 0u, 0u, 0u, 0u};
    uint32_t idx = 0;  // scan position (characters consumed)
    const bool lm_left = pol_lm == BB_LM_PLATEAU_LEFT, lm_strict = pol_lm == BB_LM_STRICT;

    auto step = [&](uint32_t ch) {
        uint32_t eq[W], d0[W], ph[W], mh[W];
        load_eq<W, S>(s_peq, ch, eq);
        myers_step<W>(pv, mv, eq, d0, ph, mh);
        sc += (int32_t)((ph[W - 1] >> TB) & 1u) - (int32_t)((mh[W - 1] >> TB) & 1u);
        ++idx;
        BB_LM_STEP_BUF(st, sc, idx);
    };
    // Fast path: the bottom-row score moves by at most 1 per column, so while it is more than 4 above
    // k no position of the next 4 columns can be reported and neither the score nor the local-minimum
    // state needs tracking; the exact score is re-derived from the vertical deltas afterwards:
    // D[m][i] = popcount(Pv) - popcount(Mv) (top row is 0).
    const uint32_t topmask = TB == 31 ? 0xFFFFFFFFu : ((2u << TB) - 1u);
    auto score_now = [&]() {
        int32_t v = 0;
#pragma unroll
        for (int w = 0; w < W; ++w) {
            const uint32_t msk = w == W - 1 ? topmask : 0xFFFFFFFFu;
            v += __popc(pv[w] & msk) - __popc(mv[w] & msk);
        }
        return v;
    };
    auto step_fast = [&](uint32_t ch) {
        uint32_t eq[W], d0[W], ph[W], mh[W];
        load_eq<W, S>(s_peq, ch, eq);
        myers_step<W>(pv, mv, eq, d0, ph, mh);
    };

    // geometry of the walk in forward byte coordinates [0, n)
    const uint64_t a0 = (uint64_t)(uintptr_t)rb;
    constexpr uint32_t LB = BB_SCAN_LQ * 16u, LSH = BB_SCAN_LQ == 16 ? 8u : BB_SCAN_LQ == 8 ? 7u : 6u;  // line bytes (256, 128 or 64)
    // one group of 4 columns of the 16-byte piece v, starting at byte b0 (scan order): wave-uniform choice of path;
    // sc is exact on entry (either stepped or re-derived)
    auto group4 = [&](const uint4& v, int b0) {
        if (__any(sc <= kk + 4)) {
#pragma unroll
            for (int b = b0; b < b0 + 4; ++b) {
                const int bb = STRAND == 0 ? b : 15 - b;
                const uint32_t word = (bb >> 2) == 0 ? v.x : (bb >> 2) == 1 ? v.y : (bb >> 2) == 2 ? v.z : v.w;
                step((word >> (8 * (bb & 3))) & 0xFFu);
            }
        } else {
#pragma unroll
            for (int b = b0; b < b0 + 4; ++b) {
                const int bb = STRAND == 0 ? b : 15 - b;
                const uint32_t word = (bb >> 2) == 0 ? v.x : (bb >> 2) == 1 ? v.y : (bb >> 2) == 2 ? v.z : v.w;
                step_fast((word >> (8 * (bb & 3))) & 0xFFu);
            }
            idx += 4;
            sc = score_now();
            st.prev = sc;  // > k: the lazily evaluated `dec` needs no update (see lm_lane)
        }
    };
#if BB_SCAN_UNALIGNED == 2
    // Line-aligned streaming: the lane's lines are the LB-byte-aligned lines of HBM that hold its read, in scan order;
    // `mis` bytes of the first line (scan order) lie before the read's first scanned byte, and the last line may end
    // early.  Those two partial lines go through the same LDS path with the bytes outside the read predicated off, so
    // every line of the batch is requested once per strand and no lane runs a byte loop of its own.  (A line that
    // holds one byte of the read lies in that byte's page: the bytes outside the read are fetched, never used.)
    const uint32_t mis = STRAND == 0 ? (uint32_t)(a0 & (LB - 1u)) : (uint32_t)((LB - (uint32_t)((a0 + n) & (LB - 1u))) & (LB - 1u));
    const uint32_t nlines = n ? (mis + n + LB - 1u) >> LSH : 0u;
    const uint8_t* line0 = STRAND == 0 ? rb - mis : rb + n + mis - LB;  // first line in scan order
    uint32_t lmax = nlines;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) lmax = max(lmax, (uint32_t)__shfl_xor((int)lmax, d, 64));
    lmax = __builtin_amdgcn_readfirstlane(lmax);
    for (uint32_t l = 0; l < lmax; ++l) {
        const bool on = l < nlines;
        if (on) {
            const uint8_t* src = STRAND == 0 ? line0 + (l << LSH) : line0 - (l << LSH);
#pragma unroll
            for (int q = 0; q < (int)BB_SCAN_LQ; ++q)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + 16 * q),
                                                 (__attribute__((address_space(3))) void*)(s_line + 64 * q), 16, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        // scan-order byte range of this line that belongs to the read
        const uint32_t lo = l == 0u ? mis : 0u;
        const uint32_t hi = on ? min(LB, mis + n - (l << LSH)) : 0u;
        if (!__any(on && (lo != 0u || hi != LB))) {
            if (on) {
                for (int q = 0; q < (int)BB_SCAN_LQ; ++q) {
                    const uint4 v = s_line[64 * (STRAND == 0 ? q : (int)BB_SCAN_LQ - 1 - q) + lane];
#pragma unroll
                    for (int b0 = 0; b0 < 16; b0 += 4) group4(v, b0);
                }
            }
        } else {  // a partial line somewhere in the wave: every column tracked, bytes outside the read skipped
            for (int q = 0; q < (int)BB_SCAN_LQ; ++q) {
                const uint4 v = s_line[64 * (STRAND == 0 ? q : (int)BB_SCAN_LQ - 1 - q) + lane];
#pragma unroll
                for (int b = 0; b < 16; ++b) {
                    const int bb = STRAND == 0 ? b : 15 - b;
                    const uint32_t word = (bb >> 2) == 0 ? v.x : (bb >> 2) == 1 ? v.y : (bb >> 2) == 2 ? v.z : v.w;
                    const uint32_t p = 16u * (uint32_t)q + (uint32_t)b;
                    if (p >= lo && p < hi) step((word >> (8 * (bb & 3))) & 0xFFu);
                }
            }
        }
    }
#else
    uint32_t head, nlines;
#if BB_SCAN_UNALIGNED
    (void)a0;
    head = 0u;  // lines start at the read's first (last) byte whatever its alignment: no per-lane head loop
#else
    if (STRAND == 0) head = (uint32_t)((LB - (uint32_t)(a0 & (LB - 1u))) & (LB - 1u));
    else head = (uint32_t)((a0 + n) & (LB - 1u));
    if (head > n) head = n;
#endif
    nlines = (n - head) >> LSH;
    const uint32_t tail = n - head - (nlines << LSH);

    // partial first line
    for (uint32_t t = 0; t < head; ++t) step(STRAND == 0 ? rb[t] : rb[n - 1 - t]);
    // whole lines through LDS
    uint32_t lmax = nlines;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) lmax = max(lmax, (uint32_t)__shfl_xor((int)lmax, d, 64));
    lmax = __builtin_amdgcn_readfirstlane(lmax);
    for (uint32_t l = 0; l < lmax; ++l) {
        const bool on = l < nlines;
        if (on) {
            const uint8_t* src = STRAND == 0 ? rb + head + (l << LSH) : rb + (n - head - ((l + 1) << LSH));
#pragma unroll
            for (int q = 0; q < (int)BB_SCAN_LQ; ++q)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + 16 * q),
                                                 (__attribute__((address_space(3))) void*)(s_line + 64 * q), 16, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (on) {
            for (int q = 0; q < (int)BB_SCAN_LQ; ++q) {
                const uint4 v = s_line[64 * (STRAND == 0 ? q : (int)BB_SCAN_LQ - 1 - q) + lane];
#pragma unroll
                for (int b0 = 0; b0 < 16; b0 += 4) group4(v, b0);
            }
        }
    }
    // partial last line
    for (uint32_t t = 0; t < tail; ++t) step(STRAND == 0 ? rb[head + (nlines << LSH) + t] : rb[tail - 1 - t]);
#endif

    scan_finish<W, STRAND>(live, n, m, kk, sc, pv, mv, idx, st, hb, ovh, read, g, n_groups, cnt, hits, hit_cap, hit_count, pol_lm, true, ovh_steps);
}

template <int W>
__global__ __launch_bounds__(256) void k_flank_scan2(const uint8_t* __restrict__ bases, const uint64_t* __restrict__ offsets,
                                                     uint32_t n_reads, const uint8_t* __restrict__ tables,
                                                     const bb_group_dev* __restrict__ groups, uint32_t g, uint32_t n_groups,
                                                     uint32_t* __restrict__ cnt, bb_hit_raw* __restrict__ hits,
                                                     uint32_t hit_cap, uint32_t* __restrict__ hit_count) {
    constexpr int S = (W <= 2 ? 2 : (W <= 4 ? 4 : 8));
    __shared__ __attribute__((aligned(16))) uint32_t s_peq[256 * S];
    __shared__ __attribute__((aligned(16))) uint4 s_lines[4][BB_SCAN_LQ * 64];
    const bb_group_dev* G = groups + g;
    const uint32_t strand = blockIdx.y;
    {
        const uint32_t* src = reinterpret_cast<const uint32_t*>(tables + G->off_peq_flank[strand]);
        for (int i = threadIdx.x; i < 256 * S; i += 256) s_peq[i] = src[i];
    }
    __syncthreads();
    uint4* line = s_lines[threadIdx.x >> 6];
    const int32_t kk = G->flank_k, score0 = G->score0;
    const int m = G->m;
    const uint32_t o_pv0 = G->off_pv0, o_ovh = G->off_ovh;
    if (strand == 0)
        flank_scan_lane<W, 0>(bases, offsets, n_reads, tables, kk, m, score0, o_pv0, o_ovh, G->ovh_steps, G->pol_lm, g, n_groups, cnt, hits, hit_cap, hit_count, s_peq, line);
    else
        flank_scan_lane<W, 1>(bases, offsets, n_reads, tables, kk, m, score0, o_pv0, o_ovh, G->ovh_steps, G->pol_lm, g, n_groups, cnt, hits, hit_cap, hit_count, s_peq, line);
}

// ------------------------------------------------------------------------------------------------
// Filtered scan (groups with bb_group_dev::filt_rows > 0): Ukkonen's cut-off — rows below the last cell <= k of a
// column need not be computed — restructured for lanes that cannot diverge cheaply.
//
//   k_flank_filter  one lane per READ, one pass over the text for BOTH strands: Myers on R <= 15 consecutive rows
//                   u..u+R-1 of the flank alone (their own semi-global problem), the forward strand's right-aligned
//                   under bit 15 and the reverse-complement strand's under bit 31 of ONE 32-bit word (carries die in
//                   the guard bits 15 and 31).  Exact matching of a sub-pattern is direction-free: the rc strand's rows
//                   against the reversed text are the reversed rows against the forward text, so its block simply
//                   holds the rows in reverse order.  The lane tracks D[R][i] of both blocks and records, per 16-byte
//                   piece of each streamed line, whether it was ever <= k (one bit per piece and strand, 4 lines to a
//                   word; a read's words sit at (offset >> 9) + 3 * read, the word after them holds the rc-begin hint).
//   k_flank_verify  lanes draw (read, strand) items from a queue: the full-height scan of k_flank_scan2 — same step,
//                   same local-minimum state machine, same overhang handling and hit buffering — but only over the
//                   columns where a hit is possible: a match of cost c <= k ending at column e holds an alignment of
//                   rows u..u+R-1 of cost <= c ending at some column b (so b is flagged) with e - b in
//                   [m-u-R-k, m-u-R+k]; the read's ends are scanned where bb_group_dev::filt_mode or the flags near them
//                   ask for it (left / right overhang).  Each interval is entered with m+k columns of lead-in from the
//                   all-insertions column (values <= k are exact after that, larger ones stay > k — the argument of
//                   k_flank_trace), and the state machine only ever acts on values <= k or on the step into / out of
//                   them, so it emits exactly the hits of the full scan.
// The reads are streamed once instead of twice and more than half of the scan's instructions go away; where no window
// says enough (k close to R: the score is <= k everywhere) the host keeps the full scan (upload_tables).
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint64_t filt_word_base(uint64_t off, uint64_t off0, uint32_t read) { return ((off - off0) >> 9) + 3ull * read; }

// WIDE: windows of up to 31 rows, one word per strand (two Myers words per column: ~37 instructions instead of ~20) — for
// flanks whose 15-row windows say too little at the group's k but whose 31-row windows do (upload_tables decides).
template <bool WIDE>
__global__ __launch_bounds__(256) void k_flank_filter(const uint8_t* __restrict__ bases, const uint64_t* __restrict__ offsets, uint32_t n_reads,
                                                      const uint8_t* __restrict__ tables, const bb_group_dev* __restrict__ groups, uint32_t g,
                                                      uint32_t* __restrict__ flags, uint64_t words_per_strand, unsigned long long* __restrict__ n_flagged) {
    __shared__ uint32_t s_fpeq[WIDE ? 512 : 256];
    __shared__ __attribute__((aligned(16))) uint4 s_lines[4][BB_SCAN_LQ * 64];
    static_assert(BB_SCAN_LQ == 8u, "piece bits assume 128-byte lines");
    const bb_group_dev* G = groups + g;
    const int R = G->filt_rows;
    const int32_t kk = min(G->flank_k, R);  // k >= R: every column qualifies
    // blocks right-aligned under the guard bits 15 and 31: forward rows at bits 15-R..14, rc rows (reversed) at bits 31-R..30
    // (WIDE: each strand's rows at bits 0..R-1 of its own word, R <= 31)
    const uint32_t maskR = (1u << R) - 1u, SA = WIDE ? 0u : 15u - (uint32_t)R, BM = WIDE ? maskR : (maskR << SA) | (maskR << (SA + 16u));
    {
        const uint32_t S = G->W <= 2 ? 2u : (G->W <= 4 ? 4u : 8u);
        const uint32_t* f = reinterpret_cast<const uint32_t*>(tables + G->off_peq_flank[0]);
        const uint32_t* r = reinterpret_cast<const uint32_t*>(tables + G->off_peq_flank[1]);
        const uint32_t c = threadIdx.x, u = (uint32_t)G->filt_off, uw = u >> 5, ub = u & 31u;
        auto rows = [&](const uint32_t* t) {  // rows u .. u+R-1 of entry c
            const uint32_t lo = t[c * S + uw], hi = ub && uw + 1u < (uint32_t)G->W ? t[c * S + uw + 1u] : 0u;
            return ((lo >> ub) | (ub ? hi << (32u - ub) : 0u)) & maskR;
        };
        if constexpr (WIDE) { s_fpeq[2 * c] = rows(f); s_fpeq[2 * c + 1] = __brev(rows(r)) >> (32 - R); }
        else s_fpeq[c] = (rows(f) << SA) | ((__brev(rows(r)) >> (32 - R)) << (SA + 16u));
    }
    __syncthreads();
    uint4* s_line = s_lines[threadIdx.x >> 6];
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t read = blockIdx.x * 256u + threadIdx.x;
    const bool live = read < n_reads;
    const uint64_t off0 = offsets[0];
    const uint64_t off = live ? offsets[read] : off0;
    const uint32_t n = live ? (uint32_t)(offsets[read + 1] - off) : 0u;
    const uint8_t* rb = bases + off;
    constexpr uint32_t LB = 128u, LSH = 7u;
    const uint32_t mis = (uint32_t)((uint64_t)(uintptr_t)rb & (LB - 1u));
    const uint32_t nlines = n ? (mis + n + LB - 1u) >> LSH : 0u;
    const uint8_t* line0 = rb - mis;
    uint32_t* fl0 = flags + filt_word_base(off, off0, read);
    uint32_t* fl1 = fl0 + words_per_strand;

    // The forward block of a window that starts at row 0 is rows 1..R of the scan's own matrix — column 0 included, i.e. the
    // left-overhang column (G->off_pv0, floor(alpha * R)) — so matches that hang over the read's start are flagged like any
    // other.  Every other block is the window's own semi-global problem (column 0: D[j][0] = j).
    const bool own_rows = (G->filt_mode & BB_FILT_TRUE_INIT) != 0;
    const int32_t* ovh = reinterpret_cast<const int32_t*>(tables + G->off_ovh);
    const uint32_t pvA0 = own_rows ? reinterpret_cast<const uint32_t*>(tables + G->off_pv0)[0] & maskR : maskR;
    const int scA0 = own_rows ? (int)__popc(pvA0) : R;
    uint32_t pv = WIDE ? pvA0 : (pvA0 << SA) | (maskR << (SA + 16u)), mv = 0u;
    uint32_t pvB = maskR, mvB = 0u;  // WIDE: the rc strand's word
    // Both blocks' D[R][i], biased by 15 - k, in the two halves of one register (the bottom rows' delta bits sit at bits 14
    // and 30: one mask, one shift): a half's bit 4 is clear exactly while its score is <= k, so AND-ing the register over
    // the columns of a piece leaves bit 4 / bit 20 clear iff the piece holds such a column.  (WIDE: one register per strand,
    // bias 31 - k, bit 5.)
    const uint32_t TOPS = 0x40004000u;
    const int bias = (WIDE ? 31 : 15) - kk;
    uint32_t sc2 = WIDE ? (uint32_t)(scA0 + bias) : ((uint32_t)(R + bias) << 16) | (uint32_t)(scA0 + bias);
    uint32_t scB = (uint32_t)(R + bias);
    uint32_t keep = WIDE ? sc2 | ~0x20u : sc2 | ~0x00100010u;  // column 0 counts for the first piece
    uint32_t keepB = scB | ~0x20u;
    uint32_t bitsA = 0u, bitsB = 0u, nflag = 0u;
    auto step = [&](uint32_t chr) {
        if constexpr (WIDE) {
            const uint2 e2 = *reinterpret_cast<const uint2*>(s_fpeq + 2u * chr);
            {
                const uint32_t eq = e2.x, x = eq & pv;
                const uint32_t d0 = bitop3<BB_TT_XOR_OR>(x + pv, pv, eq) | mv;
                const uint32_t ph = bitop3<BB_TT_OR_NOR>(mv, d0, pv) & BM, mh = pv & d0;
                sc2 = sc2 + (ph >> (R - 1)) - (mh >> (R - 1));
                keep &= sc2;
                const uint32_t phs = shl1_32(ph), mhs = shl1_32(mh);
                pv = bitop3<BB_TT_OR_NOR>(mhs, d0, phs) & BM;
                mv = phs & d0;
            }
            {
                const uint32_t eq = e2.y, x = eq & pvB;
                const uint32_t d0 = bitop3<BB_TT_XOR_OR>(x + pvB, pvB, eq) | mvB;
                const uint32_t ph = bitop3<BB_TT_OR_NOR>(mvB, d0, pvB) & BM, mh = pvB & d0;
                scB = scB + (ph >> (R - 1)) - (mh >> (R - 1));
                keepB &= scB;
                const uint32_t phs = shl1_32(ph), mhs = shl1_32(mh);
                pvB = bitop3<BB_TT_OR_NOR>(mhs, d0, phs) & BM;
                mvB = phs & d0;
            }
        } else {
            const uint32_t eq = s_fpeq[chr];
            const uint32_t x = eq & pv;
            const uint32_t d0 = bitop3<BB_TT_XOR_OR>(x + pv, pv, eq) | mv;
            const uint32_t ph = bitop3<BB_TT_OR_NOR>(mv, d0, pv) & BM, mh = pv & d0;
            sc2 = sc2 + ((ph & TOPS) >> 14) - ((mh & TOPS) >> 14);
            keep &= sc2;
            const uint32_t phs = shl1_32(ph), mhs = shl1_32(mh);
            pv = bitop3<BB_TT_OR_NOR>(mhs, d0, phs) & BM;
            mv = phs & d0;
        }
    };
    auto commit = [&](uint32_t bit) {  // end of a piece
        if constexpr (WIDE) {
            bitsA |= ((~keep >> 5) & 1u) << bit; bitsB |= ((~keepB >> 5) & 1u) << bit;
            keepB = 0xFFFFFFFFu;
        } else {
            bitsA |= ((~keep >> 4) & 1u) << bit; bitsB |= ((~keep >> 20) & 1u) << bit;
        }
        keep = 0xFFFFFFFFu;
    };
    uint32_t lmax = nlines;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) lmax = max(lmax, (uint32_t)__shfl_xor((int)lmax, d, 64));
    lmax = __builtin_amdgcn_readfirstlane(lmax);
    for (uint32_t l = 0; l < lmax; ++l) {
        const bool on = l < nlines;
        if (on) {
            const uint8_t* src = line0 + (l << LSH);
#pragma unroll
            for (int q = 0; q < 8; ++q)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + 16 * q),
                                                 (__attribute__((address_space(3))) void*)(s_line + 64 * q), 16, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const uint32_t lo = l == 0u ? mis : 0u;
        const uint32_t hi = on ? min(LB, mis + n - (l << LSH)) : 0u;
        const uint32_t qb = (l & 3u) * 8u;  // bit of this line's first 16-byte piece
        if (!__any(on && (lo != 0u || hi != LB))) {
            if (on) {
                for (int q = 0; q < 8; ++q) {
                    const uint4 v = s_line[64 * q + lane];
#pragma unroll
                    for (int b = 0; b < 16; ++b) {
                        const uint32_t word = (b >> 2) == 0 ? v.x : (b >> 2) == 1 ? v.y : (b >> 2) == 2 ? v.z : v.w;
                        step((word >> (8 * (b & 3))) & 0xFFu);
                    }
                    commit(qb + (uint32_t)q);
                }
            }
        } else {  // a partial line somewhere in the wave: bytes outside the read skipped
            for (int q = 0; q < 8; ++q) {
                const uint4 v = s_line[64 * q + lane];
#pragma unroll
                for (int b = 0; b < 16; ++b) {
                    const uint32_t word = (b >> 2) == 0 ? v.x : (b >> 2) == 1 ? v.y : (b >> 2) == 2 ? v.z : v.w;
                    const uint32_t p = 16u * (uint32_t)q + (uint32_t)b;
                    if (p >= lo && p < hi) step((word >> (8 * (b & 3))) & 0xFFu);
                }
                // a piece without a byte of the read leaves `keep` alone: what column 0 says (the left-overhang column of a
                // window that starts at row 0) belongs to the first piece that holds read bytes, whichever that is
                if (16u * (uint32_t)q + 16u > lo && 16u * (uint32_t)q < hi) commit(qb + (uint32_t)q);
            }
        }
        if (on && ((l & 3u) == 3u || l + 1u == nlines)) {
            if (bitsA) fl0[l >> 2] = bitsA;  // the array is zeroed before the launch: only words with a flag are written
            if (bitsB) fl1[l >> 2] = bitsB;
            nflag += (uint32_t)__popc(bitsA) + (uint32_t)__popc(bitsB);
            bitsA = 0u; bitsB = 0u;
        }
    }
    {   // flagged pieces of the batch (both strands): the host compares them with the break-even of the windowed verification
        uint32_t t = nflag;
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) t += (uint32_t)__shfl_xor((int)t, d, 64);
        if (lane == 0u && t) atomicAdd(n_flagged, (unsigned long long)t);
    }
    // Matches of the rc strand that hang over ITS start (the read's last bytes) with o < R rows: rows o..R-1 of the window
    // end at the read's end, i.e. the rc block's first R-o rows do in its last column: D[R-o][n] + floor(alpha * o) <= k is
    // necessary.  One bit in the word after the rc strand's piece words tells k_flank_verify to scan the rc strand's
    // first columns (groups with BB_FILT_RC_BEGIN_HINT; windows that start deeper never hang, see upload_tables).
    if (live && n && (G->filt_mode & BB_FILT_RC_BEGIN_HINT)) {
        const uint32_t pb = WIDE ? pvB & maskR : (pv >> (SA + 16u)) & maskR, mb = WIDE ? mvB & maskR : (mv >> (SA + 16u)) & maskR;
        uint32_t hint = 0u;
        for (int o = 1; o < R; ++o) {
            const uint32_t low = (1u << (R - o)) - 1u;
            if ((int32_t)__popc(pb & low) - (int32_t)__popc(mb & low) + ovh[o] <= G->flank_k) hint = 1u;
        }
        if (hint) fl1[(nlines + 3u) >> 2] = hint;
    }
}

// Items = (read, strand) pairs, handed to lanes from a queue (one counter per strand): a read's verification work ranges
// from nothing to several intervals plus both ends, and a wave that gave every lane one fixed read waited for its busiest
// lane (a third of the lane-iterations did work).  A lane takes the next item as soon as its own is finished; finishing
// (overhang positions, count, flush of the buffered hits) is wave-wide code, run whenever some lane has an item to close.
template <int W, int STRAND>
__device__ __forceinline__ void flank_verify_lane(const uint8_t* __restrict__ bases, const uint64_t* __restrict__ offsets, uint32_t n_reads,
                                                  const uint8_t* __restrict__ tables, const bb_group_dev* __restrict__ G, uint32_t g,
                                                  uint32_t n_groups, const uint32_t* __restrict__ flags, uint32_t* __restrict__ cnt,
                                                  bb_hit_raw* __restrict__ hits, uint32_t hit_cap, uint32_t* __restrict__ hit_count,
                                                  uint32_t* __restrict__ queue, const uint32_t* s_peq, bb_hit_raw* stage,
                                                  uint32_t* s_flw /* this wave's [BB_VERIFY_FLW][64] */) {
    constexpr int S = (W <= 2 ? 2 : (W <= 4 ? 4 : 8));
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t stage_fill = 0u;               // wave-uniform
    uint32_t pool_next = 0u, pool_end = 0u;  // wave-uniform: items [pool_next, pool_end) of the queue belong to this wave
    const uint64_t off0 = offsets[0];
    const int32_t kk = G->flank_k, score0 = G->score0;
    const int m = G->m, R = G->filt_rows, U = G->filt_off;
    const int TB = (m - 1) & 31;
    const uint32_t* pv0 = reinterpret_cast<const uint32_t*>(tables + G->off_pv0);
    const int32_t* ovh = reinterpret_cast<const int32_t*>(tables + G->off_ovh);
    const uint32_t fmode = (uint32_t)G->filt_mode;
    const int ovh_steps = G->ovh_steps, pol_lm = G->pol_lm;
    const bool lm_left = pol_lm == BB_LM_PLATEAU_LEFT, lm_strict = pol_lm == BB_LM_STRICT;

    // ---- the item in hand
    enum : uint32_t { FREE = 0u, WORK = 1u, FIN = 2u, EXHAUSTED = 3u };
    uint32_t state = FREE;
    uint32_t read = 0u, n = 0u;
    const uint8_t* rb = bases;
    const uint32_t* fl = flags;
    bool fl_cached = false;  // the item's flag words (and the hint word) sit in the lane's LDS column
    uint32_t misf = 0u;
    int32_t nwords = 0, wi = 0;
    uint32_t bits = 0u;
    int phase = 3;
    bool need_end = false;
    uint32_t cur = 0u, stop = 0u;  // the run in progress covers positions [.., stop); cur = idx
    uint32_t pv[W], mv[W];
#pragma unroll
    for (int w = 0; w < W; ++w) { pv[w] = 0u; mv[w] = 0u; }
    int32_t sc = score0;
    lm_lane st = {score0, 1u, 0u, 0u};
    hit_buf hb = {0u, 0u, 0u, 0u, 0u};
    uint32_t idx = 0;  // columns consumed = scan position of the next byte
    auto step = [&](uint32_t ch) {
        uint32_t eq[W], d0[W], ph[W], mh[W];
        load_eq<W, S>(s_peq, ch, eq);
        myers_step<W>(pv, mv, eq, d0, ph, mh);
        sc += (int32_t)((ph[W - 1] >> TB) & 1u) - (int32_t)((mh[W - 1] >> TB) & 1u);
        ++idx;
        BB_LM_STEP_BUF(st, sc, idx);
    };
    // 16 scan positions p0.. as 4 words in scan order (the rc strand reads the text backwards)
    auto load16 = [&](uint32_t p0, uint32_t (&wq)[4]) {
        const int64_t a = STRAND ? (int64_t)n - 16 - (int64_t)p0 : (int64_t)p0;
        if (a >= 0 && a + 16 <= (int64_t)n) {
            u32x4_t v;
            __builtin_memcpy(&v, rb + a, 16);
            if (STRAND) { wq[0] = __builtin_bswap32(v[3]); wq[1] = __builtin_bswap32(v[2]); wq[2] = __builtin_bswap32(v[1]); wq[3] = __builtin_bswap32(v[0]); }
            else { wq[0] = v[0]; wq[1] = v[1]; wq[2] = v[2]; wq[3] = v[3]; }
        } else {
            wq[0] = wq[1] = wq[2] = wq[3] = 0u;
            for (int b = 0; b < 16; ++b) {
                const uint32_t p = p0 + (uint32_t)b;
                if (p < n) wq[b >> 2] |= (uint32_t)rb[STRAND ? (n - 1u - p) : p] << (8 * (b & 3));
            }
        }
    };
    // ---- interval source: [1] columns 1..m+k+1 (left overhang; only where the flags cannot vouch for the strand's start),
    // [2] the flagged pieces in scan order, [3] the last columns (the overhang positions continue from column n; only
    // where a flag lies close to the strand's end).  Columns [a, b], 1-based, unclamped.
    auto next_interval = [&](int64_t& a, int64_t& b) -> bool {
        if (phase == 0) { phase = 1; a = 1; b = (int64_t)m + kk + 1; return true; }
        if (phase == 1) {
            for (;;) {
                if (bits == 0u) {
                    if (STRAND ? wi <= 0 : wi + 1 >= nwords) break;
                    wi += STRAND ? -1 : 1;
                    bits = fl_cached ? s_flw[(uint32_t)wi * 64u + lane] : fl[wi];
                    continue;
                }
                const int bi = STRAND ? 31 - __clz((int)bits) : __ffs((int)bits) - 1;
                bits &= ~(1u << bi);
                const int64_t q0 = (int64_t)(((uint32_t)wi * 32u + (uint32_t)bi) * 16u) - (int64_t)misf;  // first forward position of the piece
                const int64_t f0 = q0 < 0 ? 0 : q0, f1 = q0 + 16 > (int64_t)n ? (int64_t)n : q0 + 16;  // forward positions [f0, f1): columns f0+1..f1
                if (f1 <= f0) continue;
                // a flag close to the strand's last columns: the match may run past them (right overhang)
                if (STRAND == 0 ? f1 + (m - U - R) + kk + 2 > (int64_t)n : f0 < (int64_t)(m - U) + kk + 2) need_end = true;
                if (STRAND == 0) { a = f0 + 1 + (m - U - R) - kk - 1; b = f1 + (m - U - R) + kk + 1; }
                else { a = (int64_t)n - f1 + (m - U) - kk - 1; b = (int64_t)n - f0 - 1 + (m - U) + kk + 1; }
                return true;
            }
            phase = 2;
        }
        if (phase == 2) { phase = 3; if (need_end) { a = (int64_t)n - 1; b = (int64_t)n; return true; } }
        return false;
    };
    for (;;) {
        // ---- free lanes take the next items of this strand's queue: the wave draws 64 at a time (one atomic), lanes
        // help themselves from that pool; a lane the pool cannot serve this round tries again in the next
        {
            const bool want = state == FREE;
            const unsigned long long wm = __ballot(want);
            if (wm) {
                if (pool_next == pool_end) {
                    uint32_t base = 0u;
                    if (lane == 0u) base = atomicAdd(queue, 64u);
                    pool_next = (uint32_t)__builtin_amdgcn_readfirstlane((int)__shfl((int)base, 0, 64));
                    pool_end = pool_next + 64u;
                }
                const uint32_t rank = (uint32_t)__popcll(wm & ((1ull << lane) - 1ull)), avail = pool_end - pool_next;
                const uint32_t took = min((uint32_t)__popcll(wm), avail);
                if (want && rank < avail) {
                    read = pool_next + rank;
                    if (read < n_reads) {
                        const uint64_t off = offsets[read];
                        n = (uint32_t)(offsets[read + 1] - off);
                        rb = bases + off;
                        misf = (uint32_t)((uint64_t)(uintptr_t)rb & 127u);
                        const uint32_t nlines = n ? (misf + n + 127u) >> 7 : 0u;
                        nwords = (int32_t)((nlines + 3u) >> 2);
                        fl = flags + filt_word_base(off, off0, read);
                        // all of the item's words now, back to back (they share one or two sectors; fetched one by one as the walk
                        // reaches them, each cost a sector again: the lines do not survive in L2 between a lane's iterations)
                        fl_cached = nwords < (int32_t)BB_VERIFY_FLW;
                        if (fl_cached)
                            for (int32_t w = 0; w <= nwords; ++w) s_flw[(uint32_t)w * 64u + lane] = fl[w];
                        wi = STRAND ? nwords : -1;
                        bits = 0u;
                        need_end = (fmode & BB_FILT_END_ALWAYS) != 0;
                        const bool need_begin = (fmode & (STRAND ? BB_FILT_RC_BEGIN_ALWAYS : BB_FILT_FWD_BEGIN_ALWAYS)) != 0 ||
                                                (STRAND == 1 && (fmode & BB_FILT_RC_BEGIN_HINT) && n &&
                                                 (fl_cached ? s_flw[(uint32_t)nwords * 64u + lane] : fl[nwords]) != 0u);
                        phase = n ? (need_begin ? 0 : 1) : 3;
                        cur = 0u; stop = 0u; idx = 0u;
#pragma unroll
                        for (int w = 0; w < W; ++w) { pv[w] = pv0[w]; mv[w] = 0u; }
                        sc = score0;
                        st.prev = score0; st.dec = 1u; st.nrep = 0u; st.cand = 0u;
                        hb.e0 = hb.e1 = hb.e2 = hb.e3 = hb.costs = 0u;
                        state = n ? WORK : FIN;
                    } else state = EXHAUSTED;
                }
                pool_next += took;
            }
        }
        if (!__any(state != EXHAUSTED)) break;
        // ---- one chunk of up to 16 columns per working lane
        if (state == WORK && cur >= stop) {
            // take intervals until one needs columns beyond the run in hand
            for (;;) {
                int64_t a, b;
                if (!next_interval(a, b)) { state = FIN; break; }
                if (a < 1) a = 1;
                if (b > (int64_t)n) b = (int64_t)n;
                if (b < a || (uint32_t)b <= stop) continue;  // empty, or inside what has been scanned
                const int64_t s0 = a - 1 - (m + kk) < 0 ? 0 : a - 1 - (m + kk);
                if ((uint32_t)s0 > cur) {  // a gap: restart from the all-insertions column m + k columns ahead of the interval
#pragma unroll
                    for (int x = 0; x < W; ++x) { const int bt = m - 32 * x; pv[x] = bt >= 32 ? 0xFFFFFFFFu : (bt > 0 ? ((1u << bt) - 1u) : 0u); mv[x] = 0u; }
                    sc = m; st.prev = m;
                    cur = (uint32_t)s0; idx = cur;
                }
                stop = (uint32_t)b;
                break;
            }
        }
        {
            // BB_VERIFY_CHUNKS x 16 columns per round, their text in 16-byte loads issued together: a verified interval (~100 columns
            // around a flagged piece) comes in one or two rounds, so the lines it lies in are requested once (taken 16 or 32 bytes a
            // round the same sectors were fetched again: they do not survive in L2 between a lane's rounds)
            const bool work = state == WORK && cur < stop;
            uint32_t wt[BB_VERIFY_CHUNKS][4];
            const uint32_t cntb = work ? min(16u * BB_VERIFY_CHUNKS, stop - cur) : 0u;
#pragma unroll
            for (int q = 0; q < BB_VERIFY_CHUNKS; ++q) {
                wt[q][0] = wt[q][1] = wt[q][2] = wt[q][3] = 0u;
                if (cntb > (uint32_t)(16 * q)) load16(cur + 16u * (uint32_t)q, wt[q]);
            }
#pragma unroll
            for (int hb2 = 0; hb2 < BB_VERIFY_CHUNKS; ++hb2) {
                if (__any(cntb > (uint32_t)(16 * hb2))) {
#pragma unroll
                    for (int b = 0; b < 16; ++b) {
                        const uint32_t w = wt[hb2][b >> 2];
                        if ((uint32_t)(16 * hb2 + b) < cntb) step((w >> (8 * (b & 3))) & 0xFFu);
                    }
                }
            }
            cur += cntb;
        }
        // ---- close finished items: the overhang positions continue from column n, but only if a run got there (otherwise
        // none of them can be <= k); count; flush of the buffered hits (wave-wide prefix sums: every lane takes part)
        if (__any(state == FIN)) {
            const bool fin = state == FIN;
            scan_finish<W, STRAND>(fin, n, m, kk, sc, pv, mv, idx, st, hb, ovh, read, g, n_groups, cnt, hits, hit_cap, hit_count, pol_lm, idx == n, ovh_steps,
                                   stage, &stage_fill);
            if (fin) state = FREE;
        }
    }
    stage_flush(stage, stage_fill, hits, hit_cap, hit_count);
}

template <int W>
__global__ __launch_bounds__(256) void k_flank_verify(const uint8_t* __restrict__ bases, const uint64_t* __restrict__ offsets, uint32_t n_reads,
                                                      const uint8_t* __restrict__ tables, const bb_group_dev* __restrict__ groups, uint32_t g,
                                                      uint32_t n_groups, const uint32_t* __restrict__ flags, uint64_t words_per_strand,
                                                      uint32_t* __restrict__ cnt, bb_hit_raw* __restrict__ hits, uint32_t hit_cap,
                                                      uint32_t* __restrict__ hit_count, uint32_t* __restrict__ queues) {
    constexpr int S = (W <= 2 ? 2 : (W <= 4 ? 4 : 8));
    __shared__ __attribute__((aligned(16))) uint32_t s_peq[256 * S];
    __shared__ __attribute__((aligned(16))) bb_hit_raw s_stage[4][BB_VERIFY_STAGE];
    __shared__ uint32_t s_flws[4][BB_VERIFY_FLW * 64];
    const bb_group_dev* G = groups + g;
    const uint32_t strand = blockIdx.y;
    {
        const uint32_t* src = reinterpret_cast<const uint32_t*>(tables + G->off_peq_flank[strand]);
        for (int i = threadIdx.x; i < 256 * S; i += 256) s_peq[i] = src[i];
    }
    __syncthreads();
    bb_hit_raw* stage = s_stage[threadIdx.x >> 6];
    if (strand == 0)
        flank_verify_lane<W, 0>(bases, offsets, n_reads, tables, G, g, n_groups, flags, cnt, hits, hit_cap, hit_count, queues, s_peq, stage, s_flws[threadIdx.x >> 6]);
    else
        flank_verify_lane<W, 1>(bases, offsets, n_reads, tables, G, g, n_groups, flags + words_per_strand, cnt, hits, hit_cap, hit_count, queues + 1, s_peq, stage, s_flws[threadIdx.x >> 6]);
}

// ------------------------------------------------------------------------------------------------
// exclusive scan of uint32 (3 small kernels): 2048 elements per block
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_scan_block(const uint32_t* __restrict__ in, uint32_t* __restrict__ out, uint64_t n,
                                                    uint32_t* __restrict__ sums) {
    __shared__ uint32_t s_w[4];
    const uint64_t base = (uint64_t)blockIdx.x * 2048u + (uint64_t)threadIdx.x * 8u;
    uint32_t v[8], t = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) { v[i] = base + i < n ? in[base + i] : 0u; }
#pragma unroll
    for (int i = 0; i < 8; ++i) { uint32_t x = v[i]; v[i] = t; t += x; }
    // wave inclusive scan of t
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    uint32_t inc = t;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { uint32_t y = __shfl_up(inc, d, 64); if (lane >= d) inc += y; }
    if (lane == 63) s_w[wv] = inc;
    __syncthreads();
    uint32_t wbase = 0;
    for (int i = 0; i < wv; ++i) wbase += s_w[i];
    const uint32_t excl = wbase + inc - t;
#pragma unroll
    for (int i = 0; i < 8; ++i) if (base + i < n) out[base + i] = v[i] + excl;
    if (threadIdx.x == 255) sums[blockIdx.x] = wbase + inc;
}
__global__ __launch_bounds__(64) void k_scan_sums(uint32_t* __restrict__ sums, uint32_t nb) {
    // single wave, sequential chunks of 64 with carry
    uint32_t carry = 0;
    const int lane = threadIdx.x;
    for (uint32_t b = 0; b < nb; b += 64) {
        uint32_t x = b + lane < nb ? sums[b + lane] : 0u;
        uint32_t inc = x;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { uint32_t y = __shfl_up(inc, d, 64); if (lane >= d) inc += y; }
        if (b + lane < nb) sums[b + lane] = carry + inc - x;
        carry += __shfl(inc, 63, 64);
    }
}
__global__ __launch_bounds__(256) void k_scan_add(uint32_t* __restrict__ out, uint64_t n, const uint32_t* __restrict__ sums) {
    const uint64_t base = (uint64_t)blockIdx.x * 2048u + (uint64_t)threadIdx.x * 8u;
    const uint32_t a = sums[blockIdx.x];
#pragma unroll
    for (int i = 0; i < 8; ++i) if (base + i < n) out[base + i] += a;
}

// ------------------------------------------------------------------------------------------------
// k_flank_trace: one lane per raw flank hit.  Recomputes the DP on the last m+k columns before the
// hit's end with move bits kept per column (private memory), walks back, and produces the ordered
// bb_hit (flank coordinates + barcode window).
// ------------------------------------------------------------------------------------------------
// MOVES_IN_LDS: the two move bit-vectors of every column are kept in LDS ([column][word][lane], so the
// 64 lanes of the block hit 64 different banks) instead of private memory — private arrays of this
// size live in HBM-backed scratch and made this small kernel the largest HBM consumer of the pipeline
// (profiles/r01_v3_pmc.txt: 10.9 GB fetched per 2 M reads).  The host picks the LDS variant whenever
// (m + k + 1) * W * 512 bytes fit in 64 KB.
// MODE 0: move bits in private memory (any geometry); 1: in LDS, every row of every column; 2: in LDS, only the
// band of 16 rows around the end cell's diagonal (one word per column and lane: both planes).  A path of cost
// <= k leaves that diagonal by at most k rows, so for k <= 6 the band holds every cell the walk can visit, and
// a quarter of the LDS lets four times as many blocks share a CU.
// MODE 4: the band for k <= 3 — 2 (k + 1) <= 8 rows, both planes of a column in 16 bits: half the LDS again.  The kernel waits on memory
// two thirds of its time (one lane per hit: the raw hit, the read's offset, four text chunks, the Peq rows), and its LDS decides how many
// waves share a CU in the meantime (13 KB per 64-lane block: 12; 6.5 KB: 24).
// MODE 3 (k > 6 where the full height does not fit): no move bits during the forward pass, only the column state (Pv, Mv)
// every 8 columns in LDS; the walk then goes back block by block — the wave recomputes the 8 columns of a block from its
// checkpoint with their move bits into a small LDS window and every lane walks through its part of the block — so the DP
// is computed twice, and nothing lives in private memory (the k = 20 tracebacks of the rapid kits: 3.9 -> ms below).
#define BB_TRACE_CKB 8
#define BB_TRACE_REC_STRIDE 25  // words per staged bb_hit (24) + 1: lanes land in different banks
template <int W, int MODE>
__device__ __forceinline__ uint32_t flank_trace_lane(const uint8_t* __restrict__ bases, const uint64_t* __restrict__ offsets,
                                                     const uint8_t* __restrict__ tables, const bb_group_dev* __restrict__ groups,
                                                     uint32_t n_groups, const bb_hit_raw* __restrict__ raw, uint32_t n_hits,
                                                     const uint32_t* __restrict__ slot_base, uint32_t gmask, int mk_max, uint32_t* __restrict__ orec,
                                                     uint32_t* s_moves) {
    constexpr int S = (W <= 2 ? 2 : (W <= 4 ? 4 : 8));
    constexpr int MAXC = 32 * W + BB_MAX_FLANK_K + 1;  // columns of the private-memory variant: m + k
    const uint32_t t = blockIdx.x * 64u + threadIdx.x;
    if (t >= n_hits) return 0xFFFFFFFFu;
    const bb_hit_raw h = raw[t];
    if (!((gmask >> h.group) & 1u)) return 0xFFFFFFFFu;  // the launch's groups: same W, same mode (launch_trace)
    const bb_group_dev& G = groups[h.group];  // not a copy: indexing a private copy by the strand put the struct into scratch memory
    const uint64_t off = offsets[h.read_idx];
    const int32_t n = (int32_t)(offsets[h.read_idx + 1] - off);
    const uint8_t* rb = bases + off;
    const int m = G.m, k = G.flank_k, bar_lo = G.bar_lo, bar_hi = G.bar_hi;
    const uint32_t prio = (uint32_t)__builtin_amdgcn_readfirstlane(groups[0].pol_prio);  // the context's policy: the same in every group
    const uint32_t* peq = reinterpret_cast<const uint32_t*>(tables + G.off_peq_flank[h.strand]);
    const uint32_t* pv0 = reinterpret_cast<const uint32_t*>(tables + G.off_pv0);
    const int32_t* ovh = reinterpret_cast<const int32_t*>(tables + G.off_ovh);

    const int32_t e = (int32_t)h.e;
    const int32_t o = e > n ? e - n : 0;
    const int32_t j0 = m - o, i0 = e > n ? n : e;
    int32_t s0 = i0 - (m + k);
    if (s0 < 0) s0 = 0;
    const int32_t w = i0 - s0;  // <= m + k < MAXC

    // s_moves: MODE 1: [column][lo|hi][word][64 lanes]; MODE 2: [column][64 lanes]
    uint32_t plo_[MODE == 0 ? MAXC : 1][W], phi_[MODE == 0 ? MAXC : 1][W];
    // first row (0-based bit) of column c's band: the diagonal through the end cell (j0, w), k + 1 rows above it
    auto band_lo = [&](int c) -> int { const int b = (j0 - 1) - (w - c) - (k + 1); return b < 0 ? 0 : b; };
    auto bits16 = [&](const uint32_t (&v)[W], int sh) -> uint32_t {  // bits [sh, sh + 16) of the W-word vector
        const int q = sh >> 5, r = sh & 31;
        uint32_t a = v[0], b = W > 1 ? v[1] : 0u;
#pragma unroll
        for (int x = 1; x < W; ++x) { a = q == x ? v[x] : a; b = q == x ? (x + 1 < W ? v[x + 1] : 0u) : b; }
        return (uint32_t)((((unsigned long long)b << 32) | a) >> r) & 0xFFFFu;
    };
    auto put = [&](int c, int x, uint32_t l, uint32_t hh) {
        if constexpr (MODE == 1) { s_moves[((c * 2 + 0) * W + x) * 64 + threadIdx.x] = l; s_moves[((c * 2 + 1) * W + x) * 64 + threadIdx.x] = hh; }
        else if constexpr (MODE == 0) { plo_[c][x] = l; phi_[c][x] = hh; }
    };
    auto put_band = [&](int c, const uint32_t (&l)[W], const uint32_t (&hh)[W]) {
        const int sh = band_lo(c);
        if constexpr (MODE == 4) reinterpret_cast<uint16_t*>(s_moves)[c * 64 + threadIdx.x] = (uint16_t)((bits16(l, sh) & 0xFFu) | ((bits16(hh, sh) & 0xFFu) << 8));
        else s_moves[c * 64 + threadIdx.x] = bits16(l, sh) | (bits16(hh, sh) << 16);
    };
    // 2-bit move of cell (row bit `bit`, column c)
    auto get_op = [&](int c, int bit) -> uint32_t {
        if constexpr (MODE == 2) {
            const uint32_t wv = s_moves[c * 64 + threadIdx.x];
            const int rel = bit - band_lo(c);
            return ((wv >> rel) & 1u) | (((wv >> (16 + rel)) & 1u) << 1);
        } else if constexpr (MODE == 4) {
            const uint32_t wv = reinterpret_cast<const uint16_t*>(s_moves)[c * 64 + threadIdx.x];
            const int rel = bit - band_lo(c);
            return ((wv >> rel) & 1u) | (((wv >> (8 + rel)) & 1u) << 1);
        } else if constexpr (MODE == 1) {
            const uint32_t lw = s_moves[((c * 2 + 0) * W + (bit >> 5)) * 64 + threadIdx.x], hw = s_moves[((c * 2 + 1) * W + (bit >> 5)) * 64 + threadIdx.x];
            return ((lw >> (bit & 31)) & 1u) | (((hw >> (bit & 31)) & 1u) << 1);
        } else {
            return ((plo_[c][bit >> 5] >> (bit & 31)) & 1u) | (((phi_[c][bit >> 5] >> (bit & 31)) & 1u) << 1);
        }
    };
    uint32_t pv[W], mv[W];
#pragma unroll
    for (int x = 0; x < W; ++x) {
        if (s0 == 0) pv[x] = pv0[x];
        else { int bits = m - 32 * x; pv[x] = bits >= 32 ? 0xFFFFFFFFu : (bits > 0 ? ((1u << bits) - 1u) : 0u); }
        mv[x] = 0;
    }
    // The window's text is fetched 16 scan positions at a time (one unaligned 16-byte load, the next chunk
    // requested before the current one is consumed): a byte load per column left the DP waiting on ~50
    // dependent HBM round trips per hit, which was most of this kernel's time.
    auto load16 = [&](int32_t p0, uint32_t (&wq)[4]) {  // scan positions p0 .. p0+15 -> bytes 0..15 of wq (scan order)
        const int32_t a = h.strand ? (n - 16 - p0) : p0;  // forward byte offset of the chunk's lowest address
        if (a >= 0 && a + 16 <= n) {
            u32x4_t v;
            __builtin_memcpy(&v, rb + a, 16);
            if (h.strand) { wq[0] = __builtin_bswap32(v[3]); wq[1] = __builtin_bswap32(v[2]); wq[2] = __builtin_bswap32(v[1]); wq[3] = __builtin_bswap32(v[0]); }
            else { wq[0] = v[0]; wq[1] = v[1]; wq[2] = v[2]; wq[3] = v[3]; }
        } else {  // chunk sticks out of the read: byte loads, positions outside the read read as 0 (never used)
            wq[0] = wq[1] = wq[2] = wq[3] = 0u;
            for (int b = 0; b < 16; ++b) {
                const int32_t p = p0 + b;
                if (p >= 0 && p < n) wq[b >> 2] |= (uint32_t)rb[h.strand ? (n - 1 - p) : p] << (8 * (b & 3));
            }
        }
    };
    auto ck_store = [&](int blk) {
#pragma unroll
        for (int x = 0; x < W; ++x) { s_moves[((blk * 2 * W) + x) * 64 + threadIdx.x] = pv[x]; s_moves[((blk * 2 * W) + W + x) * 64 + threadIdx.x] = mv[x]; }
    };
    if constexpr (MODE == 3) ck_store(0);
    auto column = [&](int32_t c, uint32_t ch) {
        uint32_t eq[W], d0[W], ph[W], mh[W], l[W], hh[W];
        load_eq<W, S>(peq, ch, eq);
        myers_step<W>(pv, mv, eq, d0, ph, mh);
        move_bits_prio<W>(prio, eq, d0, ph, pv, l, hh);
        if constexpr (MODE == 2 || MODE == 4) put_band(c, l, hh);
        else if constexpr (MODE == 3) { if ((c & (BB_TRACE_CKB - 1)) == 0) ck_store(c / BB_TRACE_CKB); }
        else {
#pragma unroll
            for (int x = 0; x < W; ++x) put(c, x, l[x], hh[x]);
        }
    };
    if constexpr ((MODE == 2 || MODE == 4) && W <= 2) {
        // band variants (k <= 6: at most 32 W + 6 columns): every chunk of the window's text requested before the first column. Its
        // 50-70 bytes lie in one or two lines; fetched a chunk at a time as the DP got there, a line was often gone from L2 again by
        // the next request once 24 waves per CU were in flight (1.2 -> 1.7 GB of HBM reads per step with the 8-row band).
        constexpr int NCH = 4;  // 64 columns at once; the rest (m + k > 64: two-word flanks of more than 58 characters) one by one
        uint32_t buf[NCH][4];
#pragma unroll
        for (int q = 0; q < NCH; ++q) {
            buf[q][0] = buf[q][1] = buf[q][2] = buf[q][3] = 0u;
            if (16 * q < w) load16(s0 + 16 * q, buf[q]);
        }
#pragma unroll
        for (int q = 0; q < NCH; ++q) {
            if (__any(16 * q < w)) {
#pragma unroll
                for (int b = 0; b < 16; ++b) {
                    const int32_t c = 16 * q + b + 1;
                    if (c <= w) column(c, (buf[q][b >> 2] >> (8 * (b & 3))) & 0xFFu);
                }
            }
        }
        for (int32_t cb = 16 * NCH; cb < w; cb += 16) {
            uint32_t cur[4];
            load16(s0 + cb, cur);
#pragma unroll
            for (int b = 0; b < 16; ++b) {
                const int32_t c = cb + b + 1;
                if (c <= w) column(c, (cur[b >> 2] >> (8 * (b & 3))) & 0xFFu);
            }
        }
    } else {
        uint32_t cur[4], nxt[4];
        load16(s0, cur);
        for (int32_t cb = 0; cb < w; cb += 16) {
            if (cb + 16 < w) load16(s0 + cb + 16, nxt);
#pragma unroll
            for (int b = 0; b < 16; ++b) {
                const int32_t c = cb + b + 1;
                if (c <= w) column(c, (cur[b >> 2] >> (8 * (b & 3))) & 0xFFu);
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) cur[q] = nxt[q];
        }
    }
    (void)ovh;
    // traceback from (j0, w)
    int32_t j = j0, i = w, cnt = 0, first_txt = 0, last_txt = 0;
    auto take = [&](uint32_t op) {
        if (op != 2u) --j;
        if (op != 3u) --i;
        if (j >= bar_lo && j <= bar_hi) {  // path cell Pos(j, s0+i) of this op
            const int32_t sp = s0 + i;
            int32_t f = h.strand ? (n - 1 - sp) : sp;
            if (f < 0) f = 0;
            if (cnt == 0) last_txt = f;
            first_txt = f;
            ++cnt;
        }
    };
    if constexpr (MODE == 3) {
        const int NB = (m + k) / BB_TRACE_CKB + 1;          // checkpoints 0 .. NB-1 of this lane's group (the LDS is sized for the launch's largest)
        uint32_t* win = s_moves + NB * 2 * W * 64;           // [column of the block][lo | hi][word][64 lanes]
        uint32_t tq[4], tn[4] = {0u, 0u, 0u, 0u};            // the block's text; the next (lower) block's, requested a block ahead
        {
            const int32_t cl = ((w - 1) / BB_TRACE_CKB) * BB_TRACE_CKB;  // this lane's last block
            if (w > 0) load16(s0 + cl, tn);
        }
        for (int blk = (mk_max - 1) / BB_TRACE_CKB; blk >= 0; --blk) {  // wave-uniform: the largest m + k of the launch's groups
            const int32_t c0 = blk * BB_TRACE_CKB;           // the block holds columns c0+1 .. c0+8
            if (c0 < w) {
#pragma unroll
                for (int q = 0; q < 4; ++q) tq[q] = tn[q];
                if (c0 >= BB_TRACE_CKB) load16(s0 + c0 - BB_TRACE_CKB, tn);
            }
            if (c0 < w && j > 0 && i > c0) {
#pragma unroll
                for (int x = 0; x < W; ++x) { pv[x] = s_moves[((blk * 2 * W) + x) * 64 + threadIdx.x]; mv[x] = s_moves[((blk * 2 * W) + W + x) * 64 + threadIdx.x]; }
#pragma unroll
                for (int b = 0; b < BB_TRACE_CKB; ++b) {
                    if (c0 + b + 1 <= w) {
                        const uint32_t ch = (tq[b >> 2] >> (8 * (b & 3))) & 0xFFu;
                        uint32_t eq[W], d0[W], ph[W], mh[W], l[W], hh[W];
                        load_eq<W, S>(peq, ch, eq);
                        myers_step<W>(pv, mv, eq, d0, ph, mh);
                        move_bits_prio<W>(prio, eq, d0, ph, pv, l, hh);
#pragma unroll
                        for (int x = 0; x < W; ++x) { win[((b * 2 + 0) * W + x) * 64 + threadIdx.x] = l[x]; win[((b * 2 + 1) * W + x) * 64 + threadIdx.x] = hh[x]; }
                    }
                }
                while (j > 0 && i > c0) {
                    const int cc = i - c0 - 1, bit = j - 1;
                    const uint32_t lw = win[((cc * 2 + 0) * W + (bit >> 5)) * 64 + threadIdx.x], hw = win[((cc * 2 + 1) * W + (bit >> 5)) * 64 + threadIdx.x];
                    take(((lw >> (bit & 31)) & 1u) | (((hw >> (bit & 31)) & 1u) << 1));
                }
            }
        }
        while (j > 0 && s0 != 0) take(3u);  // column 0 reached inside the read: the rows left are deleted (s0 == 0: left overhang, they lie outside)
    } else {
    while (j > 0) {
        uint32_t op;
        if (i == 0) {
            if (s0 == 0) break;  // left overhang: remaining pattern is outside the read
            op = 3u;
        } else {
            op = get_op(i, j - 1);
        }
        take(op);
    }
    }
    const int32_t ts = s0 + i, te = i0;
    bb_hit out;
    out.read_idx = h.read_idx;
    out.text_start = (uint32_t)(h.strand ? n - te : ts);
    out.text_end = (uint32_t)(h.strand ? n - ts : te);
    out.cost = h.cost; out.group = h.group; out.strand = h.strand;
    out.valid = cnt >= 2;
    int32_t rlo = first_txt < last_txt ? first_txt : last_txt, rhi = first_txt < last_txt ? last_txt : first_txt;
    int32_t ws = rlo >= BB_PADDING ? rlo - BB_PADDING : 0;
    int32_t we = rhi + BB_PADDING < n ? rhi + BB_PADDING : n;
    if (we < ws) we = ws;
    out.ws = (uint32_t)ws; out.we = (uint32_t)we;
    out._pad[0] = out._pad[1] = out._pad[2] = 0;
    out.read_len = (uint32_t)n;
    // order of a read's matches: group, forward matches, rc matches — the rc ones as the rc scan found them or, policy
    // [H2], in ascending forward position (the scan runs over the reversed text: the reverse of its order)
    const uint64_t sb = ((uint64_t)h.read_idx * n_groups + h.group) * 2 + h.strand;
    const uint32_t slot = (h.strand && groups[0].pol_rc_fwd) ? slot_base[sb + 1] - 1u - h.ordinal : slot_base[sb] + h.ordinal;
    {
        const uint32_t* src = reinterpret_cast<const uint32_t*>(&out);
#pragma unroll
        for (int q = 0; q < 8; ++q) orec[q] = src[q];
    }
    const int32_t wn = we - ws;
    const uint8_t* lut = tables + G.off_lut;
    if (wn <= 64) {  // window codes for k_barcode_reg: the window's bytes in four 16-byte loads, then the base-set LUT
        u32x4_t tv[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int32_t a = ws + 16 * q;
            if (16 * q < wn && a + 16 <= n) __builtin_memcpy(&tv[q], rb + a, 16);
            else {
                tv[q] = u32x4_t{0u, 0u, 0u, 0u};
                for (int b = 0; b < 16; ++b)
                    if (16 * q + b < wn) tv[q][b >> 2] |= (uint32_t)rb[a + b] << (8 * (b & 3));
            }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            uint32_t w4[4] = {0u, 0u, 0u, 0u};
#pragma unroll
            for (int b = 0; b < 16; ++b) {
                const int c = 16 * q + b;
                const uint32_t code = c < wn ? (uint32_t)lut[(tv[q][b >> 2] >> (8 * (b & 3))) & 0xFFu] : 0u;
                w4[b >> 2] |= code << (8 * (b & 3));
            }
            orec[8 + 4 * q] = w4[0]; orec[9 + 4 * q] = w4[1]; orec[10 + 4 * q] = w4[2]; orec[11 + 4 * q] = w4[3];
        }
    } else {
#pragma unroll
        for (int q = 8; q < 24; ++q) orec[q] = 0u;
    }
    return slot;
}
// The 96-byte records leave through LDS: six adjacent lanes write one record's six 16-byte pieces, so a record goes out
// as one contiguous burst (a lane writing its own record piece by piece cost ~315 bytes of HBM writes per record,
// profiles/r02_v23 traffic).
template <int W, int MODE>
__global__ __launch_bounds__(64) void k_flank_trace(const uint8_t* __restrict__ bases, const uint64_t* __restrict__ offsets,
                                                    const uint8_t* __restrict__ tables, const bb_group_dev* __restrict__ groups,
                                                    uint32_t n_groups, const bb_hit_raw* __restrict__ raw, uint32_t n_hits,
                                                    const uint32_t* __restrict__ slot_base, bb_hit* __restrict__ hits, uint32_t gmask, int mk_max) {
    extern __shared__ uint32_t s_dyn[];
    __shared__ uint32_t s_slot[64];
    static_assert(sizeof(bb_hit) == 96, "six 16-byte pieces");
    // the staged records reuse the move bits' LDS (>= 64 * BB_TRACE_REC_STRIDE words, launch_trace): the block is one wave,
    // a lane writes its record after every lane's walk is over, and LDS operations of a wave execute in order
    uint32_t* s_rec = s_dyn;
    s_slot[threadIdx.x] = flank_trace_lane<W, MODE>(bases, offsets, tables, groups, n_groups, raw, n_hits, slot_base, gmask, mk_max,
                                                    s_rec + threadIdx.x * BB_TRACE_REC_STRIDE, s_dyn);
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < 64u * 6u; i += 64u) {
        const uint32_t hl = i / 6u, pc = i - hl * 6u, slot = s_slot[hl];
        if (slot != 0xFFFFFFFFu) {
            const uint32_t* r = s_rec + hl * BB_TRACE_REC_STRIDE + 4u * pc;
            reinterpret_cast<uint4*>(hits + slot)[pc] = make_uint4(r[0], r[1], r[2], r[3]);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// k_barcode: block = HPB hits x LPH lanes (LPH = n_seqs).  One lane per (hit, barcode pattern).
// ------------------------------------------------------------------------------------------------
struct __attribute__((aligned(16))) bb_rowtmp {  // one per flank hit: the provisional row; row._pad[0] = 1 when the hit has a row
    bb_row row;
};
static_assert(sizeof(bb_rowtmp) == 48, "bb_rowtmp is three 16-byte pieces");
// What the fast barcode kernel leaves in a hit's row slot for k_rows: the traced path of the barcode with the highest
// score BOUND (column planes, consumed rows) and the second-highest bound.  `marker` sits where bb_row keeps the
// pipeline's row flag (_pad[0], byte 45): 0 = no row, 1 = row, 2 = this record.
struct __attribute__((aligned(16))) bb_winrec {
    unsigned long long plo, phi, diagrow;
    double ub_second;
    uint8_t tstart, best_pos;
    uint16_t top;
    uint8_t flags;
    uint8_t _pad0[8];
    uint8_t marker;
    uint8_t _pad[2];
};
static_assert(sizeof(bb_winrec) == 48 && offsetof(bb_winrec, marker) == 45, "bb_winrec overlays bb_rowtmp");

__device__ __forceinline__ int32_t rel_dist_to_end(int64_t pos, int64_t read_len) {  // searcher.rs:183-199
    if (pos < 0) return 1;
    if (pos <= read_len / 2) return pos == 0 ? 1 : (int32_t)pos;
    if (pos == read_len) return -1;
    return (int32_t)-(read_len - pos);
}

template <int WB, bool PEQ_LDS>
__global__ __launch_bounds__(1024) void k_barcode(const uint8_t* __restrict__ bases, const uint64_t* __restrict__ offsets,
                                                  const uint8_t* __restrict__ tables, const bb_group_dev* __restrict__ groups,
                                                  uint32_t g, const bb_hit* __restrict__ hits, const uint32_t* __restrict__ hit_list,
                                                  const uint32_t* __restrict__ list_cnt, uint32_t n_hits_all, uint32_t hpb,
                                                  double min_score, double min_score_diff, bb_rowtmp* __restrict__ rows) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const bb_group_dev G = groups[g];
    const uint32_t n_list = hit_list ? list_cnt[g] : n_hits_all;
    if (blockIdx.x * hpb >= n_list) return;
    const int N = G.n_seqs, m = G.m_bar;
    // LDS carve: [peq: 2*16*N*WB words][win: hpb*BB_MAX_WIN bytes][score: hpb*N doubles][cnt/top: hpb*4 ints]
    uint32_t* s_peq = reinterpret_cast<uint32_t*>(smem);
    size_t o = PEQ_LDS ? (size_t)2 * 16 * N * WB * 4 : 0;
    double* s_score = reinterpret_cast<double*>(smem + o);
    o += (size_t)hpb * N * 8;
    int32_t* s_int = reinterpret_cast<int32_t*>(smem + o);  // [hpb][4]: cnt1, top, ncand, unused
    o += (size_t)hpb * 16;
    uint8_t* s_win = smem + o;

    const uint32_t* gpeq0 = reinterpret_cast<const uint32_t*>(tables + G.off_peq_bar[0]);
    if (PEQ_LDS) {
        const int words = 2 * 16 * N * WB;  // strand-1 table follows strand-0 contiguously
        for (int i = threadIdx.x; i < words; i += blockDim.x) s_peq[i] = gpeq0[i];
    }
    const int hl = threadIdx.x / N;       // local hit
    const int p = threadIdx.x - hl * N;   // pattern index
    const uint32_t li = blockIdx.x * hpb + hl;
    bool active = hl < (int)hpb && li < n_list;
    bb_hit H;
    uint32_t hit_idx = 0;
    int32_t wn = 0;
    if (active) {
        hit_idx = hit_list ? hit_list[li] : li;
        H = hits[hit_idx];
        if (!H.valid) { active = false; if (p == 0) rows[hit_idx].row._pad[0] = 0; }
    }
    if (active) {
        wn = (int32_t)(H.we - H.ws);
        const uint8_t* rb = bases + offsets[H.read_idx];
        for (int c = p; c < wn; c += N) s_win[hl * BB_MAX_WIN + c] = bb_text_code(rb[H.ws + c]);
        if (p == 0) { s_int[hl * 4 + 0] = 0; s_int[hl * 4 + 1] = -1; s_int[hl * 4 + 2] = 0; }
    }
    __syncthreads();

    // the context's policy (include/barbell_amd_policy.h): this kernel honours all of it
    const uint32_t prio = (uint32_t)G.pol_prio;
    const bool lm_left = G.pol_lm == BB_LM_PLATEAU_LEFT, lm_strict = G.pol_lm == BB_LM_STRICT, tie_last = G.pol_tie_last != 0;
    // ---- forward pass with move bits ----
    uint32_t lo[BB_MAX_WIN + 1][WB], hi[BB_MAX_WIN + 1][WB];
    int32_t best_cost = 0x7FFFFFFF, best_pos = -1;
    if (active) {
        const uint32_t* peq = PEQ_LDS ? s_peq + (size_t)H.strand * 16 * N * WB
                                      : gpeq0 + (size_t)H.strand * 16 * N * WB;
        uint32_t pv[WB], mv[WB];
#pragma unroll
        for (int x = 0; x < WB; ++x) { int bits = m - 32 * x; pv[x] = bits >= 32 ? 0xFFFFFFFFu : (bits > 0 ? ((1u << bits) - 1u) : 0u); mv[x] = 0; }
        const int TW = (m - 1) >> 5, TB = (m - 1) & 31;
        int32_t score = m, prev = m, lmc = 0;
        uint32_t dec = 1;
        for (int32_t c = 1; c <= wn; ++c) {
            const uint32_t code = s_win[hl * BB_MAX_WIN + c - 1];
            uint32_t eq[WB], d0[WB], ph[WB], mh[WB], l[WB], hh[WB];
            const uint32_t* e = peq + ((size_t)code * N + p) * WB;
#pragma unroll
            for (int x = 0; x < WB; ++x) eq[x] = e[x];
            myers_step<WB>(pv, mv, eq, d0, ph, mh);
            move_bits_prio<WB>(prio, eq, d0, ph, pv, l, hh);
#pragma unroll
            for (int x = 0; x < WB; ++x) { lo[c][x] = l[x]; hi[c][x] = hh[x]; }
            score += (int32_t)((ph[TW] >> TB) & 1u) - (int32_t)((mh[TW] >> TB) & 1u);
            // local minima (every position is <= k2 = m; policy [H1]): first strictly-lowest (searcher.rs:294-300; policy [H7])
            if (score > prev) {
                if (dec && (prev < best_cost || (tie_last && prev == best_cost))) { best_cost = prev; best_pos = lm_left ? lmc : c - 1; }
                dec = 0;
            } else if (score < prev) { dec = 1; lmc = c; }
            else if (lm_strict) dec = 0;
            prev = score;
        }
        if (dec && (prev < best_cost || (tie_last && prev == best_cost))) { best_cost = prev; best_pos = lm_left ? lmc : wn; }
        if (best_pos >= 0 && best_cost <= G.k1) atomicAdd(&s_int[hl * 4 + 0], 1);
        if (best_pos >= 0 && best_cost <= G.k2) atomicAdd(&s_int[hl * 4 + 2], 1);
    }
    __syncthreads();

    // ---- pass decision (searcher.rs:303-328), traceback, Lodhi, sub-path ----
    double s_norm = -1.0;
    int32_t pat_lo = 0, pat_hi = 0, txt_lo = 0, txt_hi = 0, bcost = 0;
    bool cand = false;
    if (active) {
        const int cnt1 = s_int[hl * 4 + 0];
        const bool pass2 = cnt1 <= 1 && G.k1 < G.k2;
        cand = best_pos >= 0 && (pass2 ? best_cost <= G.k2 : best_cost <= G.k1);
        if (cand) {
            uint8_t ops[BB_MAX_OPS];  // reversed
            int nops = 0;
            int32_t j = m, i = best_pos;
            while (j > 0) {
                uint32_t op;
                if (i == 0) op = 3u;
                else {
                    const int bit = j - 1;
                    uint32_t lw = lo[i][0], hw = hi[i][0];
#pragma unroll
                    for (int x = 1; x < WB; ++x) { lw = (bit >> 5) == x ? lo[i][x] : lw; hw = (bit >> 5) == x ? hi[i][x] : hw; }
                    op = ((lw >> (bit & 31)) & 1u) | (((hw >> (bit & 31)) & 1u) << 1);
                }
                ops[nops++] = (uint8_t)op;
                if (op != 2u) --j;
                if (op != 3u) --i;
            }
            // forward walk: Lodhi (policy [H8]: subsequence length p, lambda, decay exponent per op — the checker's sequence of
            // f64 operations, no contraction) + map_pat_to_text_with_cost (cigar_parse.rs:6-68)
            const int lp = G.pol_lodhi_p;
            double dk[4];
#pragma unroll
            for (int o = 0; o < 4; ++o) {
                double d = 1.0;
                const int ex = (G.pol_lodhi_exp >> (8 * o)) & 0xFF;
                for (int e = 0; e < ex; ++e) d = e == 0 ? G.pol_lambda : d * G.pol_lambda;
                dk[o] = d;
            }
            double a0 = 0.0, a1 = 0.0, a2 = 0.0, sc = 0.0;   // A[0], A[1], A[2] of the checker
            int32_t pj = 0, ti = i;
            bool any = false;
            for (int t = nops - 1; t >= 0; --t) {
                const uint32_t op = ops[t];
                const double d = op == 0u ? dk[0] : op == 1u ? dk[1] : op == 2u ? dk[2] : dk[3];
                if (op == 0u) {
                    sc = sc + d * (lp >= 4 ? a2 : lp == 3 ? a1 : lp == 2 ? a0 : 1.0);
                    if (lp >= 4) a2 = d * (a2 + a1);
                    if (lp >= 3) a1 = d * (a1 + a0);
                    if (lp >= 2) a0 = d * (a0 + 1.0);
                } else {
                    if (lp >= 4) a2 = d * a2;
                    if (lp >= 3) a1 = d * a1;
                    if (lp >= 2) a0 = d * a0;
                }
                if (pj >= G.rel_lo && pj < G.rel_hi) {
                    if (!any) { any = true; pat_lo = pj; txt_lo = ti; }
                    pat_hi = pj + 1; txt_hi = ti + 1; bcost += op != 0u;
                }
                if (op != 2u) ++pj;
                if (op != 3u) ++ti;
            }
            s_norm = G.perfect > 0.0 ? sc / G.perfect : 0.0;
        }
        s_score[hl * N + p] = s_norm;
    }
    __syncthreads();
    if (active && p == 0) {
        // stable sort descending by s_norm (searcher.rs:377): top = first maximum, second = best of the rest
        int top = -1, second = -1;
        double ts = 0.0, ss = 0.0;
        for (int q = 0; q < N; ++q) { double v = s_score[hl * N + q]; if (v >= 0.0 && (top < 0 || v > ts)) { top = q; ts = v; } }
        for (int q = 0; q < N; ++q) { double v = s_score[hl * N + q]; if (v >= 0.0 && q != top && (second < 0 || v > ss)) { second = q; ss = v; } }
        bool valid = top >= 0 && ts >= min_score;                         // searcher.rs:391-396
        if (valid && second >= 0) valid = (ts - ss) >= min_score_diff;
        s_int[hl * 4 + 1] = valid ? top : -1;
    }
    __syncthreads();
    if (active) {
        const int top = s_int[hl * 4 + 1];
        const uint32_t read_len = (uint32_t)(offsets[H.read_idx + 1] - offsets[H.read_idx]);
        if ((top >= 0 && p == top) || (top < 0 && p == 0)) {
            bb_rowtmp R;
            bb_row& r = R.row;
            r.read_idx = H.read_idx; r.read_len = read_len;
            r.rel_dist_to_end = rel_dist_to_end((int64_t)H.text_start, (int64_t)read_len);
            r.read_start_flank = H.text_start; r.read_end_flank = H.text_end;
            r.flank_cost = H.cost; r.group_idx = H.group; r.strand = H.strand;
            r._pad[0] = 1; r._pad[1] = r._pad[2] = 0;
            if (top >= 0) {                                                // searcher.rs:398-416
                r.read_start_bar = H.ws + (uint32_t)txt_lo; r.read_end_bar = H.ws + (uint32_t)txt_hi;
                r.bar_start = H.ws + (uint32_t)pat_lo; r.bar_end = H.ws + (uint32_t)pat_hi;
                r.match_type = (uint8_t)G.type; r.barcode_cost = (int16_t)bcost; r.barcode_idx = (int16_t)top;
            } else {                                                       // searcher.rs:241-265
                r.read_start_bar = H.text_start; r.read_end_bar = H.text_end;
                r.bar_start = 0; r.bar_end = 0;
                r.match_type = (uint8_t)(G.type == BB_FTAG ? BB_FFLANK : BB_RFLANK);
                r.barcode_cost = (int16_t)G.m_bar; r.barcode_idx = -1;
            }
            rows[hit_idx] = R;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Columns are processed in wave-uniform groups of BB_CG (a group beyond the wave's widest window is skipped).
// Measured on the headline workload (windows of 41..46 columns, mostly 44): groups of 8 -> 35.3 ms for the
// barcode stage, 2 -> 34.3, then at a later state 1 -> 28.0, 2 -> 27.05, 4 -> 26.6.
#ifndef BB_CG
#define BB_CG 4
#endif
// k_barcode_pfx: columns below this are processed without the per-group guard (kit windows are 41..63 columns wide;
// columns beyond a narrower window see base set 0 and their results are masked off).  Measured with 40: the larger
// basic block costs 59 spilled registers and 5 ms; 0 = every group guarded.
#ifndef BB_FIXED_COLS
#define BB_FIXED_COLS 0
#endif
// k_barcode_reg: register-resident, branch-free variant of k_barcode for m_bar <= 48 and windows of
// at most CW columns (CW = 48 or 64; all ONT kit presets).  Same arithmetic as k_barcode, but
//   * the two move bit-vectors of every column live in VGPRs (3 registers per column), written and
//     read with compile-time indices in fully unrolled column loops — no private memory;
//   * forward pass, traceback and replay are predicated arithmetic, not divergent branches; the
//     only branches are wave-uniform (skip 8-column chunks beyond the widest window in the wave);
//   * the traceback records the alignment per COLUMN: the text-consuming op of each column in two
//     bit planes, plus one bit per PATTERN ROW that was deleted (a run of Del moves inside a column
//     is found with one count-leading-ones instead of a loop);
//   * the Lodhi recurrence runs on power-of-two-scaled variables (b1 = 2^t a1, b2 = 2^t a2,
//     S = 2^t score): every multiply of the oracle's recurrence is by 0.5 (exact), so the scaling
//     commutes with the roundings of the adds and the result is bit-identical at 4 f64 adds per
//     match column; a run of nd Del columns is one exact ldexp;
//   * the per-hit argmax / runner-up uses 64-bit LDS atomics on the (monotone) score bit pattern.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ int clz64(unsigned long long x) { return x ? __clzll((long long)x) : 64; }
__device__ __forceinline__ int ctz64(unsigned long long x) { return x ? __ffsll((long long)x) - 1 : 64; }

// position (bit index) of the k-th (0-based) set bit of x; k < popcount(x)
__device__ __forceinline__ int select64(unsigned long long x, int k) {
    uint32_t w = (uint32_t)x;
    int base = 0;
    int pc = __popc(w);
    if (k >= pc) { k -= pc; w = (uint32_t)(x >> 32); base = 32; }
#pragma unroll
    for (int s = 16; s >= 1; s >>= 1) {
        const uint32_t lowm = (1u << s) - 1u;
        pc = __popc(w & lowm);
        const bool up = k >= pc;
        k -= up ? pc : 0;
        w = up ? (w >> s) : w;
        base += up ? s : 0;
    }
    return base;
}
__device__ __forceinline__ unsigned long long low64(int n) { return n >= 64 ? ~0ull : ((1ull << n) - 1ull); }

// map_pat_to_text_with_cost (cigar_parse.rs:6-68) in closed form from the traceback's masks, for
// pattern rows [rlo, rhi): plo/phi = text op of each column (bit c-1), diagrow = rows consumed by a
// Match/Sub, columns (tstart, best_pos] carry text ops.  Uses that an optimal alignment never has an
// Ins next to a Del.  Every pattern row has exactly one consuming op, so the pattern span is constant.
__device__ __forceinline__ void subpath_closed_form(unsigned long long plo, unsigned long long phi, unsigned long long diagrow,
                                                    int tstart, int best_pos, int m, int rlo, int rhi,
                                                    int32_t& txt_lo, int32_t& txt_hi, int32_t& bcost) {
    const unsigned long long CM = low64(best_pos) & ~low64(tstart);        // columns with a text op (bit c-1)
    const unsigned long long DG = CM & ~(phi & ~plo);                       // ... that consume a pattern row (not Ins)
    const unsigned long long NR = diagrow;                                  // rows consumed by those columns, same order
    const unsigned long long delrow = low64(m) & ~diagrow;
    // text position after the last non-deleted row below row r has been consumed
    auto pos_below = [&](int r) { const int k = __popcll(NR & low64(r)); return k == 0 ? tstart : select64(DG, k - 1) + 1; };
    txt_lo = pos_below(rlo);                                                // = tstart when rlo == 0
    {
        const int r = rhi - 1;
        if ((NR >> r) & 1ull) txt_hi = select64(DG, __popcll(NR & low64(r))) + 1;   // entry text idx = column - 1
        else txt_hi = pos_below(r) + 1;                                             // deleted: entry text idx = current position
    }
    const int ka = __popcll(NR & low64(rlo)), kb = __popcll(NR & low64(rhi));
    int32_t cost = __popcll(delrow & low64(rhi) & ~low64(rlo));
    if (kb > ka) {
        const int selA = ka == 0 ? tstart - 1 : select64(DG, ka - 1);
        const int selB = select64(DG, kb - 1);
        cost += __popcll((plo | phi) & low64(selB + 1) & ~low64(selA + 1));         // Sub / Ins entries in range
    }
    bcost = cost;
}

// Common tail of the register-resident barcode kernels, one call per block iteration (every lane of the block
// takes part: it synchronises): pass decision (searcher.rs:303-328), per-hit argmax = first maximum and runner-up
// by 64-bit LDS atomics on the score's bit pattern (searcher.rs:377,390-396), thresholds, and the row — tag row
// with the sub-path of the winning lane (cigar_parse.rs:6-68) or flank-only row (searcher.rs:241-265).
__device__ __forceinline__ void pick_and_emit(bool active, bool cand, int32_t best_cost, double s_norm, int p, int hl, const bb_hit& H,
                                              uint32_t hit_idx, const bb_group_dev& G, unsigned long long plo, unsigned long long phi,
                                              unsigned long long diagrow, int32_t tstart, int32_t best_pos, int32_t* s_cnt1,
                                              unsigned long long* s_max, unsigned long long* s_sec, int32_t* s_top, double min_score,
                                              double min_score_diff, bb_rowtmp* __restrict__ rows) {
    const int m = G.m_bar;
    const int32_t rlo = G.rel_lo, rhi = G.rel_hi;
    __syncthreads();
    if (active) {
        const bool pass2 = s_cnt1[hl] <= 1 && G.k1 < G.k2;
        cand = cand && (pass2 || best_cost <= G.k1);
    }
    const unsigned long long key = cand ? (unsigned long long)__double_as_longlong(s_norm) + 1ull : 0ull;
    if (cand) atomicMax(&s_max[hl], key);
    __syncthreads();
    if (cand && key == s_max[hl]) atomicMin(&s_top[hl], p);
    __syncthreads();
    if (active) {
        const int top = s_top[hl];
        if (cand && p != top) atomicMax(&s_sec[hl], key);
    }
    __syncthreads();
    if (active) {
        const int top = s_top[hl];
        const bool have = top != 0x7FFFFFFF;
        if ((have && p == top) || (!have && p == 0)) {
            bool valid = have && s_norm >= min_score;
            const unsigned long long sk = s_sec[hl];
            if (valid && sk != 0ull) valid = (s_norm - __longlong_as_double((long long)(sk - 1ull))) >= min_score_diff;
            const uint32_t read_len = H.read_len;
            bb_rowtmp R;
            bb_row& r = R.row;
            r.read_idx = H.read_idx; r.read_len = read_len;
            r.rel_dist_to_end = rel_dist_to_end((int64_t)H.text_start, (int64_t)read_len);
            r.read_start_flank = H.text_start; r.read_end_flank = H.text_end;
            r.flank_cost = H.cost; r.group_idx = H.group; r.strand = H.strand;
            r._pad[0] = 1; r._pad[1] = r._pad[2] = 0;
            if (valid) {
                int32_t txt_lo, txt_hi, bcost;
                subpath_closed_form(plo, phi, diagrow, tstart, best_pos, m, rlo, rhi, txt_lo, txt_hi, bcost);
                r.read_start_bar = H.ws + (uint32_t)txt_lo; r.read_end_bar = H.ws + (uint32_t)txt_hi;
                r.bar_start = H.ws + (uint32_t)rlo; r.bar_end = H.ws + (uint32_t)rhi;
                r.match_type = (uint8_t)G.type; r.barcode_cost = (int16_t)bcost; r.barcode_idx = (int16_t)top;
            } else {
                r.read_start_bar = H.text_start; r.read_end_bar = H.text_end;
                r.bar_start = 0; r.bar_end = 0;
                r.match_type = (uint8_t)(G.type == BB_FTAG ? BB_FFLANK : BB_RFLANK);
                r.barcode_cost = (int16_t)G.m_bar; r.barcode_idx = -1;
            }
            rows[hit_idx] = R;
        }
    }
}

// Lodhi (p = 3, lambda = 1/2) on the op planes of a traced path, oracle [H8]'s forward recurrence on power-of-two
// scaled variables: b1 = 2^t a1, b2 = 2^t a2 change only at Match columns and score += 2^-(t+1) * b2 (the
// product is exact, the fma rounds once like the oracle's add).  Columns (tstart, best_pos] carry the text ops
// (plo/phi bit c-1: 00 Match, 01 Sub, 10 Ins); delrow = pattern rows consumed by Del; the time t of a column's op
// counts the Dels before it.  Per column the work is three bit extractions from masks prepared once, the
// Del-run length after the column's row, and — on Match columns — three f64 operations.
// GEN (policy [H8] with decay exponents other than 1 per op; expk = one byte per op M, S, I, D): the time t advances by
// the column's exponent, a Match weighs 2^-(t + eM) — lambda stays 1/2, so every product is still exact.
template <int CW, bool GEN = false>
__device__ __forceinline__ double lodhi_replay(unsigned long long plo, unsigned long long phi, unsigned long long delrow,
                                               int32_t tstart, int32_t best_pos, int wmax, uint32_t expk = BB_LODHI_EXP_DEFAULT) {
    const unsigned long long onmask = low64(best_pos) & ~low64(tstart);  // bit c-1: column c carries an op
    const unsigned long long mmask = onmask & ~(plo | phi);                 // Match columns
    const unsigned long long amask = onmask & ~(phi & ~plo);                // the op consumes a pattern row (not Ins)
    const uint32_t on_w[2] = {(uint32_t)onmask, (uint32_t)(onmask >> 32)}, m_w[2] = {(uint32_t)mmask, (uint32_t)(mmask >> 32)},
                   a_w[2] = {(uint32_t)amask, (uint32_t)(amask >> 32)};
    // rows not consumed by Del; every bit from m up is set, so a shifted copy is never zero and a Del run that
    // reaches the last row ends at the sentinel
    const unsigned long long kept = ~delrow;
    double sc = 0.0, b1 = 0.0, b2 = 0.0;
    int32_t pj = onmask ? __builtin_ctzll(kept) : 0;  // leading Dels
    const uint32_t eM = expk & 0xFFu, eD = expk >> 24;
    const uint32_t lo_w[2] = {(uint32_t)plo, (uint32_t)(plo >> 32)}, hi_w[2] = {(uint32_t)phi, (uint32_t)(phi >> 32)};
    // high dword of 2^t, advanced with t; 2^-(t+1) has (1022 - t) << 20 = 0x7FD00000 - (t << 20) there
    uint32_t e_hi = (uint32_t)(1023 + (GEN ? pj * (int32_t)eD : pj)) << 20;
    const uint32_t w_base = GEN ? 0x7FE00000u - (eM << 20) : 0x7FD00000u;
#ifdef BB_REPLAY_FULL_UNROLL
#pragma clang loop unroll(full)
#else
#pragma unroll
#endif
    for (int c0 = 1; c0 <= CW; c0 += BB_CG) {
        if (c0 <= wmax) {  // wave-uniform
#pragma unroll
            for (int c = c0; c < c0 + BB_CG; ++c) {
                const int k = c - 1;
                const uint32_t onb = (on_w[k >> 5] >> (k & 31)) & 1u, ab = (a_w[k >> 5] >> (k & 31)) & 1u;
                if ((m_w[k >> 5] >> (k & 31)) & 1u) {
                    const double w = __hiloint2double((int)(w_base - e_hi), 0);  // 2^-(t+1) (GEN: 2^-(t+eM))
                    const double pw = __hiloint2double((int)e_hi, 0);                               // 2^t
                    sc = __fma_rn(w, b2, sc); b2 = b2 + b1; b1 = b1 + pw;
                }
                pj += (int32_t)ab;
                // Dels that follow this column's op; 0 by itself outside (tstart, best_pos] and on Ins columns,
                // where pj rests on a kept row (or on the sentinel at m)
                const int32_t nd = __builtin_ctzll(kept >> pj);
                pj += nd;
                if constexpr (GEN) {
                    const uint32_t code = ((lo_w[k >> 5] >> (k & 31)) & 1u) | (((hi_w[k >> 5] >> (k & 31)) & 1u) << 1);  // 0 Match, 1 Sub, 2 Ins
                    e_hi += ((onb ? (expk >> (8u * code)) & 0xFFu : 0u) + (uint32_t)nd * eD) << 20;
                } else e_hi += (onb + (uint32_t)nd) << 20;
            }
        }
    }
    return sc;
}

// Upper bound of the Lodhi score of a traced path from its COLUMN planes alone: the same recurrence on the string of
// text-consuming ops only (the Del ops dropped).  Dropping ops can only shorten the span of a match triple, and every
// triple's weight 2^-(span) only grows — so the value is >= the exact score of lodhi_replay, up to f64 rounding (the
// caller keeps a margin).  Time = column index (only differences of times enter), so every power of two is a
// compile-time constant and nothing of the per-column Del bookkeeping of the exact replay is left: a bit test and three
// f64 operations per Match column.
template <int CW>
__device__ __forceinline__ float lodhi_bound(unsigned long long plo, unsigned long long phi, int32_t tstart, int32_t best_pos, int wmax) {
    // f32 (full-rate v_fma_f32 / v_add_f32; f64 is half rate) and branch-free: a column that is not a Match adds zeros.
    // All terms are positive, every operation rounds to nearest with relative error <= 2^-24, fewer than 200 of them
    // enter any result: the computed value is within a factor (1 +- 2^-16) of the real one; the return value is
    // scaled up by (1 + 2^-14) so that it stays an upper bound.
    const unsigned long long mmask = low64(best_pos) & ~low64(tstart) & ~(plo | phi);   // Match columns (bit c-1)
    const uint32_t m_w[2] = {(uint32_t)mmask, (uint32_t)(mmask >> 32)};
    float sc = 0.0f, b1 = 0.0f, b2 = 0.0f;
#pragma unroll
    for (int c0 = 1; c0 <= CW; c0 += BB_CG) {
        if (c0 <= BB_FIXED_COLS || c0 <= wmax) {  // wave-uniform
#pragma unroll
            for (int c = c0; c < c0 + BB_CG; ++c) {
                const int k = c - 1;
                const uint32_t on = 0u - ((m_w[k >> 5] >> (k & 31)) & 1u);                        // all ones on a Match column
                const float w = __uint_as_float(((uint32_t)(126 - c) << 23) & on);                // 2^-(c+1) or 0
                const float pw = __uint_as_float(((uint32_t)(127 + c) << 23) & on);               // 2^c or 0
                const float b1m = __uint_as_float(__float_as_uint(b1) & on);
                sc = __fmaf_rn(w, b2, sc); b2 = b2 + b1m; b1 = b1 + pw;
            }
        }
    }
    return sc * (1.0f + 1.0f / 16384.0f);
}

// The same bound, eight columns at a time.  Over the columns 8q+1 .. 8q+8 the recurrence is affine in (sc, b2, b1), and
// with u2 = b2 / 2^8q, u1 = b1 / 2^8q its coefficients depend on the byte of Match bits only:
//   sc += A u2 + B u1 + C;   u2 = (u2 + n u1 + D) / 256;   u1 = (u1 + E) / 256
// (A = sum 2^-(r+1), B = sum 2^-(r+1) cnt(r), C = sum 2^-(r+1) P2(r) over the byte's Match positions r = 1..8, with cnt(r) the
// Matches before r, P1(r) = sum of 2^r' over them, P2(r) = sum of P1 over them; n = all Matches, D = P2(9), E = P1(9)).
// One 32-byte table entry and eight f32 operations per byte instead of nine instructions per column; the entries are
// rounded up, every term is positive, fewer than 60 roundings enter a result: the (1 + 2^-14) scale keeps it a bound.
struct __attribute__((aligned(32))) bb_lb_entry { float A, B, C, n, D, E, _p0, _p1; };
__device__ __forceinline__ void lodhi_bound_table_entry(uint32_t byte, bb_lb_entry& e) {
    double A = 0.0, B = 0.0, C = 0.0, D = 0.0, E = 0.0, cnt = 0.0, P1 = 0.0, P2 = 0.0;
    for (int r = 1; r <= 8; ++r) {
        if ((byte >> (r - 1)) & 1u) {
            const double w = __hiloint2double((int)((uint32_t)(1023 - (r + 1)) << 20), 0);  // 2^-(r+1)
            A += w; B += w * cnt; C += w * P2;
            P2 += P1; cnt += 1.0; P1 += (double)(1u << r);
        }
    }
    D = P2; E = P1;
    e.A = __double2float_ru(A); e.B = __double2float_ru(B); e.C = __double2float_ru(C); e.n = (float)cnt;
    e.D = __double2float_ru(D); e.E = __double2float_ru(E); e._p0 = 0.0f; e._p1 = 0.0f;
}
template <int CW>
__device__ __forceinline__ float lodhi_bound_tab(unsigned long long plo, unsigned long long phi, int32_t tstart, int32_t best_pos, int wmax,
                                                 const bb_lb_entry* tab) {
    const unsigned long long mmask = low64(best_pos) & ~low64(tstart) & ~(plo | phi);   // Match columns (bit c-1)
    const uint32_t m_w[2] = {(uint32_t)mmask, (uint32_t)(mmask >> 32)};
    float sc = 0.0f, u1 = 0.0f, u2 = 0.0f;
#pragma unroll
    for (int q = 0; q < CW / 8; ++q) {
        if (8 * q < wmax) {  // wave-uniform
            const uint32_t byte = (m_w[q >> 2] >> (8 * (q & 3))) & 0xFFu;
            const float4 t0 = *reinterpret_cast<const float4*>(&tab[byte].A);
            const float2 t1 = *reinterpret_cast<const float2*>(&tab[byte].D);
            sc = __fmaf_rn(t0.x, u2, __fmaf_rn(t0.y, u1, sc + t0.z));
            u2 = (__fmaf_rn(t0.w, u1, u2) + t1.x) * (1.0f / 256.0f);
            u1 = (u1 + t1.y) * (1.0f / 256.0f);
        }
    }
    return sc * (1.0f + 1.0f / 16384.0f);
}

// Policy [H1] / [H7] on the column masks of a lane's bottom row (P / M bit q: the cost rises / falls going from end
// position q to q+1, positions 0..wn, cost m at position 0): the reported positions bit-parallel, then the first
// strictly-lowest of them (searcher.rs:294-300) or the last lowest, then — plateaus at their left end — the position
// after the last change below it.
__device__ __forceinline__ void pick_minimum(unsigned long long P, unsigned long long M, int wn, int m, bool active, int pol_lm, bool tie_last,
                                             int32_t& best_cost, int32_t& best_pos) {

    // dec(q) = "last strict change before position q was a decrease" (initially true):
    // dec(q+1) = M[q] | (~(P|M)[q] & dec(q))  ==  carry chain of (M | ~P) + M + 1;  strict minima only: dec(q+1) = M[q]
    const unsigned long long A = M | ~P;
    const unsigned long long D = pol_lm == BB_LM_STRICT ? (M << 1) | 1ull : (A + M + 1ull) ^ A ^ M;   // bit q = dec(q)
    unsigned long long R = (P & D) | (D & (1ull << wn));    // reported positions (plateau right ends, or the window end)
    if (!active) R = 0ull;
    while (R) {  // 1-4 iterations
        const int q = ctz64(R);
        R &= R - 1ull;
        const unsigned long long lowq = (1ull << q) - 1ull;
        const int32_t cq = m + __popcll(P & lowq) - __popcll(M & lowq);
        if (cq - (tie_last ? 1 : 0) < best_cost) { best_cost = cq; best_pos = q; }  // tie_last: cq <= best_cost
    }
    if (pol_lm == BB_LM_PLATEAU_LEFT && best_pos > 0) {
        const unsigned long long ch = (P | M) & ((1ull << best_pos) - 1ull);
        best_pos = ch ? 64 - clz64(ch) : 0;
    }
}

template <int WB, int CW>
__global__ __launch_bounds__(512) void k_barcode_reg(const uint8_t* __restrict__ tables, const bb_group_dev* __restrict__ groups,
                                                     uint32_t g, const bb_hit* __restrict__ hits, const uint32_t* __restrict__ hit_list,
                                                     const uint32_t* __restrict__ list_cnt, uint32_t n_hits_all, uint32_t hpb,
                                                     double min_score, double min_score_diff, bb_rowtmp* __restrict__ rows) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const bb_group_dev G = groups[g];
    const uint32_t n_list = hit_list ? list_cnt[g] : n_hits_all;
    const uint32_t n_iter = (n_list + hpb - 1) / hpb;
    if (blockIdx.x >= n_iter) return;
    const int N = G.n_seqs, m = G.m_bar;
    // LDS carve: [hit records: hpb x 96 B][max u64[hpb]][second u64[hpb]][cnt1 i32[hpb]][top i32[hpb]][peq 2*16*N*WB words]
    uint4* s_hit = reinterpret_cast<uint4*>(smem);
    size_t o = (size_t)hpb * sizeof(bb_hit);
    unsigned long long* s_max = reinterpret_cast<unsigned long long*>(smem + o);
    o += (size_t)hpb * 8;
    unsigned long long* s_sec = reinterpret_cast<unsigned long long*>(smem + o);
    o += (size_t)hpb * 8;
    int32_t* s_cnt1 = reinterpret_cast<int32_t*>(smem + o);
    o += (size_t)hpb * 4;
    int32_t* s_top = reinterpret_cast<int32_t*>(smem + o);
    o += (size_t)hpb * 4;
    o = (o + 15) & ~(size_t)15;
    uint32_t* s_peq = reinterpret_cast<uint32_t*>(smem + o);
    {   // barcode Peq of both strands: loaded once per (persistent) block
        const uint32_t* gp = reinterpret_cast<const uint32_t*>(tables + G.off_peq_bar[0]);
        const int words = 2 * 16 * N * WB;
        for (int i = threadIdx.x; i < words; i += blockDim.x) s_peq[i] = gp[i];
    }
    const int hl = threadIdx.x / N;
    const int p = threadIdx.x - hl * N;
    const bool in_blk = hl < (int)hpb;
    const int hls = in_blk ? hl : 0;  // lanes past the last hit of the block shadow hit 0, results unused
    constexpr int PIECES = (int)(sizeof(bb_hit) / 16);
    // prefetch of the next iteration's hit records: the lanes of a hit share its six 16-byte pieces
    // (piece p, p+N, p+2N: one piece per lane when N >= 6, up to three for the smallest groups)
    uint4 pre[3] = {make_uint4(0u, 0u, 0u, 0u), make_uint4(0u, 0u, 0u, 0u), make_uint4(0u, 0u, 0u, 0u)};
    auto prefetch = [&](uint32_t it) {
        const uint32_t li = it * hpb + (uint32_t)hl;
        if (in_blk && p < PIECES && it < n_iter && li < n_list) {
            const uint32_t idx = hit_list ? hit_list[li] : li;
            const uint4* src = reinterpret_cast<const uint4*>(hits + idx);
#pragma unroll
            for (int q = 0; q < 3; ++q)
                if (p + q * N < PIECES) pre[q] = src[p + q * N];
        }
    };
    prefetch(blockIdx.x);
  for (uint32_t it = blockIdx.x; it < n_iter; it += gridDim.x) {
    const uint32_t li = it * hpb + (uint32_t)hl;
    const bool exists = in_blk && li < n_list;
    if (exists && p < PIECES) {
#pragma unroll
        for (int q = 0; q < 3; ++q)
            if (p + q * N < PIECES) s_hit[hl * PIECES + p + q * N] = pre[q];
    }
    if (in_blk && p == 0) { s_max[hl] = 0ull; s_sec[hl] = 0ull; s_cnt1[hl] = 0; s_top[hl] = 0x7FFFFFFF; }
    __syncthreads();
    const uint32_t hit_idx = hit_list ? (exists ? hit_list[li] : 0u) : li;
    prefetch(it + gridDim.x);  // in flight during this iteration's compute
    const bb_hit* Hs = reinterpret_cast<const bb_hit*>(s_hit + hls * PIECES);
    bb_hit H;  // header only
    {
        const uint4 h0 = s_hit[hls * PIECES], h1 = s_hit[hls * PIECES + 1];
        H.read_idx = h0.x; H.text_start = h0.y; H.text_end = h0.z; H.ws = h0.w;
        H.we = h1.x; H.cost = (int16_t)(h1.y & 0xFFFFu); H.group = (uint8_t)((h1.y >> 16) & 0xFFu); H.strand = (uint8_t)(h1.y >> 24);
        H.valid = (uint8_t)(h1.z & 0xFFu); H.read_len = h1.w;
    }
    (void)Hs;
    bool active = exists && H.valid != 0;
    if (exists && !H.valid && p == 0) rows[hit_idx].row._pad[0] = 0;
    const int32_t wn = active ? (int32_t)(H.we - H.ws) : 0;

    int wmax = wn;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) wmax = max(wmax, __shfl_xor(wmax, d, 64));
    wmax = __builtin_amdgcn_readfirstlane(wmax);

    // ---- forward pass: Myers + move bits; columns unrolled; all state in registers.  The bottom-row
    // score is not tracked per column: its +1/-1 deltas are collected in two 64-bit column masks and the
    // local-minimum rule (oracle [H1]) is resolved bit-parallel after the loop. ----
    uint32_t L0[CW], H0[CW], X[CW];
    int32_t best_cost = 0x7FFFFFFF, best_pos = -1;
    {
        uint32_t wc[CW / 4];
#pragma unroll
        for (int q = 0; q < CW / 16; ++q) { uint4 v = s_hit[hls * PIECES + 2 + q]; wc[4 * q] = v.x; wc[4 * q + 1] = v.y; wc[4 * q + 2] = v.z; wc[4 * q + 3] = v.w; }
        const uint32_t NW = (uint32_t)(N * WB);
        // byte offset into s_peq of this lane's column 0 entry; one v_mad_u32_u24 per column adds code * row bytes
        const uint32_t pb4 = ((uint32_t)((active ? H.strand : 0) * 16) * NW + (uint32_t)p * WB) * 4u, NW4 = NW * 4u;
        const uint8_t* s_peq_b = reinterpret_cast<const uint8_t*>(s_peq);
        uint32_t pv[WB], mv[WB];
#pragma unroll
        for (int x = 0; x < WB; ++x) { int bits = m - 32 * x; pv[x] = bits >= 32 ? 0xFFFFFFFFu : (bits > 0 ? ((1u << bits) - 1u) : 0u); mv[x] = 0; }
        const int TBS = 31 - ((m - 1) & 31);  // shift that brings the bottom row's bit to bit 31
        // bottom-row deltas, newest column at bit 0 (one shift + one v_alignbit per column and plane); the
        // column order is restored after the loop.  Bit c of up/dn: score rises / falls going from position c to c+1
        uint32_t upr[2] = {0u, 0u}, dnr[2] = {0u, 0u};
#pragma unroll
        for (int c0 = 0; c0 < CW; c0 += BB_CG) {
            if (c0 < wmax) {  // wave-uniform
#pragma unroll
                for (int c = c0; c < c0 + BB_CG; ++c) {
                    const uint32_t code = (wc[c >> 2] >> (8 * (c & 3))) & 0xFu;
                    uint32_t eq[WB], d0[WB], ph[WB], mh[WB], l[WB], hh[WB];
                    const uint32_t ei = __umul24(code, NW4) + pb4;
                    if constexpr (WB == 2) { uint2 v = *reinterpret_cast<const uint2*>(s_peq_b + ei); eq[0] = v.x; eq[1] = v.y; }
                    else eq[0] = *reinterpret_cast<const uint32_t*>(s_peq_b + ei);
                    myers_step<WB>(pv, mv, eq, d0, ph, mh);
                    move_bits<WB>(eq, d0, ph, l, hh);
                    // stored bit-reversed (row r <-> bit 64-r of {L0|H0 : X-part}) for the one-hot traceback below
                    L0[c] = __brev(l[0]); H0[c] = __brev(hh[0]);
                    if constexpr (WB == 2) X[c] = (__brev(l[1]) >> 16) | (__brev(hh[1]) & 0xFFFF0000u);
                    else X[c] = 0;
                    upr[c >> 5] = __builtin_amdgcn_alignbit(upr[c >> 5], ph[WB - 1] << TBS, 31);
                    dnr[c >> 5] = __builtin_amdgcn_alignbit(dnr[c >> 5], mh[WB - 1] << TBS, 31);
                }
            }
        }
        const int pc = min(CW, ((wmax + BB_CG - 1) / BB_CG) * BB_CG);  // columns processed (wave-uniform)
        const int n0 = min(pc, 32), n1 = pc - n0;
        uint32_t up[2], dn[2];
        up[0] = n0 ? __brev(upr[0]) >> (32 - n0) : 0u; dn[0] = n0 ? __brev(dnr[0]) >> (32 - n0) : 0u;
        up[1] = n1 ? __brev(upr[1]) >> (32 - n1) : 0u; dn[1] = n1 ? __brev(dnr[1]) >> (32 - n1) : 0u;
        // positions 0..wn; deltas of columns >= wn are garbage and masked off
        const unsigned long long wmask = wn >= 64 ? ~0ull : ((1ull << wn) - 1ull);
        const unsigned long long P = (((unsigned long long)up[1] << 32) | up[0]) & wmask;
        const unsigned long long M = (((unsigned long long)dn[1] << 32) | dn[0]) & wmask;
        pick_minimum(P, M, wn, m, active, G.pol_lm, G.pol_tie_last != 0, best_cost, best_pos);
        if (active && best_pos >= 0 && best_cost <= G.k1) atomicAdd(&s_cnt1[hl], 1);
    }
    // Every lane with a local minimum traces and scores (a wave executes those instructions for all
    // its lanes anyway); which of them are candidates — pass 1 (<= k1) or the deeper pass 2
    // (searcher.rs:303-328) — is decided after the block-wide count below.
    bool cand = active && best_pos >= 0 && best_cost <= G.k2;
    // ---- traceback, one predicated step per column, on a ONE-HOT row cursor over bit-reversed move
    // vectors (row r <-> bit 64-r).  With rows running towards higher bits, skipping a run of Del moves
    // is one addition: the carry ripples through the run's ones and stops at the first non-Del row,
    // nb = (Dr + b) & ~Dr.  A Match/Sub moves the cursor one row (b << 1), an Ins keeps it; the cursor
    // falls off the top (b = 0) when row 1 has been consumed.  Outputs: the text op of each column in
    // two bit planes, the rows consumed by a Match/Sub, the number of columns with a text op.
    // Once every cursor of the wave is in the high word (rows <= 32) the step runs on 32-bit words.
    unsigned long long plo = 0ull, phi = 0ull;
    uint32_t b_lo = 0u, b_hi = 0u, dg_lo = 0u, dg_hi = 0u;   // cursor and consumed rows, bit-reversed
    int32_t ntext = 0;
    const uint32_t start_lo = m > 32 ? (1u << (64 - m)) : 0u, start_hi = m > 32 ? 0u : (1u << (32 - m));
#pragma unroll
    for (int c0 = CW; c0 >= 8; c0 -= 8) {
        if (c0 - 7 <= wmax) {  // wave-uniform
            if (WB == 1 || __all(b_lo == 0u && (!cand || best_pos > c0))) {
#pragma unroll
                for (int c = c0; c > c0 - 8; --c) {
                    if constexpr (WB == 1) b_hi = (cand & (best_pos == c)) ? start_hi : b_hi;
                    const uint32_t Lr = L0[c - 1], Hr = H0[c - 1];
                    const uint32_t Dr = Lr & Hr;
                    const uint32_t nb = (Dr + b_hi) & ~Dr;
                    const bool has = nb != 0u, lo = (Lr & nb) != 0u, hi = (Hr & nb) != 0u;
                    plo |= lo ? (1ull << (c - 1)) : 0ull;
                    phi |= hi ? (1ull << (c - 1)) : 0ull;
                    const bool consume = has & !hi;
                    dg_hi |= consume ? nb : 0u;
                    b_hi = consume ? (nb << 1) : nb;
                }
            } else {
#pragma unroll
                for (int c = c0; c > c0 - 8; --c) {
                    const bool st = cand & (best_pos == c);
                    b_lo = st ? start_lo : b_lo;
                    b_hi = st ? start_hi : b_hi;
                    const uint32_t Lr_hi = L0[c - 1], Hr_hi = H0[c - 1], Lr_lo = X[c - 1] << 16, Hr_lo = X[c - 1] & 0xFFFF0000u;
                    const unsigned long long Dr = ((unsigned long long)(Lr_hi & Hr_hi) << 32) | (Lr_lo & Hr_lo);
                    const unsigned long long bb = ((unsigned long long)b_hi << 32) | b_lo;
                    const unsigned long long nb = (Dr + bb) & ~Dr;
                    const uint32_t nb_lo = (uint32_t)nb, nb_hi = (uint32_t)(nb >> 32);
                    const bool has = nb != 0ull;
                    const bool lo = ((Lr_lo & nb_lo) | (Lr_hi & nb_hi)) != 0u, hi = ((Hr_lo & nb_lo) | (Hr_hi & nb_hi)) != 0u;
                    plo |= lo ? (1ull << (c - 1)) : 0ull;
                    phi |= hi ? (1ull << (c - 1)) : 0ull;
                    const bool consume = has & !hi;
                    dg_lo |= consume ? nb_lo : 0u;
                    dg_hi |= consume ? nb_hi : 0u;
                    const unsigned long long nx = consume ? (nb << 1) : nb;
                    b_lo = (uint32_t)nx; b_hi = (uint32_t)(nx >> 32);
                }
            }
        }
    }
    // text ops = rows consumed by a Match/Sub + Ins columns
    ntext = cand ? __popc(dg_lo) + __popc(dg_hi) + __popcll(phi & ~plo) : 0;
    const int32_t tstart = cand ? best_pos - ntext : 0;   // columns (tstart, best_pos] carry the text ops
    // consumed rows back in natural order (row r <-> bit r-1); rows never consumed were deleted
    const unsigned long long diagrow = ((unsigned long long)__brev(dg_lo) << 32) | __brev(dg_hi);
    const unsigned long long delrow = cand ? (low64(m) & ~diagrow) : 0ull;
    // ---- forward replay: Lodhi only.  Scaled recurrence (see header): b1 = 2^t a1, b2 = 2^t a2 change
    // only at match columns; score += 2^-(t+1) * b2 (exact scaling, same rounding as the oracle's add).
    double s_norm = -1.0;
    {
        const bool on = cand;  // the loop is wave-uniform: idle lanes walk it with empty masks
        const double sc = (uint32_t)G.pol_lodhi_exp == (uint32_t)BB_LODHI_EXP_DEFAULT
                              ? lodhi_replay<CW>(on ? plo : 0ull, on ? phi : 0ull, on ? delrow : 0ull, on ? tstart : 0, on ? best_pos : 0, wmax)
                              : lodhi_replay<CW, true>(on ? plo : 0ull, on ? phi : 0ull, on ? delrow : 0ull, on ? tstart : 0, on ? best_pos : 0, wmax, (uint32_t)G.pol_lodhi_exp);
        if (cand) s_norm = G.perfect > 0.0 ? sc / G.perfect : 0.0;
    }
    // ---- pass decision (searcher.rs:303-328), then per-hit argmax (first maximum) and runner-up:
    // searcher.rs:377,390-396 ----
    pick_and_emit(active, cand, best_cost, s_norm, p, hl, H, hit_idx, G, plo, phi, diagrow, tstart, best_pos, s_cnt1, s_max, s_sec, s_top,
                  min_score, min_score_diff, rows);
    __syncthreads();  // LDS hit records / reduction cells are rewritten by the next iteration
  }
}

// ------------------------------------------------------------------------------------------------
// Shared-prefix split of the barcode stage (groups with bb_group_dev::pfx > 0, e.g. SQK-NBD114-96: 42-row
// padded barcodes = 10 shared pad rows + 32 rows per barcode).
//
// The first pfx rows of the DP matrix are the same for every barcode of a group (same pattern characters,
// same window), so they are computed once per hit by k_bar_prefix (one lane per hit) and every barcode lane of
// k_barcode_pfx runs Myers on ONE 32-bit word (rows pfx+1..m) with the horizontal delta of row pfx as its
// carry-in (Hyyro's block step: hin < 0 sets bit 0 of Eq for the diagonal-zero vector, the shifted Ph/Mh take
// hin as their bit 0).  Values are those of the monolithic two-word column step: both are the DP matrix.
// ------------------------------------------------------------------------------------------------
// 128 hits per block; records enter and leave through LDS so that global traffic is whole lines (a lane-per-
// record access pattern with 96-byte / 272-byte strides moved 4 GB per 2.6 M hits instead of ~1 GB).
__global__ __launch_bounds__(128) void k_bar_prefix(const uint8_t* __restrict__ tables, const bb_group_dev* __restrict__ groups,
                                                    const bb_hit* __restrict__ hits, uint32_t n_hits, bb_hit_pfx* __restrict__ out, uint32_t n_groups) {
    constexpr int HW = (int)(sizeof(bb_hit) / 4), OW = (int)(sizeof(bb_hit_pfx) / 4), OS = OW + 1;  // odd row stride: no bank conflicts
    __shared__ uint32_t s_in[128 * (HW + 1)];
    __shared__ uint32_t s_out[128 * OS];
    __shared__ uint32_t s_eqt[BB_MAX_GROUPS * 2 * 16];  // Peq of the leading shared rows per (group, strand, base set)
    __shared__ uint8_t s_tlut[BB_MAX_GROUPS * 2 * 16];  // trailing rows matched per (group, strand, base set)
    for (uint32_t i = threadIdx.x; i < n_groups * 32u; i += 128u) {
        const bb_group_dev& Gi = groups[i >> 5];
        const uint32_t st = (i >> 4) & 1u, code = i & 15u;
        const bool sp = Gi.split[st] != 0;
        s_eqt[i] = sp ? reinterpret_cast<const uint32_t*>(tables + Gi.off_peq_pfx[st])[code] : 0u;
        s_tlut[i] = sp ? (tables + Gi.off_tail_lut[st])[code] : (uint8_t)0;
    }
    const uint32_t b0 = blockIdx.x * 128u;
    const uint32_t nb = min(128u, n_hits - b0);
    {
        const uint32_t* src = reinterpret_cast<const uint32_t*>(hits + b0);
        for (uint32_t i = threadIdx.x; i < nb * HW; i += 128u) s_in[(i / HW) * (HW + 1) + (i % HW)] = src[i];
    }
    __syncthreads();
    const uint32_t t = threadIdx.x;
    const uint32_t* rec = s_in + t * (HW + 1);
    uint32_t* orow = s_out + t * OS;
    bool did = false;
    if (t < nb) {
        const uint32_t ws = rec[3], we = rec[4], grp = (rec[5] >> 16) & 0xFFu, strand = rec[5] >> 24, valid = rec[6] & 0xFFu;
        const bb_group_dev& G = groups[grp];
        const int32_t wn = (int32_t)(we - ws);
        if (valid && G.split[strand & 1u] && wn <= 64) {  // wide windows do not use the split
            did = true;
            const int P = G.pfx[strand & 1u], T = G.tail[strand & 1u];
            constexpr int SH0 = 4 + 2 * BB_MAX_TAIL;  // word index of sh[0] in the record
            const uint32_t* eqt = s_eqt + (grp * 2u + (strand & 1u)) * 16u;   // LDS lookups (a 16-way select per column cost 32 instructions)
            const uint8_t* tlut = s_tlut + (grp * 2u + (strand & 1u)) * 16u;
            uint32_t pv = P ? (P >= 32 ? 0xFFFFFFFFu : (1u << P) - 1u) : 0u, mv = 0u;
            unsigned long long PH = 0ull, MH = 0ull, TE[BB_MAX_TAIL];
#pragma unroll
            for (int q = 0; q < BB_MAX_TAIL; ++q) TE[q] = 0ull;
            for (int c = 0; c < wn; ++c) {
                const uint32_t code = (rec[8 + (c >> 2)] >> (8 * (c & 3))) & 0xFu;
                const uint32_t eq = eqt[code];
                if (T > 0) {
                    const uint32_t tb = tlut[code];
#pragma unroll
                    for (int q = 0; q < BB_MAX_TAIL; ++q) TE[q] |= (unsigned long long)((tb >> q) & 1u) << c;
                }
                uint32_t shw = 0u;
                if (P > 0) {
                    const uint32_t x = eq & pv;
                    const uint32_t d0 = (((x + pv) ^ pv) | eq | mv);
                    const uint32_t ph = mv | ~(d0 | pv), mh = pv & d0;
                    PH |= (unsigned long long)((ph >> (P - 1)) & 1u) << c;
                    MH |= (unsigned long long)((mh >> (P - 1)) & 1u) << c;
                    const uint32_t isM = d0 & eq, l = ~(isM | ph), hh = (ph & ~isM) | (l & d0);
                    shw = (__brev(l) >> (32 - P)) | ((__brev(hh) >> (32 - P)) << 16);  // row r <-> bit P - r
                    const uint32_t phs = shl1_32(ph), mhs = shl1_32(mh);  // top boundary row: D[0][c] = 0, no horizontal delta
                    pv = mhs | ~(d0 | phs);
                    mv = phs & d0;
                }
                orow[SH0 + c] = shw;
            }
            for (int c = wn; c < 64; ++c) orow[SH0 + c] = 0u;
            orow[0] = (uint32_t)PH; orow[1] = (uint32_t)(PH >> 32); orow[2] = (uint32_t)MH; orow[3] = (uint32_t)(MH >> 32);
#pragma unroll
            for (int q = 0; q < BB_MAX_TAIL; ++q) { orow[4 + 2 * q] = q < T ? (uint32_t)TE[q] : 0u; orow[5 + 2 * q] = q < T ? (uint32_t)(TE[q] >> 32) : 0u; }
        }
    }
    if (!did) for (int i = 0; i < OW; ++i) orow[i] = 0u;
    __syncthreads();
    uint32_t* dst = reinterpret_cast<uint32_t*>(out + b0);
    for (uint32_t i = threadIdx.x; i < nb * OW; i += 128u) dst[i] = s_out[(i / OW) * OS + (i % OW)];
}

// Wave-wide maximum of a u32 on the VALU's data-parallel primitives (no LDS): quad swaps, half-row and row mirrors give
// every lane of a 16-lane row the row's maximum, two row broadcasts carry it to the last row; the result is lane 63's.
__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v) {
    v = max(v, (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0xB1, 0xF, 0xF, false));   // quad_perm [1,0,3,2]
    v = max(v, (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x4E, 0xF, 0xF, false));   // quad_perm [2,3,0,1]
    v = max(v, (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x141, 0xF, 0xF, false));  // row_half_mirror
    v = max(v, (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x140, 0xF, 0xF, false));  // row_mirror
    v = max(v, (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x142, 0xA, 0xF, false));  // row_bcast15 -> rows 1, 3
    v = max(v, (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x143, 0xC, 0xF, false));  // row_bcast31 -> rows 2, 3
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}
// The two largest keys among the wave's lanes with `in` (keys = value bits : 0xFFFF - p with p ascending along the lanes
// of a hit, so the first lane holding the largest value also holds the largest key); 0 where there is none.  Wave-uniform.
__device__ __forceinline__ void wave_top2(bool in, uint32_t vbits, unsigned long long key, unsigned long long& k1, unsigned long long& k2) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t v = in ? vbits + 1u : 0u;  // members are > 0 (value bits are those of a finite non-negative float)
    const uint32_t m1 = wave_max_u32(v);
    k1 = 0ull; k2 = 0ull;
    if (m1 == 0u) return;
    const int l1 = (int)__ffsll((long long)__ballot(v == m1)) - 1;
    k1 = ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(key >> 32), l1) << 32) |
         (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)key, l1);
    const uint32_t v2 = (int)lane == l1 ? 0u : v;
    const uint32_t m2 = wave_max_u32(v2);
    if (m2 == 0u) return;
    const int l2 = (int)__ffsll((long long)__ballot(v2 == m2)) - 1;
    k2 = ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(key >> 32), l2) << 32) |
         (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)key, l2);
}
#define BB_PFX_SYNC() __syncthreads()
// DEFPOL: the default local-minimum and tie rules as compile-time constants (measured: the run-time form costs the 48-column
// fast variants 1 % — 16.40 against 16.24 ms per 2 M-read step); the host launches it when the context's policy has them
template <int CW, bool TAIL, bool FAST, bool DEFPOL = false>
__global__ __launch_bounds__(CW <= 48 ? 768 : 512) void k_barcode_pfx(const uint8_t* __restrict__ tables, const bb_group_dev* __restrict__ groups,
                                                     uint32_t g, uint32_t strand, const bb_hit* __restrict__ hits, const bb_hit_pfx* __restrict__ pfxs,
                                                     const uint32_t* __restrict__ hit_list, const uint32_t* __restrict__ list_cnt,
                                                     uint32_t n_hits_all, uint32_t hpb, double min_score, double min_score_diff,
                                                     bb_rowtmp* __restrict__ rows) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const bb_group_dev G = groups[g];
    const uint32_t n_list = hit_list ? list_cnt[g] : n_hits_all;
    const uint32_t n_iter = (n_list + hpb - 1) / hpb;
    if (blockIdx.x >= n_iter) return;
    const int N = G.n_seqs, m = G.m_bar, P = groups[g].pfx[strand], T = TAIL ? groups[g].tail[strand] : 0;  // scalar loads: no dynamic index into G
    // rows per lane: 32 = m_bar - P - T (row P+1 <-> bit 31 of the bit-reversed planes, row P+32 <-> bit 0)
    constexpr int PIECES_H = (int)(sizeof(bb_hit) / 16), PIECES_P = (int)(sizeof(bb_hit_pfx) / 16), PIECES = PIECES_H + PIECES_P;
    constexpr int SH_PIECE = PIECES_H + 1 + BB_MAX_TAIL / 2;  // first piece of sh[] inside a hit's record pair
    // LDS carve: [hit + prefix records: hpb x 400 B][max u64[hpb]][second u64[hpb]][cnt1 i32[hpb]][top i32[hpb]][walk table]
    // [peq 16*N words][move planes of the trailing rows: T x 2 x blockDim u64]
    // Everything a set of hpb hits owns exists twice ([2][..]): while the lanes work on one set, the next set's records
    // land in the other half and its per-column tables are built there, so an iteration needs two barriers, not five.
    uint4* s_hit2 = reinterpret_cast<uint4*>(smem);
    size_t o = (size_t)2 * hpb * PIECES * 16;
    unsigned long long* s_max2 = reinterpret_cast<unsigned long long*>(smem + o);
    o += (size_t)2 * hpb * 8;
    unsigned long long* s_sec2 = reinterpret_cast<unsigned long long*>(smem + o);
    o += (size_t)2 * hpb * 8;
    unsigned long long* s_maxB2 = reinterpret_cast<unsigned long long*>(smem + o);  // fast variant: top-2 of the pass-2 candidate set
    o += (size_t)2 * hpb * 8;
    unsigned long long* s_secB2 = reinterpret_cast<unsigned long long*>(smem + o);
    o += (size_t)2 * hpb * 8;
    int32_t* s_cnt12 = reinterpret_cast<int32_t*>(smem + o);
    o += (size_t)2 * hpb * 4;
    int32_t* s_top2 = reinterpret_cast<int32_t*>(smem + o);
    o += (size_t)2 * hpb * 4;
    o = (o + 15) & ~(size_t)15;
    uint2* s_tab2 = reinterpret_cast<uint2*>(smem + o);  // [2][hpb][CW]: the walk through the shared rows per entry column
    o += (size_t)2 * hpb * CW * 8;
    uint4* s_col2 = reinterpret_cast<uint4*>(smem + o);  // [2][hpb][CW]: what every barcode lane of a hit needs of a column
    o += (size_t)2 * hpb * CW * 16;
    uint32_t* s_peq = reinterpret_cast<uint32_t*>(smem + o);
    o += (size_t)16 * N * 4;
    o = (o + 15) & ~(size_t)15;
    o = (o + 31) & ~(size_t)31;
    bb_lb_entry* s_lb = reinterpret_cast<bb_lb_entry*>(smem + o);  // FAST: the bound's table (one entry per byte of Match bits)
    o += FAST ? 256 * sizeof(bb_lb_entry) : 0;
    unsigned long long* s_tail = reinterpret_cast<unsigned long long*>(smem + o);  // [t][lo|hi][thread]
    {
        const uint32_t* gp = reinterpret_cast<const uint32_t*>(tables + groups[g].off_peq_sub[strand]);
        const int words = 16 * N;
        for (int i = threadIdx.x; i < words; i += blockDim.x) s_peq[i] = gp[i];
        if constexpr (FAST)
            for (uint32_t i = threadIdx.x; i < 256u; i += blockDim.x) lodhi_bound_table_entry(i, s_lb[i]);
    }
    const int hl = threadIdx.x / N;
    const int p = threadIdx.x - hl * N;
    const bool in_blk = hl < (int)hpb;
    const int hls = in_blk ? hl : 0;
    // prefetch of the next iteration's records: lane p of a hit fetches piece p (hit record pieces first, then the
    // prefix record).  Groups with fewer barcodes than pieces (2 N >= PIECES) fetch pieces N.. at the start of the
    // iteration instead, unprefetched — they have many hits per block iteration to hide it behind.
    uint4 pre = make_uint4(0u, 0u, 0u, 0u);
    auto piece = [&](uint32_t idx, int pc) -> uint4 {
        return pc < PIECES_H ? reinterpret_cast<const uint4*>(hits + idx)[pc] : reinterpret_cast<const uint4*>(pfxs + idx)[pc - PIECES_H];
    };
    auto prefetch = [&](uint32_t it) {
        const uint32_t li = it * hpb + (uint32_t)hl;
        if (in_blk && p < PIECES && it < n_iter && li < n_list) pre = piece(hit_list ? hit_list[li] : li, p);
    };
    // set `it` -> half h: lane p of a hit stores piece p of its record pair (prefetched in `pre`)
    auto store_set = [&](uint32_t it, uint32_t h) {
        const uint32_t li = it * hpb + (uint32_t)hl;
        if (in_blk && p < PIECES && it < n_iter && li < n_list) {
            uint4* dst = s_hit2 + ((size_t)h * hpb + hl) * PIECES;
            dst[p] = pre;
            if (p + N < PIECES) dst[p + N] = piece(hit_list ? hit_list[li] : li, p + N);
        }
    };
    // per-column table and reduction cells of the set in half h (all lanes)
    auto build_cols = [&](uint32_t h) {
        const uint4* hitb = s_hit2 + (size_t)h * hpb * PIECES;
        uint4* colb = s_col2 + (size_t)h * hpb * CW;
        if (in_blk && p == 0) {
            const uint32_t x = h * hpb + (uint32_t)hl;
            s_max2[x] = 0ull; s_sec2[x] = 0ull; s_cnt12[x] = 0; s_top2[x] = 0x7FFFFFFF; s_maxB2[x] = 0ull; s_secB2[x] = 0ull;
        }
    // Per (hit, column), once for the hit's N barcode lanes: x = byte offset of the column's base-set row in the Peq table,
    // y / z = carry-in of the shared rows (horizontal +1 / -1 of row P) as words of their own.  The lanes then spend one
    // 16-byte LDS read (a broadcast: the lanes of a hit read the same address) and one addition per column instead of
    // three bit-field extractions and a multiply-add (all half rate, profiles/valu_ceiling.json).
    for (uint32_t l = threadIdx.x; l < hpb * (uint32_t)CW; l += blockDim.x) {
        const uint32_t hw = l / (uint32_t)CW, c = l % (uint32_t)CW;
        const uint32_t* rec = reinterpret_cast<const uint32_t*>(hitb + hw * PIECES);
        const uint32_t code = (rec[8 + (c >> 2)] >> (8u * (c & 3u))) & 0xFu;
        const uint32_t* hv = reinterpret_cast<const uint32_t*>(hitb + hw * PIECES + PIECES_H);  // {ph lo, ph hi, mh lo, mh hi}
        const uint32_t hp = (hv[c >> 5] >> (c & 31u)) & 1u, hm = (hv[2 + (c >> 5)] >> (c & 31u)) & 1u;
        colb[l] = make_uint4(code * (uint32_t)N * 4u, hp, hm, 0u);  // the carry-in bits as words of their own: no extraction per lane
    }
    };
    auto build_walks = [&](uint32_t h) {
        const uint4* hitb = s_hit2 + (size_t)h * hpb * PIECES;
        uint2* tabb = s_tab2 + (size_t)h * hpb * CW;
    // The walk of a traced path through the shared rows depends only on the hit and on the column in which the
    // path enters row P, not on the barcode: the first hpb * CW lanes of the block each walk one (hit, entry column)
    // once — 16 columns from independent LDS reads, static register indices — and every barcode lane later looks
    // its entry up instead of walking (the walk was 12 % of this kernel).  Entry: x = text-op planes of the columns
    // cx, cx-1, .. (bit i <-> column cx - i; lo | hi << 16), y = consumed rows (bits 0..15) | text ops (bits 16..20) |
    // bit position of a cursor still alive after the 16 columns (bits 24..27, flag in bit 31: the lane then finishes
    // in a loop).
    {
        const uint32_t pm = (1u << P) - 1u;  // P <= 16
        for (uint32_t l = threadIdx.x; l < hpb * (uint32_t)CW; l += blockDim.x) {
            const uint32_t hw = l / (uint32_t)CW;
            const int32_t cxw = (int32_t)(l % (uint32_t)CW) + 1;
            const uint32_t* shw = reinterpret_cast<const uint32_t*>(hitb + hw * PIECES + SH_PIECE);
            uint32_t bh = 1u, lo2 = 0u, hi2 = 0u, dgw = 0u, n2 = 0u;
#pragma unroll 1
            for (int i = 0; i < 16 && bh != 0u && cxw - i >= 1; ++i) {  // rolled: short, and the registers are wanted elsewhere
                const uint32_t w = shw[cxw - 1 - i];
                const uint32_t Lr = w & 0xFFFFu, Hr = w >> 16;
                const uint32_t Dr = Lr & Hr;
                const uint32_t nb = ((Dr + bh) & ~Dr) & pm;
                const bool has = nb != 0u, lo = (Lr & nb) != 0u, hi = (Hr & nb) != 0u;
                lo2 |= lo ? (1u << i) : 0u;
                hi2 |= hi ? (1u << i) : 0u;
                const bool consume = has & !hi;
                dgw |= consume ? nb : 0u;
                bh = consume ? ((nb << 1) & pm) : nb;
                n2 += has ? 1u : 0u;
            }
            if (cxw - 16 < 1) bh = 0u;
            tabb[l] = make_uint2(lo2 | (hi2 << 16), dgw | (n2 << 16) | (bh ? 0x80000000u | ((uint32_t)(__ffs(bh) - 1) << 24) : 0u));
        }
    }
    };
    prefetch(blockIdx.x);
    store_set(blockIdx.x, 0u);
    prefetch(blockIdx.x + gridDim.x);
    BB_PFX_SYNC();
    build_cols(0u);
    build_walks(0u);
    BB_PFX_SYNC();
    uint32_t half = 0u;
  for (uint32_t it = blockIdx.x; it < n_iter; it += gridDim.x, half ^= 1u) {
    const uint32_t li = it * hpb + (uint32_t)hl;
    const bool exists = in_blk && li < n_list;
    const uint32_t hit_idx = hit_list ? (exists ? hit_list[li] : 0u) : li;
    // the next set's records go to the other half (its last readers finished before the barrier this wave just left)
    store_set(it + gridDim.x, half ^ 1u);
    prefetch(it + 2u * gridDim.x);
    const uint4* s_hit = s_hit2 + (size_t)half * hpb * PIECES;
    const uint4* s_col = s_col2 + (size_t)half * hpb * CW;
    const uint2* s_tab = s_tab2 + (size_t)half * hpb * CW;
    unsigned long long* s_max = s_max2 + half * hpb, *s_sec = s_sec2 + half * hpb, *s_maxB = s_maxB2 + half * hpb, *s_secB = s_secB2 + half * hpb;
    int32_t* s_cnt1 = s_cnt12 + half * hpb, *s_top = s_top2 + half * hpb;
    bb_hit H;  // header only
    {
        const uint4 h0 = s_hit[hls * PIECES], h1 = s_hit[hls * PIECES + 1];
        H.read_idx = h0.x; H.text_start = h0.y; H.text_end = h0.z; H.ws = h0.w;
        H.we = h1.x; H.cost = (int16_t)(h1.y & 0xFFFFu); H.group = (uint8_t)((h1.y >> 16) & 0xFFu); H.strand = (uint8_t)(h1.y >> 24);
        H.valid = (uint8_t)(h1.z & 0xFFu); H.read_len = h1.w;
    }
    bool active = exists && H.valid != 0;
    if (exists && !H.valid && p == 0) rows[hit_idx].row._pad[0] = 0;
    const int32_t wn = active ? (int32_t)(H.we - H.ws) : 0;
    const uint32_t* s_sh = reinterpret_cast<const uint32_t*>(s_hit + hls * PIECES + SH_PIECE);  // sh[64] of the prefix record

    int wmax = wn;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) wmax = max(wmax, __shfl_xor(wmax, d, 64));
    wmax = __builtin_amdgcn_readfirstlane(wmax);

    // ---- forward pass on the lane's own rows (one word), carry-in from the shared rows ----
    uint32_t L0[CW], H0[CW];
    int32_t best_cost = 0x7FFFFFFF, best_pos = -1;
    {
        const uint4* colv = s_col + hls * CW;
        const uint32_t pb4 = (uint32_t)p * 4u;
        const uint8_t* s_peq_b = reinterpret_cast<const uint8_t*>(s_peq);
        uint32_t pv = 0xFFFFFFFFu, mv = 0u;
        // bottom-row deltas (bit 31 of ph / mh), newest column at bit 0: one v_alignbit per column and plane;
        // the column order is restored after the loop
        uint32_t upr[2] = {0u, 0u}, dnr[2] = {0u, 0u};
#pragma unroll
        for (int c0 = 0; c0 < CW; c0 += BB_CG) {
            if (c0 < BB_FIXED_COLS || c0 < wmax) {  // wave-uniform; the first BB_FIXED_COLS columns unconditionally (straight-line code)
#pragma unroll
                for (int c = c0; c < c0 + BB_CG; ++c) {
                    const uint4 cv = colv[c];
                    const uint32_t eq = *reinterpret_cast<const uint32_t*>(s_peq_b + (cv.x + pb4));
                    const uint32_t hp = cv.y, hm = cv.z;
                    // Every boolean step as ONE three-input v_bitop3 (at three waves per SIMD v_bitop3 issues at 941 G/s, v_and / v_or
                    // at 760: profiles/valu_ceiling.json): 10 v_bitop3 + 1 add + 2 v_bfrev + 2 v_lshlrev_b64 per column (was 8 + 5 + 2 + 2)
                    const uint32_t x = bitop3<0xC8>(eq, pv, hm);                       // (eq | hm) & pv
                    const uint32_t t = bitop3<BB_TT_XOR_OR>(x + pv, pv, eq);           // ((x + pv) ^ pv) | eq
                    const uint32_t d0 = bitop3<0xFE>(t, hm, mv);                       // t | hm | mv
                    const uint32_t ph = bitop3<BB_TT_OR_NOR>(mv, d0, pv), mh = bitop3<0xC0>(pv, d0, 0u);   // mv | ~(d0 | pv),  pv & d0
                    const uint32_t l = bitop3<0x15>(d0, eq, ph), hh = bitop3<0x3A>(d0, eq, ph);  // move planes (see move_bits)
                    L0[c] = __brev(l); H0[c] = __brev(hh);  // row P+1 <-> bit 31, row P+32 <-> bit 0
                    // {accumulator : vector} shifted as ONE 64-bit value: the vector's top bit (the bottom row's delta) lands in
                    // the accumulator and the vector is shifted, in one half-rate instruction instead of v_alignbit + v_lshl_or
                    const unsigned long long tp = shl1_64(((unsigned long long)upr[c >> 5] << 32) | ph);
                    const unsigned long long tm = shl1_64(((unsigned long long)dnr[c >> 5] << 32) | mh);
                    upr[c >> 5] = (uint32_t)(tp >> 32); dnr[c >> 5] = (uint32_t)(tm >> 32);
                    // with phs = (ph << 1) | hp and mhs = (mh << 1) | hm (carry-in of the shared rows in bit 0):
                    const uint32_t nph = bitop3<0x01>((uint32_t)tp, hp, d0);           // ~(phs | d0)
                    mv = bitop3<0xA8>((uint32_t)tp, hp, d0);                           // phs & d0
                    pv = bitop3<0xFE>(nph, (uint32_t)tm, hm);                          // mhs | ~(d0 | phs)
                }
            }
        }
        // columns processed (wave-uniform): the groups below wmax; word w holds its columns newest-first
        const int pc = min(CW, ((max(wmax, BB_FIXED_COLS) + BB_CG - 1) / BB_CG) * BB_CG);
        const int n0 = min(pc, 32), n1 = pc - n0;
        uint32_t up[2], dn[2];
        up[0] = n0 ? __brev(upr[0]) >> (32 - n0) : 0u; dn[0] = n0 ? __brev(dnr[0]) >> (32 - n0) : 0u;
        up[1] = n1 ? __brev(upr[1]) >> (32 - n1) : 0u; dn[1] = n1 ? __brev(dnr[1]) >> (32 - n1) : 0u;
        const unsigned long long wmask = wn >= 64 ? ~0ull : ((1ull << wn) - 1ull);
        unsigned long long Pm = (((unsigned long long)up[1] << 32) | up[0]) & wmask;   // horizontal deltas of row P+32
        unsigned long long Mm = (((unsigned long long)dn[1] << 32) | dn[0]) & wmask;
        // The trailing shared rows, row-wise: the same recurrence with the roles of rows and columns exchanged — bit-vectors
        // run along the window's columns, the state is the horizontal deltas of the row above, the carry-in is the vertical
        // delta +1 of column 0 (D[r][0] = r), Eq comes from the prefix record (the rows' characters are the same for every
        // barcode).  ~20 64-bit operations per row and lane instead of a second word in every column step.  The rows' move
        // planes (as column masks) are parked in LDS for the start of the traceback.
        if constexpr (TAIL) {
            const uint2* teq = reinterpret_cast<const uint2*>(s_hit + hls * PIECES + PIECES_H + 1);
#pragma unroll 1
            for (int t = 0; t < T; ++t) {
                const uint2 e2 = teq[t];
                const unsigned long long Eq = ((unsigned long long)e2.y << 32) | e2.x;
                const unsigned long long D0 = (((Eq & Pm) + Pm) ^ Pm) | Eq | Mm;
                const unsigned long long Pvv = Mm | ~(D0 | Pm), Mvv = Pm & D0;
                const unsigned long long Pvs = (Pvv << 1) | 1ull, Mvs = Mvv << 1;
                const unsigned long long Ph = Mvs | ~(D0 | Pvs), Mh = Pvs & D0;
                const unsigned long long isM = D0 & Eq, tl = ~(isM | Ph), th = (Ph & ~isM) | (tl & D0);
                s_tail[(size_t)(2 * t) * blockDim.x + threadIdx.x] = tl;
                s_tail[(size_t)(2 * t + 1) * blockDim.x + threadIdx.x] = th;
                Pm = Ph & wmask; Mm = Mh & wmask;
            }
        }
        pick_minimum(Pm, Mm, wn, m, active, DEFPOL ? BB_LM_PLATEAU_RIGHT : G.pol_lm, DEFPOL ? false : G.pol_tie_last != 0, best_cost, best_pos);
        if constexpr (!FAST) { if (active && best_pos >= 0 && best_cost <= G.k1) atomicAdd(&s_cnt1[hl], 1); }
    }
    bool cand = active && best_pos >= 0 && best_cost <= G.k2;
    // ---- traceback, phase 1: the lane's own rows, one-hot cursor on one 32-bit word (row P+1 <-> bit 31).
    // The cursor leaves the word either by the carry of the Del-run addition (the run continues in the shared
    // rows at the same column, which then has no text op in this phase) or by a Match/Sub out of row P+1 (next
    // column); either way phase 2 starts with the cursor entering row P. ----
    unsigned long long plo = 0ull, phi = 0ull;
    uint32_t b = 0u, dg = 0u;
    // ---- phase 0: the trailing shared rows, from (row m, column best_pos) on their column masks: one step per loop
    // iteration (Match/Sub: row and column, Ins: column, Del: row) until the cursor reaches row P+32 — typically T
    // iterations.  Rows left over when the window's first column is passed are deleted, like everything above them. ----
    int32_t c_ent = best_pos;   // column in which the cursor enters the lane's word
    int32_t tr = cand ? T - 1 : -1;
    uint32_t dgt = 0u;          // trailing rows consumed by a Match/Sub
    while (TAIL && __any(tr >= 0 && c_ent >= 1)) {
        const bool on = tr >= 0 && c_ent >= 1;
        const int rr = on ? tr : 0, sh = on ? c_ent - 1 : 0;
        const unsigned long long l64 = s_tail[(size_t)(2 * rr) * blockDim.x + threadIdx.x], h64 = s_tail[(size_t)(2 * rr + 1) * blockDim.x + threadIdx.x];
        const uint32_t lo = (uint32_t)(l64 >> sh) & 1u, hi = (uint32_t)(h64 >> sh) & 1u;
        const bool del = on && (lo & hi) != 0u, text = on && !del, diag = on && hi == 0u;
        plo |= text ? (unsigned long long)lo << sh : 0ull;
        phi |= text ? (unsigned long long)hi << sh : 0ull;
        dgt |= diag ? 1u << rr : 0u;
        tr -= (del || diag) ? 1 : 0;
        c_ent -= text ? 1 : 0;
    }
    // the cursor enters at row P+32 = bit 0 of the word in column c_ent: that column's bit of this mask is simply
    // added in with the Del-run sum (v_add3)
    const unsigned long long smask = (cand && tr < 0 && c_ent >= 1) ? 1ull << (c_ent - 1) : 0ull;  // column 0: nothing to walk
    const uint32_t sm_w[2] = {(uint32_t)smask, (uint32_t)(smask >> 32)};
    // Mask arithmetic only (profiles/valu_ceiling.json: v_cmp / v_cndmask / shifts issue at half the rate of and/or/add):
    // nb is one-hot or zero, so "the landing cell has lo" is (Lr & nb) != 0 — brought to bit 31 by negation and shifted
    // into the column accumulators with one v_alignbit per plane (word 1: columns 33.., word 0: columns 1..32, newest
    // column at bit 0 = its final place); a Match/Sub step is cm = nb & ~Hr (one-hot or zero): consumed rows |= cm,
    // and the cursor moves by b = nb + cm (nb << 1 when it consumed, nb when it did not).
    uint32_t pl_acc[2] = {0u, 0u}, ph_acc[2] = {0u, 0u};
#pragma unroll
    for (int c0 = CW; c0 >= BB_CG; c0 -= BB_CG) {
        if (c0 <= BB_FIXED_COLS || c0 - (BB_CG - 1) <= wmax) {  // wave-uniform
#pragma unroll
            for (int c = c0; c > c0 - BB_CG; --c) {
                const uint32_t Lr = L0[c - 1], Hr = H0[c - 1];
                const uint32_t Dr = Lr & Hr;
                const uint32_t nb = bitop3<0x0C>(Dr, Dr + b + ((sm_w[(c - 1) >> 5] >> ((c - 1) & 31)) & 1u), 0u);  // ~Dr & sum
                const uint32_t tl = Lr & nb, th = Hr & nb;
                const uint32_t cm = bitop3<0x0C>(Hr, nb, 0u);  // ~Hr & nb
                pl_acc[(c - 1) >> 5] = __builtin_amdgcn_alignbit(pl_acc[(c - 1) >> 5], 0u - tl, 31);
                ph_acc[(c - 1) >> 5] = __builtin_amdgcn_alignbit(ph_acc[(c - 1) >> 5], 0u - th, 31);
                dg |= cm;
                b = nb + cm;
            }
        }
    }
    plo |= ((unsigned long long)pl_acc[1] << 32) | pl_acc[0];
    phi |= ((unsigned long long)ph_acc[1] << 32) | ph_acc[0];
    // Text ops of phase 1 = rows it consumed + its Ins columns.  Whichever way the cursor left the word, the
    // columns best_pos .. cx+1 carry exactly those ops: phase 2 starts at column cx = best_pos - ntext.
    int32_t ntext = cand ? __popc(dg) + __popc(dgt) + __popcll(phi & ~plo) : 0;
    const int32_t cx = cand ? best_pos - ntext : 0;
    // ---- phase 2: the shared rows (row r <-> bit P - r): looked up in the block's walk table; a cursor still
    // alive after the table's 16 columns (more than 16 - P insertions inside the shared rows) finishes in the
    // loop underneath on the move bits of the hit's prefix record. ----
    uint32_t dgh = 0u;
    BB_PFX_SYNC();  // barrier A: every wave is done with the previous set; the next set's records are in place
    {
        const uint32_t pm = (1u << P) - 1u;
        const uint2 e = (cand && cx >= 1) ? s_tab[hls * CW + cx - 1] : make_uint2(0u, 0u);
        const uint32_t lo2 = e.x & 0xFFFFu, hi2 = e.x >> 16;
        dgh = e.y & 0xFFFFu;
        uint32_t bh = (e.y >> 31) ? 1u << ((e.y >> 24) & 0xFu) : 0u;
        ntext += (int32_t)((e.y >> 16) & 0x1Fu);
        // local bit i <-> column cx - i <-> plane bit cx - i - 1: reverse the 16 bits and slide them under cx
        const unsigned long long rl = (unsigned long long)(__brev(lo2) >> 16), rh = (unsigned long long)(__brev(hi2) >> 16);
        plo |= cx >= 16 ? (rl << (cx - 16)) : (rl >> (16 - cx));
        phi |= cx >= 16 ? (rh << (cx - 16)) : (rh >> (16 - cx));
        int32_t col = cx - 16;
        if (col < 1) bh = 0u;
        while (__any(bh != 0u)) {  // rare: more than 16 columns inside the shared rows
            const uint32_t w = s_sh[col >= 1 ? col - 1 : 0];
            const uint32_t Lr = w & 0xFFFFu, Hr = w >> 16;
            const uint32_t Dr = Lr & Hr;
            const uint32_t nb = bh ? (((Dr + bh) & ~Dr) & pm) : 0u;
            const bool has = nb != 0u, lo = (Lr & nb) != 0u, hi = (Hr & nb) != 0u;
            const unsigned long long bit = 1ull << (col >= 1 ? col - 1 : 0);
            plo |= lo ? bit : 0ull;
            phi |= hi ? bit : 0ull;
            const bool consume = has & !hi;
            dgh |= consume ? nb : 0u;
            bh = consume ? ((nb << 1) & pm) : nb;
            ntext += has ? 1 : 0;
            col -= has ? 1 : 0;
            if (col < 1) bh = 0u;
        }
    }
    const int32_t tstart = cand ? best_pos - ntext : 0;
    // consumed rows in natural order (row r <-> bit r-1)
    const unsigned long long diagrow = ((unsigned long long)__brev(dg) << P) | (P ? (unsigned long long)(__brev(dgh) >> (32 - P)) : 0ull) |
                                       ((unsigned long long)dgt << (P + 32));
    const unsigned long long delrow = cand ? (low64(m) & ~diagrow) : 0ull;
    if constexpr (FAST) {
        // A bound for every lane; the exact score of the best-bounded lane only, later (k_rows).  Per hit the two highest
        // bounds of BOTH candidate sets of searcher.rs:303-328 — pass 1: lowest cost <= k1, pass 2: every lane with a local
        // minimum — are collected with one pair of returning LDS atomics per set and lane (key = bound bits : 0xFFFF - p, so
        // the maximum is also the FIRST maximum; whatever a lane's atomicMax displaces or fails to displace, min(old, key),
        // is a candidate for second place, and the true second always shows up as one).  Which set counts is known after
        // the single barrier: pass 2 iff pass 1 has fewer than two members, i.e. its second place is empty.
        const float ubf = lodhi_bound_tab<CW>(cand ? plo : 0ull, cand ? phi : 0ull, cand ? tstart : 0, cand ? best_pos : 0, wmax, s_lb);
        (void)delrow;
        const unsigned long long key = ((unsigned long long)__float_as_uint(ubf) << 16) | (unsigned long long)(0xFFFFu - (uint32_t)p);
        if (N >= 64) {
            // A wave holds lanes of at most two hits.  96 lanes posting to one LDS address serialise inside the LDS unit (and
            // hold up the other waves' table reads): the wave finds its own top-2 per (hit, candidate set) on the VALU first
            // and four lanes post them.
            const uint32_t lane = threadIdx.x & 63u;
            const int hA = __builtin_amdgcn_readfirstlane(hl);
            const bool c2 = cand, c1 = cand && best_cost <= G.k1;
            const uint32_t vb = __float_as_uint(ubf);
            unsigned long long t[4], u[4];  // combos: 0 = (hit A, pass 1), 1 = (A, pass 2), 2 = (B, pass 1), 3 = (B, pass 2)
            // the two candidate sets differ only if some candidate costs more than k1, and two waves in three hold one hit:
            // usually one reduction serves all
            const bool sets_differ = __any(c2 && !c1), two_hits = __any(hl != hA);
            wave_top2(c2 && hl == hA, vb, key, t[1], u[1]);
            if (sets_differ) wave_top2(c1 && hl == hA, vb, key, t[0], u[0]);
            else { t[0] = t[1]; u[0] = u[1]; }
            t[2] = t[3] = u[2] = u[3] = 0ull;
            if (two_hits) {
                wave_top2(c2 && hl != hA, vb, key, t[3], u[3]);
                if (sets_differ) wave_top2(c1 && hl != hA, vb, key, t[2], u[2]);
                else { t[2] = t[3]; u[2] = u[3]; }
            }
            if (lane < 4u) {
                const unsigned long long kt = lane == 0u ? t[0] : lane == 1u ? t[1] : lane == 2u ? t[2] : t[3];
                const unsigned long long ku = lane == 0u ? u[0] : lane == 1u ? u[1] : lane == 2u ? u[2] : u[3];
                if (kt != 0ull) {
                    const int hx = hA + (int)(lane >> 1);
                    unsigned long long* pm = (lane & 1u) ? &s_maxB[hx] : &s_max[hx];
                    unsigned long long* ps = (lane & 1u) ? &s_secB[hx] : &s_sec[hx];
                    const unsigned long long o = atomicMax(pm, kt);
                    atomicMax(ps, o < kt ? o : kt);
                    if (ku != 0ull) atomicMax(ps, ku);
                }
            }
        } else if (cand) {
            const unsigned long long o2 = atomicMax(&s_maxB[hl], key);
            atomicMax(&s_secB[hl], o2 < key ? o2 : key);
            if (best_cost <= G.k1) {
                const unsigned long long o1 = atomicMax(&s_max[hl], key);
                atomicMax(&s_sec[hl], o1 < key ? o1 : key);
            }
        }
        build_cols(half ^ 1u);
        BB_PFX_SYNC();  // barrier B: the set's candidates are posted, the next set's column table is complete
        if (active) {
            const bool pass2 = s_sec[hl] == 0ull && G.k1 < G.k2;
            const unsigned long long mx = pass2 ? s_maxB[hl] : s_max[hl], sx = pass2 ? s_secB[hl] : s_sec[hl];
            if (mx != 0ull) {
                if ((uint32_t)p == 0xFFFFu - (uint32_t)(mx & 0xFFFFull)) {
                    bb_winrec W;
                    W.plo = plo; W.phi = phi; W.diagrow = diagrow;
                    W.ub_second = sx ? (double)__uint_as_float((uint32_t)(sx >> 16)) / G.perfect : -1.0;   // -1: no other candidate
                    W.tstart = (uint8_t)tstart; W.best_pos = (uint8_t)best_pos; W.top = (uint16_t)p;
                    W.flags = 0; W.marker = 2; W._pad[0] = W._pad[1] = 0;
#pragma unroll
                    for (int q = 0; q < 8; ++q) W._pad0[q] = 0;
                    *reinterpret_cast<bb_winrec*>(rows + hit_idx) = W;
                }
            } else if (p == 0) {  // no candidate at all: flank-only row (searcher.rs:353-362)
                bb_rowtmp R;
                bb_row& r = R.row;
                r.read_idx = H.read_idx; r.read_len = H.read_len;
                r.rel_dist_to_end = rel_dist_to_end((int64_t)H.text_start, (int64_t)H.read_len);
                r.read_start_flank = H.text_start; r.read_end_flank = H.text_end;
                r.flank_cost = H.cost; r.group_idx = H.group; r.strand = H.strand;
                r._pad[0] = 1; r._pad[1] = r._pad[2] = 0;
                r.read_start_bar = H.text_start; r.read_end_bar = H.text_end;
                r.bar_start = 0; r.bar_end = 0;
                r.match_type = (uint8_t)(G.type == BB_FTAG ? BB_FFLANK : BB_RFLANK);
                r.barcode_cost = (int16_t)G.m_bar; r.barcode_idx = -1;
                rows[hit_idx] = R;
            }
        }
        build_walks(half ^ 1u);  // read after the next barrier A; built while the slower waves finish this set
    } else {
        build_cols(half ^ 1u);
        build_walks(half ^ 1u);
        double s_norm = -1.0;
        {
            const bool on = cand;  // the loop is wave-uniform: idle lanes walk it with empty masks
            const double sc = (uint32_t)G.pol_lodhi_exp == (uint32_t)BB_LODHI_EXP_DEFAULT
                                  ? lodhi_replay<CW>(on ? plo : 0ull, on ? phi : 0ull, on ? delrow : 0ull, on ? tstart : 0, on ? best_pos : 0, wmax)
                                  : lodhi_replay<CW, true>(on ? plo : 0ull, on ? phi : 0ull, on ? delrow : 0ull, on ? tstart : 0, on ? best_pos : 0, wmax, (uint32_t)G.pol_lodhi_exp);
            if (cand) s_norm = G.perfect > 0.0 ? sc / G.perfect : 0.0;
        }
        pick_and_emit(active, cand, best_cost, s_norm, p, hl, H, hit_idx, G, plo, phi, diagrow, tstart, best_pos, s_cnt1, s_max, s_sec, s_top,
                      min_score, min_score_diff, rows);
        BB_PFX_SYNC();
    }
  }
}

// k_rows: one lane per flank hit whose row slot holds a bb_winrec (marker 2).  Scores the recorded path exactly
// (lodhi_replay: the oracle's f64 recurrence) and decides with the runner-up's BOUND:
//   * top - bound(second) >= min_score_diff (with a margin far above f64 rounding): no other barcode can reach the top's
//     score or come within min_score_diff of it, so the recorded barcode is the first maximum and the difference test of
//     searcher.rs:393-395 passes whatever the others' exact scores are -> tag row if top >= min_score, else flank-only row;
//   * top < min_score and bound(second) < min_score: no barcode reaches min_score -> flank-only row;
//   * otherwise the bounds do not decide: the hit goes to the exact kernel (all barcodes scored exactly) through the
//     fallback list of its (group, strand).
// The decision of k_rows for one hit, as a function: k_barcode_lane's final trip calls it on the record it would otherwise have
// written (no winrec round trip, no k_rows launch for its hits).  Wave-wide: lanes without a record pass mine = false.
__device__ __forceinline__ void rows_decide(bool mine, const bb_winrec& W, const uint4 h0, const uint4 h1, uint32_t t, int wmax,
                                            const bb_group_dev* __restrict__ groups, bb_rowtmp* __restrict__ rows, double min_score,
                                            double min_score_diff, double margin, uint32_t* __restrict__ fb_lists, uint32_t list_stride,
                                            uint32_t* __restrict__ fb_cnt) {
    const uint32_t grp = (h1.y >> 16) & 0xFFu, strand = (h1.y >> 24) & 1u;
    const bb_group_dev& G = groups[mine ? grp : 0u];
    const int m = G.m_bar;
    const unsigned long long delrow = mine ? (low64(m) & ~W.diagrow) : 0ull;
    const uint32_t expk = (uint32_t)groups[0].pol_lodhi_exp;  // the context's policy: the same in every group
    const double sc = expk == (uint32_t)BB_LODHI_EXP_DEFAULT
                          ? lodhi_replay<64>(mine ? W.plo : 0ull, mine ? W.phi : 0ull, delrow, mine ? (int32_t)W.tstart : 0, mine ? (int32_t)W.best_pos : 0, wmax)
                          : lodhi_replay<64, true>(mine ? W.plo : 0ull, mine ? W.phi : 0ull, delrow, mine ? (int32_t)W.tstart : 0, mine ? (int32_t)W.best_pos : 0, wmax, expk);
    if (!mine) return;
    const double s_norm = G.perfect > 0.0 ? sc / G.perfect : 0.0;
    const bool clear = W.ub_second < 0.0 || (s_norm - W.ub_second) >= min_score_diff + margin;
    const bool none = s_norm < min_score && W.ub_second < min_score - margin;
    if (!clear && !none) {  // the bounds do not decide this hit
        const uint32_t slot = 4u * grp + ((h1.x - h0.w) > 48u ? 2u : 0u) + strand;  // {we - ws}: the window class of k_hit_lists
        const uint32_t at = atomicAdd(&fb_cnt[slot], 1u);
        fb_lists[(size_t)slot * list_stride + at] = t;
        rows[t].row._pad[0] = 0;  // no row yet (and no stale record in the slot): the exact kernel writes it
        return;
    }
    const bool valid = clear && s_norm >= min_score;
    bb_rowtmp R;
    bb_row& r = R.row;
    const uint32_t read_len = h1.w, text_start = h0.y, text_end = h0.z, ws = h0.w;
    r.read_idx = h0.x; r.read_len = read_len;
    r.rel_dist_to_end = rel_dist_to_end((int64_t)text_start, (int64_t)read_len);
    r.read_start_flank = text_start; r.read_end_flank = text_end;
    r.flank_cost = (int16_t)(h1.y & 0xFFFFu); r.group_idx = (uint8_t)grp; r.strand = (uint8_t)strand;
    r._pad[0] = 1; r._pad[1] = r._pad[2] = 0;
    if (valid) {
        int32_t txt_lo, txt_hi, bcost;
        subpath_closed_form(W.plo, W.phi, W.diagrow, (int)W.tstart, (int)W.best_pos, m, G.rel_lo, G.rel_hi, txt_lo, txt_hi, bcost);
        r.read_start_bar = ws + (uint32_t)txt_lo; r.read_end_bar = ws + (uint32_t)txt_hi;
        r.bar_start = ws + (uint32_t)G.rel_lo; r.bar_end = ws + (uint32_t)G.rel_hi;
        r.match_type = (uint8_t)G.type; r.barcode_cost = (int16_t)bcost; r.barcode_idx = (int16_t)W.top;
    } else {
        r.read_start_bar = text_start; r.read_end_bar = text_end;
        r.bar_start = 0; r.bar_end = 0;
        r.match_type = (uint8_t)(G.type == BB_FTAG ? BB_FFLANK : BB_RFLANK);
        r.barcode_cost = (int16_t)G.m_bar; r.barcode_idx = -1;
    }
    rows[t] = R;
}

__global__ __launch_bounds__(256) void k_rows(const bb_group_dev* __restrict__ groups, const bb_hit* __restrict__ hits, uint32_t n_hits,
                                              bb_rowtmp* __restrict__ rows, double min_score, double min_score_diff, double margin,
                                              uint32_t* __restrict__ fb_lists, uint32_t list_stride, uint32_t* __restrict__ fb_cnt) {
    const uint32_t t = blockIdx.x * 256u + threadIdx.x;
    const bool in = t < n_hits;
    bb_winrec W;
    if (in) W = *reinterpret_cast<const bb_winrec*>(rows + t);
    const bool mine = in && W.marker == 2;
    int wmax = mine ? (int)W.best_pos : 0;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) wmax = max(wmax, __shfl_xor(wmax, d, 64));
    wmax = __builtin_amdgcn_readfirstlane(wmax);
    if (!__any(mine)) return;
    const uint4 h0 = mine ? reinterpret_cast<const uint4*>(hits + t)[0] : make_uint4(0u, 0u, 0u, 0u);
    const uint4 h1 = mine ? reinterpret_cast<const uint4*>(hits + t)[1] : make_uint4(0u, 0u, 0u, 0u);
    rows_decide(mine, W, h0, h1, t, wmax, groups, rows, min_score, min_score_diff, margin, fb_lists, list_stride, fb_cnt);
}

// Hit lists for the barcode kernels: slot 4g + 2w + s holds the hits of group g on strand s (the row split of a group —
// bb_group_dev::pfx / tail — differs per strand, and every launch is uniform in it) whose barcode window is at most 48
// columns wide (w = 0) or wider (w = 1): the kernels keep the move bits of every column in registers, and the 48-column
// instantiation runs at 3 waves per SIMD where the 64-column one has room for 2 — with large flank error budgets the
// WIDEST possible window exceeds 48 columns while nearly every actual window does not.  Hits whose
// get_matching_region was None (searcher.rs:445-449) are skipped here and marked row-less.  One atomic per
// (block, slot): ballots + LDS.
__global__ __launch_bounds__(256) void k_hit_lists(const bb_hit* __restrict__ hits, uint32_t n_hits, bb_rowtmp* __restrict__ rows,
                                                   uint32_t* __restrict__ lists, uint32_t list_stride, uint32_t* __restrict__ list_cnt,
                                                   uint32_t n_groups, const bb_group_dev* __restrict__ groups) {
    const uint32_t t = blockIdx.x * 256u + threadIdx.x;
    const bool in = t < n_hits;
    uint32_t grp = 0, strand = 0, vld = 0, wide = 0;
    if (in) {  // second 16-byte piece of the record: {we, cost|group|strand, valid, read_len}; ws is the last word of the first
        const uint4 h1 = reinterpret_cast<const uint4*>(hits + t)[1];
        const uint32_t ws = reinterpret_cast<const uint32_t*>(hits + t)[3];
        grp = (h1.y >> 16) & 0xFFu; strand = h1.y >> 24; vld = h1.z & 0xFFu; wide = (h1.x - ws) > 48u ? 1u : 0u;
    }
    const bool valid = in && vld;
    if (in && !valid) rows[t].row._pad[0] = 0;
    const unsigned lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    const uint32_t my_slot = valid ? 4u * grp + 2u * wide + (strand & 1u) : 0xFFFFFFFFu;
    // one atomic per (block, slot): the four waves' counts meet in LDS
    __shared__ uint32_t s_cnt[4][4 * BB_MAX_GROUPS], s_base[4 * BB_MAX_GROUPS];
    const uint32_t n_slots = 4u * n_groups;
    unsigned long long my_mask = 0ull;
    for (uint32_t slot = 0; slot < n_slots; ++slot) {
        const unsigned long long mask = __ballot(my_slot == slot);
        if (lane == 0) s_cnt[wv][slot] = (uint32_t)__popcll(mask);
        if (my_slot == slot) my_mask = mask;
    }
    __syncthreads();
    if (threadIdx.x < n_slots) {
        const uint32_t tot = s_cnt[0][threadIdx.x] + s_cnt[1][threadIdx.x] + s_cnt[2][threadIdx.x] + s_cnt[3][threadIdx.x];
        s_base[threadIdx.x] = tot ? atomicAdd(&list_cnt[threadIdx.x], tot) : 0u;
    }
    __syncthreads();
    if (valid) {
        uint32_t base = s_base[my_slot];
        for (unsigned w = 0; w < wv; ++w) base += s_cnt[w][my_slot];
        lists[(size_t)my_slot * list_stride + base + (uint32_t)__popcll(my_mask & ((1ull << lane) - 1ull))] = t;
    }
}

// ------------------------------------------------------------------------------------------------
// k_collapse: one lane per read; rows of the read are rows[b0..b1) in reference order
// (group, forward hits, rc hits).  collapse_overlapping_matches(.., 0.8) in place (interval.rs:4-79).
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ bool rows_overlap(const bb_row& a, const bb_row& b, float thr) {  // interval.rs:30-42
    const uint32_t start = max(a.read_start_flank, b.read_start_flank);
    const uint32_t end = min(a.read_end_flank, b.read_end_flank);
    if (end <= start) return false;
    const uint32_t overlap = end - start;
    const uint32_t min_len = min(a.read_end_flank - a.read_start_flank, b.read_end_flank - b.read_start_flank);
    return ((float)overlap / (float)min_len) >= thr;
}
__device__ __forceinline__ int rows_cmp(const bb_row& a, const bb_row& b) {  // interval.rs:48-76
    const int pa = (a.match_type == BB_FTAG || a.match_type == BB_RTAG) ? 1 : 2;
    const int pb = (b.match_type == BB_FTAG || b.match_type == BB_RTAG) ? 1 : 2;
    if (pa != pb) return pa < pb ? -1 : 1;
    if (pa == 1) {
        if (a.barcode_cost != b.barcode_cost) return a.barcode_cost < b.barcode_cost ? -1 : 1;
        if (a.flank_cost != b.flank_cost) return a.flank_cost < b.flank_cost ? -1 : 1;
        return 0;
    }
    const uint32_t la = a.read_end_flank - a.read_start_flank, lb = b.read_end_flank - b.read_start_flank;
    if (la != lb) return la > lb ? -1 : 1;
    return 0;
}
__global__ __launch_bounds__(256) void k_collapse(bb_rowtmp* __restrict__ rows, const uint32_t* __restrict__ slot_base,
                                                  uint32_t n_reads, uint32_t n_groups, uint32_t* __restrict__ nrows) {
    const uint32_t read = blockIdx.x * 256u + threadIdx.x;
    if (read >= n_reads) return;
    const uint32_t b0 = slot_base[(uint64_t)read * n_groups * 2], b1 = slot_base[(uint64_t)(read + 1) * n_groups * 2];
    if (b0 == b1) { nrows[read] = 0; return; }
    bb_rowtmp* R = rows + b0;
    int n = 0;
    for (uint32_t i = 0; i < b1 - b0; ++i)  // drop hits without a row, keep order
        if (R[i].row._pad[0]) { if ((int)i != n) R[n].row = R[i].row; ++n; }  // (a read's rows are only written when they move)
    for (int i = 1; i < n; ++i) {  // stable insertion sort by read_start_flank (interval.rs:12)
        const bb_row x = R[i].row;
        int j = i - 1;
        while (j >= 0 && R[j].row.read_start_flank > x.read_start_flank) { R[j + 1].row = R[j].row; --j; }
        if (j + 1 != i) R[j + 1].row = x;
    }
    int out = 0, gs = 0;
    for (int i = 1; i <= n; ++i) {
        bool joins = false;
        if (i < n) {
            const bb_row cur = R[i].row;
            for (int q = gs; q < i && !joins; ++q) joins = rows_overlap(R[q].row, cur, 0.8f);
        }
        if (!joins) {
            int best = gs;
            for (int q = gs + 1; q < i; ++q)
                if (rows_cmp(R[q].row, R[best].row) < 0) best = q;
            if (best != out) { const bb_row b = R[best].row; R[out].row = b; }
            ++out;
            gs = i;
        }
    }
    nrows[read] = (uint32_t)out;
}

__global__ __launch_bounds__(256) void k_emit(const bb_rowtmp* __restrict__ rows, const uint32_t* __restrict__ slot_base,
                                              const uint32_t* __restrict__ row_off, uint32_t n_reads, uint32_t n_groups,
                                              const bb_group_dev* __restrict__ groups, bb_row* __restrict__ out,
                                              unsigned long long* __restrict__ counts, uint32_t counts_len) {
    extern __shared__ uint32_t s_hist[];  // per-block histogram, flushed with one global atomic per non-empty bin
    for (uint32_t i = threadIdx.x; i < counts_len; i += 256u) s_hist[i] = 0u;
    __syncthreads();
    const uint32_t read = blockIdx.x * 256u + threadIdx.x;
    if (read < n_reads) {
        const uint32_t b0 = slot_base[(uint64_t)read * n_groups * 2];
        const uint32_t r0 = row_off[read], r1 = row_off[read + 1];
        for (uint32_t i = 0; i < r1 - r0; ++i) {
            const uint4* src = reinterpret_cast<const uint4*>(rows + b0 + i);
            uint4 a = src[0], b = src[1], c = src[2];
            const uint32_t group_idx = (c.z >> 16) & 0xFFu;           // bb_row bytes 42..43: barcode_idx(40..41), group_idx(42), match_type(43)
            const int32_t barcode_idx = (int32_t)(int16_t)(c.z & 0xFFFFu);
            c.w &= 0xFFFF00FFu;                                       // clear the pipeline's row flag (_pad[0], byte 45)
            uint4* dst = reinterpret_cast<uint4*>(out + r0 + i);
            dst[0] = a; dst[1] = b; dst[2] = c;
            const bb_group_dev& G = groups[group_idx];
            atomicAdd(&s_hist[G.count_off + (barcode_idx >= 0 ? barcode_idx : G.n_seqs)], 1u);
        }
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < counts_len; i += 256u) {
        const uint32_t v = s_hist[i];
        if (v) atomicAdd(&counts[i], (unsigned long long)v);
    }
}

// ------------------------------------------------------------------------------------------------
// synthetic reads on the device: one lane per read
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_synth(bb_synth_params P, const uint8_t* __restrict__ table, uint64_t first_read,
                                               uint32_t n, const uint64_t* __restrict__ offsets, uint8_t* __restrict__ bases) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const uint64_t o = offsets[i];
    bb_synth_fill(P, table, first_read + i, bases + o, (uint32_t)(offsets[i + 1] - o));
}

// ------------------------------------------------------------------------------------------------
// k_filter — SURVEY §8(f-1): the reference's filter step (match_pattern pattern.rs:205-240,
// check_filter_pass filter.rs:183-214) on the rows of a batch.  One lane per row; the first row of
// every read walks the read's rows against every pattern (element e <-> row e), keeps the longest
// matching pattern (first among equals) and writes one verdict per row.
// ------------------------------------------------------------------------------------------------
struct bb_pat_elem_dev {
    uint8_t match_type; int8_t orientation; uint8_t relative_to; uint8_t n_cuts;
    int32_t placeholder;
    int64_t lo, hi;
    uint32_t label_off;  // byte offset into the label_ok blob, 0xFFFFFFFF = any label
    bb_cut cuts[BB_MAX_CUTS];
};
struct bb_pat_dev { uint32_t first, n; };

__global__ __launch_bounds__(256) void k_filter(const bb_row* __restrict__ rows, uint64_t n_rows, const bb_group_dev* __restrict__ groups,
                                                const bb_pat_dev* __restrict__ pats, uint32_t n_pats,
                                                const bb_pat_elem_dev* __restrict__ elems, const uint8_t* __restrict__ label_ok,
                                                const uint32_t* __restrict__ label_ids, bb_row_verdict* __restrict__ out) {
    const uint64_t t = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    if (t >= n_rows) return;
    const uint32_t read = rows[t].read_idx;
    if (t > 0 && rows[t - 1].read_idx == read) return;  // not the first row of its read
    uint64_t j = t + 1;
    while (j < n_rows && rows[j].read_idx == read) ++j;
    const uint32_t n = (uint32_t)(j - t);
    uint32_t max_matches = 0, best = 0xFFFFFFFFu;
    for (uint32_t p = 0; p < n_pats; ++p) {
        const bb_pat_dev P = pats[p];
        if (n < P.n || P.n <= max_matches) continue;  // a shorter-or-equal pattern can not replace the current best
        int32_t ph_key[16]; uint32_t ph_label[16]; int n_ph = 0;
        int64_t prev_end = 0; bool have_prev = false, ok = true;
        for (uint32_t e = 0; e < P.n && ok; ++e) {
            const bb_pat_elem_dev el = elems[P.first + e];
            const bb_row m = rows[t + e];
            const bb_group_dev& G = groups[m.group_idx];
            const uint32_t slot = (uint32_t)G.count_off + (m.barcode_idx >= 0 ? (uint32_t)m.barcode_idx : (uint32_t)G.n_seqs);
            if (m.match_type != el.match_type) { ok = false; break; }
            if ((m.match_type == BB_FTAG || m.match_type == BB_RTAG) && el.label_off != 0xFFFFFFFFu && !label_ok[el.label_off + slot]) { ok = false; break; }
            if (el.placeholder >= 0) {
                int found = -1;
                for (int q = 0; q < n_ph; ++q) if (ph_key[q] == el.placeholder) found = q;
                if (found >= 0) { if (ph_label[found] != label_ids[slot]) { ok = false; break; } }
                else if (n_ph < 16) { ph_key[n_ph] = el.placeholder; ph_label[n_ph] = label_ids[slot]; ++n_ph; }
            }
            if (el.orientation >= 0 && el.orientation != (int8_t)m.strand) { ok = false; break; }
            const int64_t ms = m.read_start_bar, me = m.read_end_bar, sl = m.read_len;
            if (el.relative_to == BB_REL_LEFT) ok = !(ms < el.lo || ms > el.hi);
            else if (el.relative_to == BB_REL_RIGHT) ok = !(me < sl - el.hi || me > sl - el.lo);
            else if (el.relative_to == BB_REL_PREV_LEFT) ok = !(have_prev && (ms < prev_end + el.lo || ms > prev_end + el.hi));
            prev_end = me; have_prev = true;
        }
        if (ok) { max_matches = P.n; best = p; }
    }
    for (uint32_t r = 0; r < n; ++r) {
        bb_row_verdict v;
        v.pass = max_matches == n; v.n_cuts = 0; v.match_idx = (uint16_t)r;
#pragma unroll
        for (int q = 0; q < BB_MAX_CUTS; ++q) { v.cuts[q].direction = 0; v.cuts[q]._pad = 0; v.cuts[q].group_id = 0; }
        if (best != 0xFFFFFFFFu && r < pats[best].n) {
            const bb_pat_elem_dev el = elems[pats[best].first + r];
            v.n_cuts = el.n_cuts;
#pragma unroll
            for (int q = 0; q < BB_MAX_CUTS; ++q) if (q < el.n_cuts) v.cuts[q] = el.cuts[q];
        }
        out[t + r] = v;
    }
}

// ------------------------------------------------------------------------------------------------
// k_inspect — SURVEY §8(f-4): get_group_structure (inspect.rs:15-117), one lane per row.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t bb_bucket(uint32_t pos, uint32_t bs) { return ((pos ? pos - 1u : 0u) / bs) * bs; }
__global__ __launch_bounds__(256) void k_inspect(const bb_row* __restrict__ rows, const bb_row_verdict* __restrict__ ver, uint64_t n_rows,
                                                 uint32_t bs, bb_inspect_elem* __restrict__ out) {
    const uint64_t t = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    if (t >= n_rows) return;
    const bb_row a = rows[t];
    const bool first = t == 0 || rows[t - 1].read_idx != a.read_idx;
    const uint32_t start = a.read_start_bar, end = a.read_end_bar, len = a.read_len;
    const uint32_t d_right = len > end ? len - end : 0u, d_right_s = len > start ? len - start : 0u;
    bb_inspect_elem e;
    e.match_type = a.match_type; e.strand = a.strand; e.has_cut = ver ? (ver[t].n_cuts > 0) : 0; e.first = first;
    bool right = !first ? false : !(a.rel_dist_to_end > 0);
    if (!first) {
        const uint32_t pe = rows[t - 1].read_end_bar, d_prev = start > pe ? start - pe : 0u;
        if (d_prev <= d_right) { e.tag = BB_REL_PREV_LEFT; e.lo = bb_bucket(d_prev, bs); e.hi = e.lo + bs; }
        else right = true;
    } else if (!right) { e.tag = BB_REL_LEFT; e.lo = bb_bucket(start, bs); e.hi = e.lo + bs; }
    if (right) { e.tag = BB_REL_RIGHT; e.lo = bb_bucket(d_right, bs); e.hi = bb_bucket(d_right_s, bs) + bs; }
    out[t] = e;
}
