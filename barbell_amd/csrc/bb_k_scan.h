// bb_k_scan.h — the flank scan (sassy's search of the N-masked flank over the whole read, both strands, searcher.rs:438):
// k_flank_scan2 (full height), k_flank_filter + k_flank_verify (the filtered scan), and the exclusive scans of the hit counts.
#pragma once
#include "bb_myers.h"

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

#ifndef BB_VERIFY_FAST
#define BB_VERIFY_FAST 1   // groups of four columns without score / local-minimum tracking while every lane's score is out of reach of k (below)
#endif
#ifndef BB_VERIFY_CHUNKS
#define BB_VERIFY_CHUNKS 5   // 16-byte text loads per lane and round in k_flank_verify.  A verified interval is m + k columns of lead-in plus a flagged piece and
                             // its margins (~75 columns for SQK-NBD114-96): five chunks take it in ONE round, so its one or two 128-byte lines are fetched once
                             // (they do not survive in L2 between a lane's rounds: 58 MB of lines are in flight).  Measured, scan stage / kernel's HBM bytes per
                             // 2 M-read step: 2 chunks 4.75-4.80 ms / 2.40 GB, 3: 4.70-4.79 / 2.0, 4: 4.88-4.96 / 1.95, 5: 4.75-4.76 / 1.5
#endif
#ifndef BB_VERIFY_CHUNKS_WIDE
#define BB_VERIFY_CHUNKS_WIDE 2   // the same for flanks of three words and more (67..256 nt).  Measured on the custom dual-end set (76 / 67 nt, k = 5, 31-row filter
                                  // windows that flag almost nothing): scan stage 14.1-14.2 ms with 2 chunks, 14.7 with 5, 14.8 with 7 (an interval is ~110 columns)
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
                                            bb_hit_raw* stage = nullptr, uint32_t* stage_fill = nullptr,
                                            // a SEGMENT of a read (flank_scan_lane<.., SEG>): where the count goes, whether the lane reports yet (armed) and
                                            // whether the overhang positions are its own (the read's last segment) or only the valley it is in
                                            uint32_t* cnt_cell = nullptr, bool seg = false, bool armed = true, bool last_seg = true) {
    const uint32_t lane = threadIdx.x & 63u;
    const int TB = (m - 1) & 31;
    const bool lm_left = pol_lm == BB_LM_PLATEAU_LEFT, lm_strict = pol_lm == BB_LM_STRICT;
    const int32_t kk_true = kk;
    int32_t kkl = (seg && !armed) ? -1 : kk;   // the budget the local-minimum machine sees: -1 while the lane must not report
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
            { const int32_t kk = kkl; BB_LM_STEP_BUF(st, d + ovh[o], idx); }
            if (seg && d + ovh[o] > kk_true) {   // between valleys: the last segment starts to report, an earlier one that followed its valley here is done
                if (last_seg) kkl = kk_true; else kkl = -1;
            }
        }
        const int32_t kk = kkl;
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
        if (cnt_cell) *cnt_cell = st.nrep; else cnt[((uint64_t)read * n_groups + g) * 2 + STRAND] = st.nrep;
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
// SEG: the lane takes ONE SEGMENT of a read (bb_len.h: vtab; reads of more than split_above lines are cut into segments of seg_lines lines) —
// for batches whose reads differ in length, where a lane per read makes a wave wait for its longest read and the batch for its longest lane.
// The hits must be the whole read's scan's, so the segments divide them by VALLEY (a maximal run of columns with cost <= k: the local-minimum
// machine acts inside valleys and on the steps into / out of them only): a valley belongs to the segment that holds its first column.
//   * a segment other than the first starts m + k columns early from the all-insertions column (values <= k are exact after that, larger ones
//     stay larger: k_flank_verify's lead-in) and does not report (the machine sees a budget of -1) until it has seen a column > k at or after
//     the last column before its own first — a valley in progress there began earlier and is the previous segment's;
//   * a segment follows a valley it owns past its last line, until the first column > k (then it is done), to the read's end and over the
//     overhang positions if need be; valleys that begin in the overhang positions are the last segment's.
// Counts go to a cell per (cut segment, group, strand), hits carry the segment (bit 31 set): k_seg_fold / k_seg_hits turn both into the
// read's (ordinals run on from segment to segment in scan order).
template <int W, int STRAND, bool SEG = false>
__device__ __forceinline__ void flank_scan_lane(const uint8_t* __restrict__ bases, const uint64_t* __restrict__ offsets,
                                                uint32_t n_reads, const uint8_t* __restrict__ tables, int32_t kk, int m, int32_t score0,
                                                uint32_t off_pv0, uint32_t off_ovh, int ovh_steps, int pol_lm,
                                                uint32_t g, uint32_t n_groups, uint32_t* __restrict__ cnt,
                                                bb_hit_raw* __restrict__ hits, uint32_t hit_cap, uint32_t* __restrict__ hit_count,
                                                const uint32_t* s_peq, uint4* s_line /* this wave's [BB_SCAN_LQ][64] */, uint32_t chunk /* which 256 reads (SEG: segments) */,
                                                const uint2* __restrict__ vtab = nullptr, uint32_t n_virtual = 0, uint32_t seg_lines = 0, uint32_t split_above = 0,
                                                const uint32_t* __restrict__ vcut = nullptr, uint32_t* __restrict__ vcnt = nullptr) {
    constexpr int S = (W <= 2 ? 2 : (W <= 4 ? 4 : 8));
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t slot = chunk * 256u + threadIdx.x;
    const bool live = slot < (SEG ? n_virtual : n_reads);
    const uint2 vt = SEG && live ? vtab[slot] : make_uint2(0u, 0u);
    const uint32_t rd = SEG ? vt.x : slot;   // the read
    const uint64_t off = live ? offsets[rd] : 0ull;
    const uint32_t n = live ? (uint32_t)(offsets[rd + 1] - off) : 0u;
    const uint8_t* rb = bases + off;
    const int TB = (m - 1) & 31;
    const uint32_t* pv0 = reinterpret_cast<const uint32_t*>(tables + off_pv0);
    const int32_t* ovh = reinterpret_cast<const int32_t*>(tables + off_ovh);

    // geometry of the walk in forward byte coordinates [0, n): the lane's lines are the LB-byte-aligned lines of HBM that hold its read, in scan order
    const uint64_t a0 = (uint64_t)(uintptr_t)rb;
    constexpr uint32_t LB = BB_SCAN_LQ * 16u, LSH = BB_SCAN_LQ == 16 ? 8u : BB_SCAN_LQ == 8 ? 7u : 6u;  // line bytes (256, 128 or 64)
    static_assert(!SEG || BB_SCAN_LQ == 8u, "segments are counted in 128-byte lines");
    const uint32_t mis = STRAND == 0 ? (uint32_t)(a0 & (LB - 1u)) : (uint32_t)((LB - (uint32_t)((a0 + n) & (LB - 1u))) & (LB - 1u));
    const uint32_t nlines_read = n ? (mis + n + LB - 1u) >> LSH : 0u;
    // SEG: lines [l_first, l_end) of the read are the lane's own, scanning starts at l_begin (lead-in) and may go on past l_end (a valley it owns)
    const bool cut = SEG && nlines_read > split_above;
    const uint32_t l_first = cut ? vt.y * seg_lines : 0u;
    const uint32_t l_end = cut ? min(nlines_read, l_first + seg_lines) : nlines_read;
    const bool last_seg = l_end == nlines_read;
    const uint32_t l_begin = l_first - min(l_first, ((uint32_t)m + (uint32_t)kk + LB) >> LSH);
    const uint32_t s_pos = l_first ? (l_first << LSH) - mis : 0u;            // first own scan position
    const uint32_t e_pos = last_seg ? 0xFFFFFFFFu : (l_end << LSH) - mis;    // one past the last own one (the last segment owns the overhang positions too)
    const uint32_t cs = cut ? vcut[slot] : 0u;                               // the segment's cell (cut reads)
    const uint32_t read = cut ? (cs | 0x80000000u) : rd;                     // what the hits carry
    uint32_t* const cnt_cell = live ? (cut ? vcnt + ((uint64_t)cs * n_groups + g) * 2 + STRAND : cnt + ((uint64_t)rd * n_groups + g) * 2 + STRAND) : nullptr;
    bool armed = l_first == 0u, fin = false;
    int32_t kkl = armed ? kk : -1;   // the budget the local-minimum machine sees

    uint32_t pv[W], mv[W];
#pragma unroll
    for (int w = 0; w < W; ++w) { pv[w] = pv0[w]; mv[w] = 0; }
    int32_t sc = score0;
    lm_lane st = {score0, 1u, 0u, 0u};
    hit_buf hb = {0u, 0u, 0u, 0u, 0u};
    uint32_t idx = 0;  // scan position (characters consumed)
    if (SEG && l_begin) {   // from the all-insertions column, m + k columns (whole lines of them) ahead of the segment
#pragma unroll
        for (int x = 0; x < W; ++x) { const int bt = m - 32 * x; pv[x] = bt >= 32 ? 0xFFFFFFFFu : (bt > 0 ? ((1u << bt) - 1u) : 0u); }
        sc = m; st.prev = m;
        idx = (l_begin << LSH) - mis;
    }
    const bool lm_left = pol_lm == BB_LM_PLATEAU_LEFT, lm_strict = pol_lm == BB_LM_STRICT;
    // SEG, after a column (or four) that left the score at sc: the steps between valleys
    auto seg_after = [&]() {
        if (!armed) {
            if (idx >= e_pos) fin = true;                                   // a valley of an earlier segment covers this one
            else if (sc > kk && idx >= s_pos) { armed = true; kkl = kk; }
        } else if (sc > kk && idx >= e_pos) { kkl = -1; fin = true; }
    };

    auto step = [&](uint32_t ch) {
        uint32_t eq[W], d0[W], ph[W], mh[W];
        load_eq<W, S>(s_peq, ch, eq);
        myers_step<W>(pv, mv, eq, d0, ph, mh);
        sc += (int32_t)((ph[W - 1] >> TB) & 1u) - (int32_t)((mh[W - 1] >> TB) & 1u);
        ++idx;
        if constexpr (SEG) {
            { const int32_t kk = kkl; BB_LM_STEP_BUF(st, sc, idx); }
            seg_after();
        } else {
            BB_LM_STEP_BUF(st, sc, idx);
        }
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
            if constexpr (SEG) seg_after();
        }
    };
#if BB_SCAN_UNALIGNED == 2
    // Line-aligned streaming: the lane's lines are the LB-byte-aligned lines of HBM that hold its read, in scan order;
    // `mis` bytes of the first line (scan order) lie before the read's first scanned byte, and the last line may end
    // early.  Those two partial lines go through the same LDS path with the bytes outside the read predicated off, so
    // every line of the batch is requested once per strand and no lane runs a byte loop of its own.  (A line that
    // holds one byte of the read lies in that byte's page: the bytes outside the read are fetched, never used.)
    const uint32_t nlines = nlines_read;
    const uint8_t* line0 = STRAND == 0 ? rb - mis : rb + n + mis - LB;  // first line in scan order
    uint32_t lmax = nlines;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) lmax = max(lmax, (uint32_t)__shfl_xor((int)lmax, d, 64));
    lmax = __builtin_amdgcn_readfirstlane(lmax);
    for (uint32_t lj = 0; SEG || lj < lmax; ++lj) {
        const uint32_t l = l_begin + lj;   // line of the read, in scan order
        // SEG: the lane's own lines, then on while a valley it owns is open
        const bool on = SEG ? (live && !fin && l < nlines && (l < l_end || armed)) : l < nlines;
        if (SEG && !__any(on)) break;
        if (on) {
            const uint8_t* src = STRAND == 0 ? line0 + ((uint64_t)l << LSH) : line0 - ((uint64_t)l << LSH);
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

    if constexpr (SEG)   // the overhang positions and the pending minimum: the lane that got to the read's end with the right to report, or to earn it there
        scan_finish<W, STRAND>(live, n, m, kk, sc, pv, mv, idx, st, hb, ovh, read, g, n_groups, cnt, hits, hit_cap, hit_count, pol_lm, !fin && idx == n, ovh_steps,
                               nullptr, nullptr, cnt_cell, true, armed, last_seg);
    else
        scan_finish<W, STRAND>(live, n, m, kk, sc, pv, mv, idx, st, hb, ovh, read, g, n_groups, cnt, hits, hit_cap, hit_count, pol_lm, true, ovh_steps);
}

// The groups of one launch (round 5).  Every group of a context scans the same reads; launched one after the other each pass streamed the
// batch from HBM again (custom dual-end 3.1 x, SQK-RBK114-96 extended 4.8 x the algorithmic bytes).  One launch now carries all groups of
// a kind, and the block index is laid out so that the blocks that stream the SAME 256 reads in the SAME direction — one per group — are
// dispatched back to back onto the SAME XCD (block b is observed to run on XCD b % 8: a speed assumption, never a correctness one): they
// run side by side and the second to ask for a line finds it in that XCD's L2.
struct bb_glist { uint32_t n; uint8_t g[28]; };
// block -> (group index, strand or pass, chunk of 256 reads): blocks lin and lin + 8 sit on one XCD, so `gi` varies fastest there
__device__ __forceinline__ bool bb_coscheduled(uint32_t lin, uint32_t n_g, uint32_t n_pass, uint32_t n_chunks, uint32_t& gi, uint32_t& pass, uint32_t& chunk) {
    const uint32_t xcd = lin & 7u, slot = lin >> 3;
    gi = slot % n_g;
    const uint32_t rest = slot / n_g;
    pass = rest % n_pass;
    chunk = (rest / n_pass) * 8u + xcd;
    return chunk < n_chunks;
}
static inline uint32_t bb_coscheduled_blocks(uint32_t n_g, uint32_t n_pass, uint32_t n_chunks) { return ((n_chunks + 7u) / 8u) * 8u * n_g * n_pass; }

template <int W>
__global__ __launch_bounds__(256) void k_flank_scan2(const uint8_t* __restrict__ bases, const uint64_t* __restrict__ offsets,
                                                     uint32_t n_reads, const uint8_t* __restrict__ tables,
                                                     const bb_group_dev* __restrict__ groups, bb_glist gl, uint32_t n_groups,
                                                     uint32_t* __restrict__ cnt, bb_hit_raw* __restrict__ hits,
                                                     uint32_t hit_cap, uint32_t* __restrict__ hit_count) {
    constexpr int S = (W <= 2 ? 2 : (W <= 4 ? 4 : 8));
    __shared__ __attribute__((aligned(16))) uint32_t s_peq[256 * S];
    __shared__ __attribute__((aligned(16))) uint4 s_lines[4][BB_SCAN_LQ * 64];
    uint32_t gi, strand, chunk;
    if (!bb_coscheduled(blockIdx.x, gl.n, 2u, (n_reads + 255u) / 256u, gi, strand, chunk)) return;
    const uint32_t g = gl.g[gi];
    const bb_group_dev* G = groups + g;
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
        flank_scan_lane<W, 0>(bases, offsets, n_reads, tables, kk, m, score0, o_pv0, o_ovh, G->ovh_steps, G->pol_lm, g, n_groups, cnt, hits, hit_cap, hit_count, s_peq, line, chunk);
    else
        flank_scan_lane<W, 1>(bases, offsets, n_reads, tables, kk, m, score0, o_pv0, o_ovh, G->ovh_steps, G->pol_lm, g, n_groups, cnt, hits, hit_cap, hit_count, s_peq, line, chunk);
}

// The full scan of a batch whose reads differ in length: a lane per SEGMENT (flank_scan_lane<.., SEG>)
template <int W>
__global__ __launch_bounds__(256) void k_flank_scan_seg(const uint8_t* __restrict__ bases, const uint64_t* __restrict__ offsets,
                                                        uint32_t n_reads, const uint8_t* __restrict__ tables,
                                                        const bb_group_dev* __restrict__ groups, bb_glist gl, uint32_t n_groups,
                                                        uint32_t* __restrict__ cnt, bb_hit_raw* __restrict__ hits,
                                                        uint32_t hit_cap, uint32_t* __restrict__ hit_count,
                                                        const uint2* __restrict__ vtab, uint32_t n_virtual, uint32_t seg_lines, uint32_t split_above,
                                                        const uint32_t* __restrict__ vcut, uint32_t* __restrict__ vcnt) {
    constexpr int S = (W <= 2 ? 2 : (W <= 4 ? 4 : 8));
    __shared__ __attribute__((aligned(16))) uint32_t s_peq[256 * S];
    __shared__ __attribute__((aligned(16))) uint4 s_lines[4][BB_SCAN_LQ * 64];
    uint32_t gi, strand, chunk;
    if (!bb_coscheduled(blockIdx.x, gl.n, 2u, (n_virtual + 255u) / 256u, gi, strand, chunk)) return;
    const uint32_t g = gl.g[gi];
    const bb_group_dev* G = groups + g;
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
        flank_scan_lane<W, 0, true>(bases, offsets, n_reads, tables, kk, m, score0, o_pv0, o_ovh, G->ovh_steps, G->pol_lm, g, n_groups, cnt, hits, hit_cap, hit_count, s_peq, line, chunk,
                                    vtab, n_virtual, seg_lines, split_above, vcut, vcnt);
    else
        flank_scan_lane<W, 1, true>(bases, offsets, n_reads, tables, kk, m, score0, o_pv0, o_ovh, G->ovh_steps, G->pol_lm, g, n_groups, cnt, hits, hit_cap, hit_count, s_peq, line, chunk,
                                    vtab, n_virtual, seg_lines, split_above, vcut, vcnt);
}
// After it: the counts of a cut read's segments folded into the read's (each cell left holding the hits of the segments before it) ...
__global__ __launch_bounds__(256) void k_seg_fold(const uint4* __restrict__ cutlist /* read, first cell, segments, - */, uint32_t n_cut, uint32_t n_groups, uint32_t gmask,
                                                  uint32_t* __restrict__ vcnt, uint32_t* __restrict__ cnt) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    const uint32_t per = n_groups * 2u;
    if (i >= n_cut * per) return;
    const uint4 c = cutlist[i / per];
    const uint32_t gs = i % per;
    if (!((gmask >> (gs >> 1)) & 1u)) return;   // a group whose scan went by reads (filter + verification) wrote the read's count itself
    uint32_t run = 0u;
    for (uint32_t t = 0; t < c.z; ++t) {
        uint32_t* cell = vcnt + (uint64_t)(c.y + t) * per + gs;
        const uint32_t v = *cell;
        *cell = run;
        run += v;
    }
    cnt[(uint64_t)c.x * per + gs] = run;
}
// ... and the hits of segments made hits of their reads: ordinals run on from segment to segment
__global__ __launch_bounds__(256) void k_seg_hits(bb_hit_raw* __restrict__ hits, const uint32_t* __restrict__ hit_count, uint32_t hit_cap, uint32_t n_groups,
                                                  const uint32_t* __restrict__ vcnt, const uint32_t* __restrict__ cutread) {
    const uint32_t n = min(*hit_count, hit_cap);
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < n; i += gridDim.x * 256u) {
        bb_hit_raw h = hits[i];
        if (!(h.read_idx & 0x80000000u)) continue;
        const uint32_t cs = h.read_idx & 0x7FFFFFFFu;
        h.ordinal += vcnt[((uint64_t)cs * n_groups + h.group) * 2u + h.strand];
        h.read_idx = cutread[cs];
        hits[i] = h;
    }
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
#ifndef BB_FILT_SCORE5
#define BB_FILT_SCORE5 1   // the two halves' score updates as one subtraction and one arithmetic shift (18.75 instead of 19.75 instructions per column)
#endif
template <bool WIDE>
__global__ __launch_bounds__(256) void k_flank_filter(const uint8_t* __restrict__ bases, const uint64_t* __restrict__ offsets, uint32_t n_reads,
                                                      const uint8_t* __restrict__ tables, const bb_group_dev* __restrict__ groups, bb_glist gl,
                                                      uint32_t* __restrict__ flags_all, uint64_t words_per_strand, unsigned long long* __restrict__ n_flagged_all,
                                                      const uint2* __restrict__ vtab, uint32_t n_virtual, uint32_t seg_lines, uint32_t split_above) {
    __shared__ uint32_t s_fpeq[WIDE ? 512 : 256];
    __shared__ __attribute__((aligned(16))) uint4 s_lines[4][BB_SCAN_LQ * 64];
    static_assert(BB_SCAN_LQ == 8u, "piece bits assume 128-byte lines");
    // one block per (group of the launch, 256 reads); the groups' blocks for the same reads co-scheduled on one XCD (bb_coscheduled).
    // Group gi of the launch writes its flag words into region gi of the array (2 * words_per_strand words each) and counts into its own cell.
    uint32_t gi, pass_, chunk;
    if (!bb_coscheduled(blockIdx.x, gl.n, 1u, ((vtab ? n_virtual : n_reads) + 255u) / 256u, gi, pass_, chunk)) return;
    const uint32_t g = gl.g[gi];
    uint32_t* flags = flags_all + (uint64_t)gi * 2ull * words_per_strand;
    unsigned long long* n_flagged = n_flagged_all + g;
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
    // A lane takes a read, or — where the batch's reads differ in length (bb_len.h: vtab, sorted by falling length so that a wave's lanes
    // finish together) — ONE SEGMENT of a read: reads of more than split_above lines are cut into segments of seg_lines lines (a multiple
    // of 4: a flag word holds 4 lines, so no two segments share one).  A segment other than the read's first starts from the all-insertions
    // column `lead` lines early: a window score <= k depends on the last R + k columns only, so after them the values <= k are exact and
    // the larger ones stay larger (the argument of k_flank_verify's lead-in); what the lead-in lines flag belongs to the segment before.
    const uint32_t vslot = chunk * 256u + threadIdx.x;
    const bool live = vslot < (vtab ? n_virtual : n_reads);
    const uint2 vt = live && vtab ? vtab[vslot] : make_uint2(vslot, 0u);
    const uint32_t read = vt.x, seg = vt.y;
    const uint64_t off0 = offsets[0];
    const uint64_t off = live ? offsets[read] : off0;
    const uint32_t n = live ? (uint32_t)(offsets[read + 1] - off) : 0u;
    const uint8_t* rb = bases + off;
    constexpr uint32_t LB = 128u, LSH = 7u;
    const uint32_t mis = (uint32_t)((uint64_t)(uintptr_t)rb & (LB - 1u));
    const uint32_t nlines_read = n ? (mis + n + LB - 1u) >> LSH : 0u;
    const bool cut = vtab && nlines_read > split_above;
    const uint32_t l_first = cut ? seg * seg_lines : 0u;
    const uint32_t nlines = cut ? min(nlines_read, l_first + seg_lines) : nlines_read;   // one past the segment's last line
    const bool last_seg = nlines == nlines_read;
    const uint32_t lead = l_first ? min(l_first, ((uint32_t)R + (uint32_t)min(G->flank_k, 127) + LB) >> LSH) : 0u;
    const uint32_t l_begin = l_first - lead;
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
    uint32_t pv = WIDE ? (l_first ? maskR : pvA0) : ((l_first ? maskR : pvA0) << SA) | (maskR << (SA + 16u)), mv = 0u;
    uint32_t pvB = maskR, mvB = 0u;  // WIDE: the rc strand's word
    // Both blocks' D[R][i], biased by 15 - k, in the two halves of one register (the bottom rows' delta bits sit at bits 14
    // and 30: one mask, one shift): a half's bit 4 is clear exactly while its score is <= k, so AND-ing the register over
    // the columns of a piece leaves bit 4 / bit 20 clear iff the piece holds such a column.  (WIDE: one register per strand,
    // bias 31 - k, bit 5.)
    const uint32_t TOPS = 0x40004000u;
    const int bias = (WIDE ? 31 : 15) - kk;
    const int scA_init = l_first ? R : scA0;
    uint32_t sc2 = WIDE ? (uint32_t)(scA_init + bias) : ((uint32_t)(R + bias) << 16) | (uint32_t)(scA_init + bias);
    uint32_t scB = (uint32_t)(R + bias);
    uint32_t keep = l_first ? 0xFFFFFFFFu : (WIDE ? sc2 | ~0x20u : sc2 | ~0x00100010u);  // column 0 (of the read) counts for the first piece
    uint32_t keepB = scB | ~0x20u;
    uint32_t bitsA = 0u, bitsB = 0u, nflag = 0u;
    // one column of the narrow form on the column's Eq word (both blocks)
    auto step_eq = [&](const uint32_t eq) {
        const uint32_t x = eq & pv;
        const uint32_t d0 = bitop3<BB_TT_XOR_OR>(x + pv, pv, eq) | mv;
        const uint32_t ph = bitop3<BB_TT_OR_NOR>(mv, d0, pv) & BM, mh = pv & d0;
#if BB_FILT_SCORE5
        // both halves' +-1 at once: the bottom rows' delta bits (14 and 30) subtracted as whole words — a borrow out of the low half is paid
        // back by the arithmetic shift, and the biased scores never go below zero — one instruction less than two shifts and two masks
        sc2 += (uint32_t)((int32_t)((ph & TOPS) - (mh & TOPS)) >> 14);
#else
        sc2 = sc2 + ((ph & TOPS) >> 14) - ((mh & TOPS) >> 14);
#endif
        keep &= sc2;
        const uint32_t phs = shl1_32(ph), mhs = shl1_32(mh);
        pv = bitop3<BB_TT_OR_NOR>(mhs, d0, phs) & BM;
        mv = phs & d0;
    };
    auto step = [&](uint32_t chr) {
        if constexpr (WIDE) {
            const uint2 e2 = *reinterpret_cast<const uint2*>(s_fpeq + 2u * chr);
            {
                // (a word per strand, rows at bits 0..R-1: what lies above them never flows down — carries and shifts go up — so pv keeps its
                // upper bits unmasked and only the two horizontal vectors the score reads are cut to the rows: one mask less per column)
                const uint32_t eq = e2.x, x = eq & pv;
                const uint32_t d0 = bitop3<BB_TT_XOR_OR>(x + pv, pv, eq) | mv;
                const uint32_t ph = bitop3<BB_TT_OR_NOR>(mv, d0, pv) & BM, mh = bitop3<0x80>(pv, d0, BM);
                sc2 = sc2 + (ph >> (R - 1)) - (mh >> (R - 1));
                keep &= sc2;
                const uint32_t phs = shl1_32(ph), mhs = shl1_32(mh);
                pv = bitop3<BB_TT_OR_NOR>(mhs, d0, phs);
                mv = phs & d0;
            }
            {
                const uint32_t eq = e2.y, x = eq & pvB;
                const uint32_t d0 = bitop3<BB_TT_XOR_OR>(x + pvB, pvB, eq) | mvB;
                const uint32_t ph = bitop3<BB_TT_OR_NOR>(mvB, d0, pvB) & BM, mh = bitop3<0x80>(pvB, d0, BM);
                scB = scB + (ph >> (R - 1)) - (mh >> (R - 1));
                keepB &= scB;
                const uint32_t phs = shl1_32(ph), mhs = shl1_32(mh);
                pvB = bitop3<BB_TT_OR_NOR>(mhs, d0, phs);
                mvB = phs & d0;
            }
        } else {
            step_eq(s_fpeq[chr]);
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
    uint32_t lmax = nlines - l_begin;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) lmax = max(lmax, (uint32_t)__shfl_xor((int)lmax, d, 64));
    lmax = __builtin_amdgcn_readfirstlane(lmax);
    for (uint32_t lj = 0; lj < lmax; ++lj) {
        const uint32_t l = l_begin + lj;   // line of the read
        const bool on = l < nlines;
        if (on) {
            const uint8_t* src = line0 + ((uint64_t)l << LSH);
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
        if (on && l < l_first) { bitsA = 0u; bitsB = 0u; }   // a lead-in line: its flags are the previous segment's
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
    if (live && n && last_seg && (G->filt_mode & BB_FILT_RC_BEGIN_HINT)) {
        const uint32_t pb = WIDE ? pvB & maskR : (pv >> (SA + 16u)) & maskR, mb = WIDE ? mvB & maskR : (mv >> (SA + 16u)) & maskR;
        uint32_t hint = 0u;
        for (int o = 1; o < R; ++o) {
            const uint32_t low = (1u << (R - o)) - 1u;
            if ((int32_t)__popc(pb & low) - (int32_t)__popc(mb & low) + ovh[o] <= G->flank_k) hint = 1u;
        }
        if (hint) fl1[(nlines_read + 3u) >> 2] = hint;
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
    const uint32_t topmask = TB == 31 ? 0xFFFFFFFFu : ((2u << TB) - 1u);
    auto score_now = [&]() {   // D[m][idx] from the vertical deltas (the top row is 0)
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
            constexpr int CH = W <= 2 ? BB_VERIFY_CHUNKS : BB_VERIFY_CHUNKS_WIDE;   // three-word flanks (67..96 nt): an interval is ~110 columns
            uint32_t wt[CH][4];
            const uint32_t cntb = work ? min(16u * CH, stop - cur) : 0u;
#pragma unroll
            for (int q = 0; q < CH; ++q) {
                wt[q][0] = wt[q][1] = wt[q][2] = wt[q][3] = 0u;
                if (cntb > (uint32_t)(16 * q)) load16(cur + 16u * (uint32_t)q, wt[q]);
            }
#pragma unroll
            for (int hb2 = 0; hb2 < CH; ++hb2) {
                if (__any(cntb > (uint32_t)(16 * hb2))) {
#if BB_VERIFY_FAST
                    // The lead-in of an interval starts from the all-insertions column (score m) and the score falls by at most one per column:
                    // for its first m - k - 4 columns nothing can be reported and neither the score nor the local-minimum state needs tracking
                    // (k_flank_scan2's fast path: the score re-derived from the vertical deltas afterwards).  Most lanes of a wave start their
                    // intervals with the round, so the test is wave-uniform often enough: 15 instead of 27 instructions per column there.
#pragma unroll
                    for (int b0 = 0; b0 < 16; b0 += 4) {
                        const bool in4 = (uint32_t)(16 * hb2 + b0 + 3) < cntb;   // all four columns are the lane's
                        const bool part = !in4 && (uint32_t)(16 * hb2 + b0) < cntb;
                        if (__any(part || (in4 && sc <= kk + 4))) {
#pragma unroll
                            for (int b = b0; b < b0 + 4; ++b) {
                                const uint32_t w = wt[hb2][b >> 2];
                                if ((uint32_t)(16 * hb2 + b) < cntb) step((w >> (8 * (b & 3))) & 0xFFu);
                            }
                        } else if (in4) {   // (lanes without these columns sit the group out)
#pragma unroll
                            for (int b = b0; b < b0 + 4; ++b) step_fast((wt[hb2][b >> 2] >> (8 * (b & 3))) & 0xFFu);
                            idx += 4;
                            sc = score_now();
                            st.prev = sc;  // > k: the lazily evaluated `dec` needs no update (see lm_lane)
                        }
                    }
#else
#pragma unroll
                    for (int b = 0; b < 16; ++b) {
                        const uint32_t w = wt[hb2][b >> 2];
                        if ((uint32_t)(16 * hb2 + b) < cntb) step((w >> (8 * (b & 3))) & 0xFFu);
                    }
#endif
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

// (106 VGPRs in the W = 2 instantiation: four waves per SIMD.  Held to five — __launch_bounds__(256, 5): 96 VGPRs, 9 spilled — the scan stage
// was 0.05 ms faster and the kernel fetched 0.2 GB more per step, the lines of its intervals surviving less often between a lane's chunks:
// BB_VERIFY_MINBLOCKS=5)
#ifndef BB_VERIFY_MINBLOCKS
#define BB_VERIFY_MINBLOCKS 1
#endif
template <int W>
__global__ __launch_bounds__(256, W <= 4 ? BB_VERIFY_MINBLOCKS : 1) void k_flank_verify(const uint8_t* __restrict__ bases, const uint64_t* __restrict__ offsets, uint32_t n_reads,
                                                      const uint8_t* __restrict__ tables, const bb_group_dev* __restrict__ groups, uint32_t g,
                                                      uint32_t n_groups, const uint32_t* __restrict__ flags, uint64_t words_per_strand,
                                                      uint32_t* __restrict__ cnt, bb_hit_raw* __restrict__ hits, uint32_t hit_cap,
                                                      uint32_t* __restrict__ hit_count, uint32_t* __restrict__ queues, uint32_t swap_strands) {
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
    // swap_strands: the flags are a TWIN group's (bb_ctx::filt_twin) — its rc block is this group's forward window and the other way round
    const uint32_t* fls = flags + ((strand ^ swap_strands) ? words_per_strand : 0ull);
    if (strand == 0)
        flank_verify_lane<W, 0>(bases, offsets, n_reads, tables, G, g, n_groups, fls, cnt, hits, hit_cap, hit_count, queues, s_peq, stage, s_flws[threadIdx.x >> 6]);
    else
        flank_verify_lane<W, 1>(bases, offsets, n_reads, tables, G, g, n_groups, fls, cnt, hits, hit_cap, hit_count, queues + 1, s_peq, stage, s_flws[threadIdx.x >> 6]);
}

// ------------------------------------------------------------------------------------------------
// exclusive scan of uint32 (3 small kernels): 2048 elements per block
// ------------------------------------------------------------------------------------------------
// (the scans' last input is a place holder whose output is the total: read as 0 whatever it holds, so that nobody has to zero it)
__global__ __launch_bounds__(256) void k_scan_block(const uint32_t* __restrict__ in, uint32_t* __restrict__ out, uint64_t n,
                                                    uint32_t* __restrict__ sums) {
    __shared__ uint32_t s_w[4];
    const uint64_t base = (uint64_t)blockIdx.x * 2048u + (uint64_t)threadIdx.x * 8u;
    uint32_t v[8], t = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) { v[i] = base + i + 1 < n ? in[base + i] : 0u; }
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
__global__ __launch_bounds__(256) void k_scan_add(uint32_t* __restrict__ out, uint64_t n, const uint32_t* __restrict__ sums, uint32_t* __restrict__ total) {
    const uint64_t base = (uint64_t)blockIdx.x * 2048u + (uint64_t)threadIdx.x * 8u;
    const uint32_t a = sums[blockIdx.x];
#pragma unroll
    for (int i = 0; i < 8; ++i) if (base + i < n) { const uint32_t v = out[base + i] + a; out[base + i] = v; if (total && base + i + 1 == n) *total = v; }
}
// The same scan in ONE launch for inputs one block walks in a few rounds (a small batch: three launches cost its host thread more than the
// scan costs the device): 1024 lanes, 8 values each per round, the carry in a register of every lane.
__global__ __launch_bounds__(1024) void k_scan_one(const uint32_t* __restrict__ in, uint32_t* __restrict__ out, uint32_t n, uint32_t* __restrict__ total) {
    __shared__ uint32_t s_w[16];
    const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    uint32_t carry = 0u;
    for (uint32_t r0 = 0; r0 < n; r0 += 8192u) {   // (block-uniform trip count)
        const uint32_t base = r0 + threadIdx.x * 8u;
        uint32_t v[8], t = 0u;
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = base + i + 1 < n ? in[base + i] : 0u;
#pragma unroll
        for (int i = 0; i < 8; ++i) { const uint32_t x = v[i]; v[i] = t; t += x; }
        uint32_t inc = t;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const uint32_t y = __shfl_up(inc, d, 64); if (lane >= d) inc += y; }
        __syncthreads();   // the last round's s_w has been read by everyone
        if (lane == 63u) s_w[wv] = inc;
        __syncthreads();
        uint32_t wbase = 0u, all = 0u;
#pragma unroll
        for (uint32_t i = 0; i < 16u; ++i) { const uint32_t x = s_w[i]; if (i < wv) wbase += x; all += x; }
        const uint32_t excl = carry + wbase + inc - t;
#pragma unroll
        for (int i = 0; i < 8; ++i) if (base + i < n) { out[base + i] = v[i] + excl; if (total && base + i + 1 == n) *total = v[i] + excl; }
        carry += all;
    }
}
#define BB_SCAN_ONE_MAX (1u << 16)   // inputs up to this many values (+ the place holder) take k_scan_one

