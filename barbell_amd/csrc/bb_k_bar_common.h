// bb_k_bar_common.h — what every barcode-stage kernel shares (searcher.rs:267-426): the provisional row / winner record, the
// closed-form sub-path, the per-hit argmax + row writer of the exact kernels, the Lodhi replay (exact, f64) and its bound (f32,
// table-driven), the bit-parallel local-minimum pick.
#pragma once
#include "bb_myers.h"

__device__ __forceinline__ int32_t rel_dist_to_end(int64_t pos, int64_t read_len) {  // searcher.rs:183-199
    if (pos < 0) return 1;
    if (pos <= read_len / 2) return pos == 0 ? 1 : (int32_t)pos;
    if (pos == read_len) return -1;
    return (int32_t)-(read_len - pos);
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

// The same bound, eight columns at a time, under the policy's decay exponents ([H8]; default 1 per op).  A text column advances the
// time by its op's exponent: eM on a Match column, at least eX = min(eS, eI) on any other (the bound sees Match bits only: taking the
// smaller of the two can only shorten spans); Del columns are dropped as before.  Over the columns 8q+1 .. 8q+8 the recurrence is
// affine in (sc, b2, b1), and with u2 = b2 / 2^tau, u1 = b1 / 2^tau (tau = the time before the byte) its coefficients depend on the
// byte of Match bits only:
//   sc += A u2 + B u1 + C;   u2 = (u2 + n u1 + D) S;   u1 = (u1 + E) S
// With s(r) = the time after position r of the byte: A = sum 2^-s(r), B = sum 2^-s(r) cnt(r), C = sum 2^-s(r) P2(r) over the byte's Match
// positions r = 1..8, cnt(r) the Matches before r, P1(r) = sum of 2^(s(r') - eM) over them, P2(r) = sum of P1 over them; n = all
// Matches, D = P2(9), E = P1(9), S = 2^-s(8) (1/256 for the default exponents).  Two 16-byte table reads and eight f32 operations per
// byte instead of nine instructions per column; the entries are rounded up, every term is positive, fewer than 60 roundings enter a
// result: the (1 + 2^-14) scale keeps it a bound.
struct __attribute__((aligned(32))) bb_lb_entry { float A, B, C, n, D, E, S, _p1; };
// In LDS the 256 entries sit as two planes of 16 bytes — (A, B, C, n) of every entry, then (D, E, S, -) — not as 32-byte rows: a lane reads
// both halves at a data-dependent index, and 16-byte rows fall on sixteen bank groups where 32-byte rows fall on eight.
#ifndef BB_LB_SPLIT
#define BB_LB_SPLIT 1
#endif
__device__ __forceinline__ void lb_put(bb_lb_entry* tab, uint32_t i, const bb_lb_entry& e) {
#if BB_LB_SPLIT
    float4* pl = reinterpret_cast<float4*>(tab);
    pl[i] = make_float4(e.A, e.B, e.C, e.n); pl[256u + i] = make_float4(e.D, e.E, e.S, 0.0f);
#else
    tab[i] = e;
#endif
}
__device__ __forceinline__ float4 lb_a(const bb_lb_entry* tab, uint32_t i) {
#if BB_LB_SPLIT
    return reinterpret_cast<const float4*>(tab)[i];
#else
    return *reinterpret_cast<const float4*>(&tab[i].A);
#endif
}
__device__ __forceinline__ float4 lb_d(const bb_lb_entry* tab, uint32_t i) {
#if BB_LB_SPLIT
    return reinterpret_cast<const float4*>(tab)[256u + i];
#else
    return *reinterpret_cast<const float4*>(&tab[i].D);
#endif
}
__device__ __forceinline__ void lodhi_bound_table_entry(uint32_t byte, uint32_t expk, bb_lb_entry& e) {
    const int eM = (int)(expk & 0xFFu), eS = (int)((expk >> 8) & 0xFFu), eI = (int)((expk >> 16) & 0xFFu), eX = eS < eI ? eS : eI;
    auto p2 = [](int k) -> double { return __hiloint2double((int)((uint32_t)(1023 + k) << 20), 0); };  // 2^k, |k| < 1023
    double A = 0.0, B = 0.0, C = 0.0, cnt = 0.0, P1 = 0.0, P2 = 0.0;
    int s = 0;
    for (int r = 1; r <= 8; ++r) {
        if ((byte >> (r - 1)) & 1u) {
            s += eM;
            const double w = p2(-s);
            A += w; B += w * cnt; C += w * P2;
            P2 += P1; cnt += 1.0; P1 += p2(s - eM);
        } else s += eX;
    }
    e.A = __double2float_ru(A); e.B = __double2float_ru(B); e.C = __double2float_ru(C); e.n = (float)cnt;
    e.D = __double2float_ru(P2); e.E = __double2float_ru(P1); e.S = (float)p2(-s); e._p1 = 0.0f;
}
// The table for a bound over the pattern's ROWS when Sub and Del rows decay differently (eS != eD): four rows per entry, two bits per row —
// index = a | b << 4 with (a, b) of a row = (1, 1) Match, (0, 1) Sub, (0, 0) Del, (1, 0) unknown (a shared row the walk did not match:
// the smaller of the two exponents).  Same recurrence, same coefficients; only the time a non-Match row takes depends on its class.
__device__ __forceinline__ void lodhi_bound_table_entry4(uint32_t idx, uint32_t e4, bb_lb_entry& e) {
    const int eM = (int)(e4 & 0xFFu), eS = (int)((e4 >> 8) & 0xFFu), eD = (int)((e4 >> 24) & 0xFFu), eX = eS < eD ? eS : eD;
    auto p2 = [](int k) -> double { return __hiloint2double((int)((uint32_t)(1023 + k) << 20), 0); };  // 2^k, |k| < 1023
    double A = 0.0, B = 0.0, C = 0.0, cnt = 0.0, P1 = 0.0, P2 = 0.0;
    int s = 0;
    for (int r = 0; r < 4; ++r) {
        const uint32_t a = (idx >> r) & 1u, b = (idx >> (4 + r)) & 1u;
        if (a & b) {
            s += eM;
            const double w = p2(-s);
            A += w; B += w * cnt; C += w * P2;
            P2 += P1; cnt += 1.0; P1 += p2(s - eM);
        } else s += a ? eX : (b ? eS : eD);
    }
    e.A = __double2float_ru(A); e.B = __double2float_ru(B); e.C = __double2float_ru(C); e.n = (float)cnt;
    e.D = __double2float_ru(P2); e.E = __double2float_ru(P1); e.S = (float)p2(-s); e._p1 = 0.0f;
}
// rows 4q+1 .. 4q+4 <-> bits 4q .. 4q+3 of the two masks
// the state after eight Matches, by the class table's own steps (two entries of four Match rows)
__device__ __forceinline__ void lodhi_bound_first_rows4(const bb_lb_entry* tab, float& sc, float& u1, float& u2) {
    sc = 0.0f; u1 = 0.0f; u2 = 0.0f;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const float4 t0 = lb_a(tab, 0xFFu), t1 = lb_d(tab, 0xFFu);
        sc = __fmaf_rn(t0.x, u2, __fmaf_rn(t0.y, u1, sc + t0.z));
        u2 = (__fmaf_rn(t0.w, u1, u2) + t1.x) * t1.z;
        u1 = (u1 + t1.y) * t1.z;
    }
}
// skip8: the first eight rows are Matches in every lane that matters and (sc, u1, u2) is the state after them (lodhi_bound_first_rows4)
template <int CW>
__device__ __forceinline__ float lodhi_bound_mask4(unsigned long long am, unsigned long long bm, int wmax, const bb_lb_entry* tab, bool skip8, float sc, float u1, float u2) {
    const uint32_t a_w[2] = {(uint32_t)am, (uint32_t)(am >> 32)}, b_w[2] = {(uint32_t)bm, (uint32_t)(bm >> 32)};
#pragma unroll
    for (int q = 0; q < CW / 4; ++q) {
        if (4 * q < wmax && !(q < 2 && skip8)) {  // wave-uniform
            const uint32_t idx = ((a_w[q >> 3] >> (4 * (q & 7))) & 0xFu) | (((b_w[q >> 3] >> (4 * (q & 7))) & 0xFu) << 4);
            const float4 t0 = lb_a(tab, idx), t1 = lb_d(tab, idx);
            sc = __fmaf_rn(t0.x, u2, __fmaf_rn(t0.y, u1, sc + t0.z));
            u2 = (__fmaf_rn(t0.w, u1, u2) + t1.x) * t1.z;
            u1 = (u1 + t1.y) * t1.z;
        }
    }
    return sc * (1.0f + 1.0f / 16384.0f);
}
template <int CW>
__device__ __forceinline__ float lodhi_bound_mask(unsigned long long mmask, int wmax, const bb_lb_entry* tab, bool skip0, float sc, float u1, float u2);
template <int CW>
__device__ __forceinline__ float lodhi_bound_tab(unsigned long long plo, unsigned long long phi, int32_t tstart, int32_t best_pos, int wmax,
                                                 const bb_lb_entry* tab) {
    return lodhi_bound_mask<CW>(low64(best_pos) & ~low64(tstart) & ~(plo | phi), wmax, tab, false, 0.0f, 0.0f, 0.0f);   // Match columns (bit c-1)
}
// the state after a first byte of eight Matches
__device__ __forceinline__ void lodhi_bound_first_byte(const bb_lb_entry* tab, float& sc, float& u1, float& u2) {
    const float4 t0 = lb_a(tab, 0xFFu), t1 = lb_d(tab, 0xFFu);
    sc = t0.z; u2 = t1.x * t1.z; u1 = t1.y * t1.z;   // from (0, 0, 0): sc = C, u2 = D S, u1 = E S
}
// skip0: the mask's first byte is 0xFF in every lane that matters (eight granted shared rows) and (sc, u1, u2) is the state after it —
// computed once per block (lodhi_bound_first_byte), not once per barcode
template <int CW>
__device__ __forceinline__ float lodhi_bound_mask(unsigned long long mmask, int wmax, const bb_lb_entry* tab, bool skip0, float sc, float u1, float u2) {
    const uint32_t m_w[2] = {(uint32_t)mmask, (uint32_t)(mmask >> 32)};
#pragma unroll
    for (int q = 0; q < CW / 8; ++q) {
        if (8 * q < wmax && !(q == 0 && skip0)) {  // wave-uniform
            const uint32_t byte = (m_w[q >> 2] >> (8 * (q & 3))) & 0xFFu;
            const float4 t0 = lb_a(tab, byte), t1 = lb_d(tab, byte);
            sc = __fmaf_rn(t0.x, u2, __fmaf_rn(t0.y, u1, sc + t0.z));
            u2 = (__fmaf_rn(t0.w, u1, u2) + t1.x) * t1.z;
            u1 = (u1 + t1.y) * t1.z;
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
    // 1-4 reported positions per lane, visited in ascending order; the two words of the masks one after the other — 32-bit counts and
    // shifts (a dozen instructions per position) instead of 64-bit ones (two dozen)
    const uint32_t Pl = (uint32_t)P, Ml = (uint32_t)M, Ph = (uint32_t)(P >> 32), Mh = (uint32_t)(M >> 32);
    const int32_t tie = tie_last ? 1 : 0;   // tie_last: cq <= best_cost
    for (uint32_t Rl = (uint32_t)R; Rl; Rl &= Rl - 1u) {
        const int q = __ffs((int)Rl) - 1;
        const uint32_t low = (1u << q) - 1u;
        const int32_t cq = m + __popc(Pl & low) - __popc(Ml & low);
        if (cq - tie < best_cost) { best_cost = cq; best_pos = q; }
    }
    const int32_t base = m + __popc(Pl) - __popc(Ml);
    for (uint32_t Rh = (uint32_t)(R >> 32); Rh; Rh &= Rh - 1u) {
        const int q = __ffs((int)Rh) - 1;
        const uint32_t low = (1u << q) - 1u;
        const int32_t cq = base + __popc(Ph & low) - __popc(Mh & low);
        if (cq - tie < best_cost) { best_cost = cq; best_pos = 32 + q; }
    }
    if (pol_lm == BB_LM_PLATEAU_LEFT && best_pos > 0) {
        const unsigned long long ch = (P | M) & ((1ull << best_pos) - 1ull);
        best_pos = ch ? 64 - clz64(ch) : 0;
    }
}

// One column of the DP on the leading shared rows (k_bar_prefix's step: P <= 16 rows in one word, no carry-in — row 0 is the text's free start):
// the horizontal deltas of row P (-> the lane rows' carry-in) and the column's move planes, row r <-> bit P - r, lo | hi << 16.
// PRIO: the traceback order's class (bb_prio.h), or BB_PRIO_RT with the order in `prio`.
template <uint32_t PRIO>
__device__ __forceinline__ void shared_rows_column(uint32_t prio, uint32_t eq, int P, uint32_t& pv, uint32_t& mv, uint32_t& hp, uint32_t& hm, uint32_t& shw) {
    const uint32_t x = eq & pv;
    const uint32_t d0 = (((x + pv) ^ pv) | eq | mv);
    const uint32_t ph = mv | ~(d0 | pv), mh = pv & d0;
    hp = (ph >> (P - 1)) & 1u; hm = (mh >> (P - 1)) & 1u;
    const uint32_t phs = shl1_32(ph), mhs = shl1_32(mh);
    pv = mhs | ~(d0 | phs);
    mv = phs & d0;
    uint32_t l, hh;
    move_planes_any<PRIO>(prio, d0, eq, ph, pv, l, hh);
    shw = (__brev(l) >> (32 - P)) | ((__brev(hh) >> (32 - P)) << 16);
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
