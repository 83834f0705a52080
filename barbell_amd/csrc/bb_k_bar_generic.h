// bb_k_bar_generic.h — the barcode kernels outside the row split: k_barcode (any geometry, any policy; move bits in private
// memory) and k_barcode_reg (two-word, register-resident).
#pragma once
#include "bb_k_bar_common.h"

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
    // prefetch of the next iteration's hit records: the lanes of a hit share its six 16-byte pieces: lane p prefetches piece p; groups of fewer
    // than six sequences fetch pieces p + N, p + 2N at the start of the iteration instead, unprefetched (they have many hits per block
    // iteration to hide it behind) — one uint4 of prefetch state instead of three: k_barcode_reg<2, 48> held 256 VGPRs + 4 spilled with three
    uint4 pre = make_uint4(0u, 0u, 0u, 0u);
    auto prefetch = [&](uint32_t it) {
        const uint32_t li = it * hpb + (uint32_t)hl;
        if (in_blk && p < PIECES && it < n_iter && li < n_list) pre = reinterpret_cast<const uint4*>(hits + (hit_list ? hit_list[li] : li))[p];
    };
    prefetch(blockIdx.x);
  for (uint32_t it = blockIdx.x; it < n_iter; it += gridDim.x) {
    const uint32_t li = it * hpb + (uint32_t)hl;
    const bool exists = in_blk && li < n_list;
    if (exists && p < PIECES) {
        s_hit[hl * PIECES + p] = pre;
        if (p + N < PIECES) {
            const uint4* src = reinterpret_cast<const uint4*>(hits + (hit_list ? hit_list[li] : li));
            s_hit[hl * PIECES + p + N] = src[p + N];
            if (p + 2 * N < PIECES) s_hit[hl * PIECES + p + 2 * N] = src[p + 2 * N];
        }
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
        // the window's base-set codes, four columns per word, read from the hit's LDS record as the columns come (the lanes of a hit read
        // the same word: a broadcast) instead of held in CW / 4 registers through the whole pass
        const uint32_t* wcs = reinterpret_cast<const uint32_t*>(s_hit + hls * PIECES + 2);
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
                uint32_t wcw = 0u;
#pragma unroll
                for (int c = c0; c < c0 + BB_CG; ++c) {
                    if ((c & 3) == 0) wcw = wcs[c >> 2];
                    const uint32_t code = (wcw >> (8 * (c & 3))) & 0xFu;
                    uint32_t eq[WB], d0[WB], ph[WB], mh[WB], l[WB], hh[WB];
                    const uint32_t ei = __umul24(code, NW4) + pb4;
                    if constexpr (WB == 2) { uint2 v = *reinterpret_cast<const uint2*>(s_peq_b + ei); eq[0] = v.x; eq[1] = v.y; }
                    else eq[0] = *reinterpret_cast<const uint32_t*>(s_peq_b + ei);
                    myers_step<WB>(pv, mv, eq, d0, ph, mh);
                    move_bits_prio<WB>((uint32_t)G.pol_prio, eq, d0, ph, pv, l, hh);  // pv: the new column's
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

