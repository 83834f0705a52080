// bb_k_rows.h — from provisional rows to the output: rows_decide / k_rows (exact score of the best-bounded path, decision),
// k_hit_lists, k_collapse (interval.rs:4-79), k_emit (compaction in read order + histogram).
#pragma once
#include "bb_k_bar_common.h"

__global__ __launch_bounds__(256) void k_rows(const bb_group_dev* __restrict__ groups, const bb_hit* __restrict__ hits, uint32_t n_hits,
                                              bb_rowtmp* __restrict__ rows, double min_score, double min_score_diff, double margin,
                                              uint32_t* __restrict__ fb_lists, uint32_t list_stride, uint32_t* __restrict__ fb_cnt,
                                              const uint32_t* __restrict__ n_hits_dev) {
    BB_HITS_ON_DEVICE(n_hits, n_hits_dev, 256u);
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
__global__ __launch_bounds__(256) void k_hit_lists(const uint32_t* __restrict__ hit_meta, uint32_t n_hits, bb_rowtmp* __restrict__ rows,
                                                   uint32_t* __restrict__ lists, uint32_t list_stride, uint32_t* __restrict__ list_cnt,
                                                   uint32_t n_groups, const bb_group_dev* __restrict__ groups, const uint32_t* __restrict__ n_hits_dev) {
    BB_HITS_ON_DEVICE(n_hits, n_hits_dev, 256u);
    const uint32_t t = blockIdx.x * 256u + threadIdx.x;
    const bool in = t < n_hits;
    const uint32_t meta = in ? hit_meta[t] : 0u;   // bb_hit_meta: 4 bytes per hit instead of its 96-byte record (0.26 GB per 2 M-read step)
    const bool valid = in && (meta & 0xFFu);
    if (in && !valid) rows[t].row._pad[0] = 0;
    const unsigned lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    const uint32_t my_slot = valid ? (meta >> 8) & 0xFFu : 0xFFFFFFFFu;
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
                                                  uint32_t n_reads, uint32_t n_groups, uint32_t* __restrict__ nrows,
                                                  const uint32_t* __restrict__ n_hits_dev, uint32_t cap_hits, const uint32_t* __restrict__ hit_meta) {
    if (n_hits_dev && *n_hits_dev > cap_hits) return;   // deferred batch whose hits overflowed their buffers: the host runs it again (slots beyond the buffers)
    const uint32_t read = blockIdx.x * 256u + threadIdx.x;
    if (read >= n_reads) return;
    const uint32_t b0 = slot_base[(uint64_t)read * n_groups * 2], b1 = slot_base[(uint64_t)(read + 1) * n_groups * 2];
    if (b0 == b1) { nrows[read] = 0; return; }
    // one hit (the common case): nothing to sort or to collapse, and whether it has a row is its meta word's say (a hit has a row — a barcode's
    // or the flank's alone — unless get_matching_region gave None: k_hit_lists) — 4 bytes instead of the row's line
    if (b1 - b0 == 1u) { nrows[read] = (hit_meta[b0] & 0xFFu) ? 1u : 0u; return; }
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
                                              unsigned long long* __restrict__ counts, uint32_t counts_len,
                                              const uint32_t* __restrict__ n_hits_dev, uint32_t cap_hits, uint64_t rows_cap) {
    // a deferred batch (no round trip between the scans and here): nothing is emitted or counted if the host is going to run the batch again
    // (hits overflowed) or to hand it back (rows beyond the caller's capacity) — it reads the same two numbers after this launch
    if (n_hits_dev && (*n_hits_dev > cap_hits || (uint64_t)row_off[n_reads] > rows_cap)) return;
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

