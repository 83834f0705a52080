// bb_k_misc.h — k_synth (benchmark reads generated in HBM), k_filter (pattern.rs:96-240), k_inspect (inspect.rs:15-117).
#pragma once
#include "bb_myers.h"

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
