// bb_trim.hip — trim/split step (include/barbell_amd_trim.h) on the GPU.
//
// Pipeline for one batch (rows + verdicts + reads + qualities + headers resident in HBM):
//   k_trim_plan<false>  one lane per row; the first row of every passing read restates preprocess_cuts
//                       (trim.rs:127-254) on the read's <= 32 cut entries and counts surviving slices
//   scan                exclusive scan of the counts -> slot of every read's first record
//   k_trim_plan<true>   same walk, writes bb_slice {read, start, end, label key, suffix, flip, rec_len}
//   radix sort          stable LSD sort of the records by label key (bb_scan.h) -> records of one output
//                       file contiguous, read order inside (the order trim_matches writes them in)
//   k_trim_gather + 64-bit scan of rec_len -> out_off of every record, span starts
//   k_trim_render       one wave per record: header bytes, then the read slice and its qualities as
//                       16-byte chunks (aligned stores, unaligned loads); flip = reverse complement
// HBM-bound byte work: algorithmic bytes per record = 2 x (2 L + header) (read once, written once).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/barbell_amd_trim.h"
#include "bb_common.h"
#include "bb_bytes.h"
#include "bb_ctx_view.h"
#include "bb_scan.h"

#define BB_TRIM_MAX_E 32  // cut entries per read (a passing read's cuts come from one pattern)

struct bb_trim_cfg_dev { bb_trim_config c; };

struct bb_trim_state {
    bool set = false;
    bb_trim_config cfg{};
    uint8_t* d_is_flank = nullptr;
    uint32_t* d_part_rank = nullptr;
    uint32_t n_label_ids = 0;
    // work buffers
    uint32_t *d_cnt = nullptr, *d_base = nullptr, *d_sums = nullptr, *d_err = nullptr;
    uint64_t cap_cnt = 0, cap_base = 0, cap_sums = 0;
    bb_slice *d_tmp = nullptr, *d_sorted = nullptr;
    uint32_t *d_k0 = nullptr, *d_k1 = nullptr, *d_v0 = nullptr, *d_v1 = nullptr;
    uint64_t cap_sl = 0;
    uint64_t* d_sums64 = nullptr; uint64_t cap_sums64 = 0;
    void* d_cub = nullptr; uint64_t cap_cub = 0;
    bb_label_span* d_spans = nullptr; uint64_t cap_spans = 0;
    uint32_t* d_nspans = nullptr;
    float last_ms[3] = {0, 0, 0};  // plan+sort, render, total
    hipEvent_t ev[3] = {nullptr, nullptr, nullptr};
    bb_label_span* d_sp_host = nullptr; uint64_t cap_sp_host = 0;  // span staging of the host-pointer variant
    // staging for the host-pointer variant
    void* d_stage[12] = {};
    uint64_t cap_stage[12] = {};
};

namespace {

#define TCHK(v, call)                                                                  \
    do {                                                                               \
        hipError_t e_ = (call);                                                        \
        if (e_ != hipSuccess) {                                                        \
            *(v).last_error = std::string(#call) + ": " + hipGetErrorString(e_);       \
            return BB_E_HIP;                                                           \
        }                                                                              \
    } while (0)

template <typename T>
int tgrow(bb_ctx_view& v, T*& p, uint64_t& cap, uint64_t need) {
    if (need <= cap && p) return BB_OK;
    if (p) TCHK(v, hipFree(p));
    p = nullptr;
    const uint64_t ncap = need + need / 4 + 64;
    TCHK(v, hipMalloc((void**)&p, ncap * sizeof(T)));
    cap = ncap;
    return BB_OK;
}

__device__ __forceinline__ uint32_t row_slot_dev(const bb_group_dev* __restrict__ groups, const bb_row& m) {
    const bb_group_dev& G = groups[m.group_idx];
    return (uint32_t)G.count_off + (m.barcode_idx >= 0 ? (uint32_t)m.barcode_idx : (uint32_t)G.n_seqs);
}
__device__ __forceinline__ uint32_t dec_digits(uint32_t v) { return v >= 10000 ? 5u : v >= 1000 ? 4u : v >= 100 ? 3u : v >= 10 ? 2u : 1u; }

// ------------------------------------------------------------------------------------------------
// k_trim_plan — preprocess_cuts + the bookkeeping of process_read_and_anno for one read per lane.
// ------------------------------------------------------------------------------------------------
template <bool WRITE>
__global__ __launch_bounds__(128) void k_trim_plan(const bb_row* __restrict__ rows, const bb_row_verdict* __restrict__ ver, uint64_t n_rows,
                                                   const bb_group_dev* __restrict__ groups, const uint32_t* __restrict__ label_ids,
                                                   const uint8_t* __restrict__ is_flank, const uint32_t* __restrict__ part_rank,
                                                   bb_trim_config cfg, const uint64_t* __restrict__ offsets,
                                                   const uint64_t* __restrict__ hdr_offsets, const uint32_t* __restrict__ id_len,
                                                   const uint32_t* __restrict__ desc_start, uint32_t n_reads, uint32_t* __restrict__ cnt,
                                                   const uint32_t* __restrict__ base, bb_slice* __restrict__ out, uint32_t* __restrict__ keys,
                                                   uint8_t* __restrict__ status, uint32_t* __restrict__ err) {
    const uint64_t t = (uint64_t)blockIdx.x * 128u + threadIdx.x;
    if (t >= n_rows) return;
    if (!WRITE) cnt[t] = 0;
    const uint32_t read = rows[t].read_idx;
    if (t > 0 && rows[t - 1].read_idx == read) return;  // not the first row of its read
    if (!ver[t].pass) return;                           // filtered.tsv holds passing reads only
    if (read >= n_reads) { atomicOr(err, 1u); return; }
    uint64_t j = t + 1;
    while (j < n_rows && rows[j].read_idx == read) ++j;
    const uint32_t n = (uint32_t)(j - t);
    const uint32_t seq_len = (uint32_t)(offsets[read + 1] - offsets[read]);

    // cut entries in row order: (start_flank, end_flank, cut, anno) of trim.rs:133-141
    uint32_t e_gid[BB_TRIM_MAX_E], e_start[BB_TRIM_MAX_E], e_end[BB_TRIM_MAX_E];
    uint16_t e_row[BB_TRIM_MAX_E];
    uint32_t after_mask = 0, n_e = 0;
    bool overflow = false;
    for (uint32_t r = 0; r < n && !overflow; ++r) {
        const bb_row_verdict v = ver[t + r];
        for (uint32_t q = 0; q < v.n_cuts && q < BB_MAX_CUTS; ++q) {
            if (n_e == BB_TRIM_MAX_E) { overflow = true; break; }
            e_gid[n_e] = v.cuts[q].group_id; e_start[n_e] = rows[t + r].read_start_flank; e_end[n_e] = rows[t + r].read_end_flank;
            // rows that do not belong to this read (an annotation file of other reads with the same ids: the reference panics on seq[start..end])
            if (e_end[n_e] > seq_len || e_start[n_e] > e_end[n_e]) { atomicOr(err, 4u); atomicMin(err + 1, read); return; }
            e_row[n_e] = (uint16_t)r;
            if (v.cuts[q].direction == BB_CUT_AFTER) after_mask |= 1u << n_e;
            ++n_e;
        }
    }
    if (overflow) { atomicOr(err, 2u); return; }
    // groups by id, in first-appearance order, then stably sorted by the start of their first entry
    // (trim.rs:145-152; equal starts keep first-appearance order — hazard H10, see oracle/README.md)
    uint8_t g_first[BB_TRIM_MAX_E], g_n[BB_TRIM_MAX_E], ord[BB_TRIM_MAX_E];
    uint32_t n_g = 0;
    for (uint32_t i = 0; i < n_e; ++i) {
        uint32_t g = 0;
        while (g < n_g && e_gid[g_first[g]] != e_gid[i]) ++g;
        if (g == n_g) { g_first[n_g] = (uint8_t)i; g_n[n_g] = 0; ++n_g; }
        g_n[g]++;
    }
    for (uint32_t i = 0; i < n_g; ++i) {
        uint32_t k = i;
        while (k > 0 && e_start[g_first[ord[k - 1]]] > e_start[g_first[i]]) { ord[k] = ord[k - 1]; --k; }
        ord[k] = (uint8_t)i;
    }
    uint32_t written = 0, q_slice = 0;
    const uint32_t slot0 = WRITE ? base[t] : 0u;
    for (uint32_t i = 0; i < n_g; ++i) {
        const uint32_t g = ord[i], m0 = g_first[g];
        uint32_t s_start, s_end;
        int32_t a0 = -1, a1 = -1;
        if (g_n[g] == 2) {  // trim.rs:156-181
            uint32_t m1 = m0 + 1;
            while (e_gid[m1] != e_gid[m0]) ++m1;
            s_start = (after_mask >> m0 & 1u) ? e_end[m0] : e_start[m0];
            s_end = (after_mask >> m1 & 1u) ? e_end[m1] : e_start[m1];
            a0 = e_row[m0]; a1 = e_row[m1];
        } else if (g_n[g] == 1) {
            if (!(after_mask >> m0 & 1u)) {  // Before: look left (trim.rs:186-215), max_by_key = last maximum
                s_start = 0; s_end = e_start[m0];
                if (i > 0) {
                    const uint32_t pg = e_gid[g_first[ord[i - 1]]];
                    int32_t best = -1;
                    for (uint32_t k = 0; k < n_e; ++k)
                        if (e_gid[k] == pg && (best < 0 || e_end[k] >= e_end[best])) best = (int32_t)k;
                    s_start = e_end[best]; a0 = e_row[best];
                }
                if (a0 < 0) a0 = e_row[m0]; else a1 = e_row[m0];
            } else {                        // After: look right (trim.rs:217-248), min_by_key = first minimum
                s_start = e_end[m0]; s_end = seq_len;
                a0 = e_row[m0];
                if (i + 1 < n_g) {
                    const uint32_t ng = e_gid[g_first[ord[i + 1]]];
                    int32_t best = -1;
                    for (uint32_t k = 0; k < n_e; ++k)
                        if (e_gid[k] == ng && (best < 0 || e_start[k] < e_start[best])) best = (int32_t)k;
                    s_end = e_start[best]; a1 = e_row[best];
                }
            }
        } else continue;  // three or more cuts in one group: no slice (trim.rs:154-250 has no branch for it)
        const uint32_t q = q_slice++;  // slice_count counts skipped slices too (trim.rs:270-273)
        if (s_start >= s_end) continue;
        if (WRITE) {
            // LabelConfig::create_label (trim.rs:58-105) as a key
            uint32_t key = 0;
            uint8_t flip = 0;
            uint32_t parts[2], np = 0;
            for (int a = 0; a < 2; ++a) {
                const int32_t ra = a ? a1 : a0;
                if (ra < 0) continue;
                const bb_row m = rows[t + (uint32_t)ra];
                if (cfg.flip && m.match_type == BB_FTAG && m.strand == BB_RC) flip = 1;  // should_flip trim.rs:310-315
                const uint32_t id = label_ids[row_slot_dev(groups, m)];
                if (!cfg.add_flank && is_flank[id]) continue;
                parts[np++] = id * 2u + (cfg.add_orientation ? (uint32_t)(m.strand == BB_RC) : 0u);
            }
            if (cfg.add_labels && np) {
                if (cfg.sort_labels) {
                    if (np == 2 && part_rank[parts[1]] < part_rank[parts[0]]) { const uint32_t x = parts[0]; parts[0] = parts[1]; parts[1] = x; }
                } else if (cfg.only_side != BB_SIDE_NONE) {
                    parts[0] = cfg.only_side == BB_SIDE_LEFT ? parts[0] : parts[np - 1];
                    np = 1;
                }
                key = (parts[0] + 1u) << 16 | (np == 2 ? parts[1] + 1u : 0u);
            }
            const uint32_t hl = (uint32_t)(hdr_offsets[read + 1] - hdr_offsets[read]);
            const uint32_t desc_len = hl - desc_start[read];
            const uint32_t L = cfg.skip_trim ? seq_len : s_end - s_start;
            bb_slice o;
            o.read_idx = read; o.start = s_start; o.end = s_end; o.label_key = key; o.suffix = (uint16_t)q; o.flip = flip; o._pad = 0;
            o.rec_len = 1u + id_len[read] + (q ? 1u + dec_digits(q) : 0u) + ((cfg.write_full_header && desc_len) ? 1u + desc_len : 0u) + 1u + L + 3u + L + 1u;
            o.out_off = 0;
            out[slot0 + written] = o;
            keys[slot0 + written] = key;
        }
        ++written;
    }
    if (!WRITE) {
        cnt[t] = written;
        status[read] = written ? BB_TRIM_TRIMMED : BB_TRIM_FAILED;
    }
}

__global__ __launch_bounds__(256) void k_iota(uint32_t* __restrict__ v, uint32_t n) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i < n) v[i] = i;
}

// records in sorted order + per-block sums of rec_len (64-bit), 1024 records per block
__global__ __launch_bounds__(256) void k_trim_gather(const bb_slice* __restrict__ tmp, const uint32_t* __restrict__ perm, uint32_t n,
                                                     bb_slice* __restrict__ sorted, uint64_t* __restrict__ sums) {
    __shared__ uint64_t s_w[4];
    const uint32_t b0 = blockIdx.x * 1024u + threadIdx.x * 4u;
    uint64_t t = 0, pre[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        pre[i] = t;
        if (b0 + i < n) {
            const uint4* src = (const uint4*)&tmp[perm[b0 + i]];
            uint4 lo = src[0], hi = src[1];
            t += hi.y;  // rec_len: second word of the upper half
            ((uint4*)&sorted[b0 + i])[0] = lo;
            ((uint4*)&sorted[b0 + i])[1] = hi;
        }
    }
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    uint64_t inc = t;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const uint64_t y = __shfl_up(inc, d, 64); if (lane >= d) inc += y; }
    if (lane == 63) s_w[wv] = inc;
    __syncthreads();
    uint64_t wbase = 0;
    for (int i = 0; i < wv; ++i) wbase += s_w[i];
    const uint64_t excl = wbase + inc - t;
#pragma unroll
    for (int i = 0; i < 4; ++i) if (b0 + i < n) sorted[b0 + i].out_off = excl + pre[i];  // block-local for now
    if (threadIdx.x == 255) sums[blockIdx.x] = wbase + inc;
}
// adds the block bases to out_off and records the span starts (label changes)
__global__ __launch_bounds__(256) void k_trim_offsets(bb_slice* __restrict__ sorted, uint32_t n, const uint64_t* __restrict__ sums,
                                                      bb_label_span* __restrict__ spans, uint32_t spans_cap, uint32_t* __restrict__ n_spans) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const uint64_t off = sorted[i].out_off + sums[i >> 10];
    sorted[i].out_off = off;
    const uint32_t key = sorted[i].label_key;
    if (i == 0 || sorted[i - 1].label_key != key) {
        const uint32_t s = atomicAdd(n_spans, 1u);
        if (s < spans_cap) { bb_label_span sp; sp.label_key = key; sp.n_records = 0; sp.first = i; sp.off = off; sp.len = 0; spans[s] = sp; }
    }
}

// ------------------------------------------------------------------------------------------------
// k_trim_render — one wave per record.
// ------------------------------------------------------------------------------------------------
// complement of trim.rs:486-530: A<->T C<->G R<->Y K<->M B<->V D<->H in both cases, everything else fixed
__device__ __forceinline__ uint8_t comp_char(uint8_t ch) {
    const uint8_t up = ch & 0xDFu;  // fold case for letters
    if (up < 'A' || up > 'Z' || (ch & 0xC0u) != 0x40u) return ch;
    uint8_t r;
    switch (up) {
        case 'A': r = 'T'; break; case 'T': r = 'A'; break; case 'C': r = 'G'; break; case 'G': r = 'C'; break;
        case 'R': r = 'Y'; break; case 'Y': r = 'R'; break; case 'K': r = 'M'; break; case 'M': r = 'K'; break;
        case 'B': r = 'V'; break; case 'V': r = 'B'; break; case 'D': r = 'H'; break; case 'H': r = 'D'; break;
        default: return ch;
    }
    return (uint8_t)(r | (ch & 0x20u));
}

template <bool COMP>
__device__ __forceinline__ void wave_copy_rev(uint8_t* __restrict__ dst, const uint8_t* __restrict__ src, uint32_t L, int lane) {
    for (uint32_t k = (uint32_t)lane; k < L; k += 64u) {
        const uint8_t ch = src[L - 1u - k];
        dst[k] = COMP ? comp_char(ch) : ch;
    }
}

__global__ __launch_bounds__(256) void k_trim_render(const bb_slice* __restrict__ slices, uint32_t n, const uint8_t* __restrict__ bases,
                                                     const uint8_t* __restrict__ quals, const uint64_t* __restrict__ offsets,
                                                     const uint8_t* __restrict__ hdr, const uint64_t* __restrict__ hdr_offsets,
                                                     const uint32_t* __restrict__ id_len, const uint32_t* __restrict__ desc_start,
                                                     bb_trim_config cfg, uint8_t* __restrict__ text) {
    const int lane = threadIdx.x & 63;
    const uint32_t rec = blockIdx.x * 4u + (threadIdx.x >> 6);
    if (rec >= n) return;
    const bb_slice s = slices[rec];
    const uint32_t read = s.read_idx;
    const uint64_t h0 = hdr_offsets[read];
    const uint32_t hl = (uint32_t)(hdr_offsets[read + 1] - h0), idl = id_len[read], ds = desc_start[read];
    const uint32_t sfx = s.suffix ? 1u + dec_digits(s.suffix) : 0u;
    const uint32_t dl = (cfg.write_full_header && hl > ds) ? hl - ds : 0u;
    const uint32_t hlen = 1u + idl + sfx + (dl ? 1u + dl : 0u) + 1u;  // '@' id suffix [' ' desc] '\n'
    uint8_t* w = text + s.out_off;
    for (uint32_t p = (uint32_t)lane; p < hlen; p += 64u) {
        uint8_t ch;
        if (p == 0) ch = '@';
        else if (p <= idl) ch = hdr[h0 + p - 1u];
        else if (p <= idl + sfx) {
            const uint32_t k = p - idl - 1u;  // 0 = '_', then the digits, most significant first
            if (k == 0) ch = '_';
            else {
                uint32_t v = s.suffix;
                for (uint32_t z = sfx - 1u - k; z > 0; --z) v /= 10u;
                ch = (uint8_t)('0' + v % 10u);
            }
        } else if (p == hlen - 1u) ch = '\n';
        else if (p == idl + sfx + 1u) ch = ' ';
        else ch = hdr[h0 + ds + (p - idl - sfx - 2u)];
        w[p] = ch;
    }
    const uint64_t b0 = offsets[read];
    const uint32_t seq_len = (uint32_t)(offsets[read + 1] - b0);
    const uint32_t s0 = cfg.skip_trim ? 0u : s.start, L = (cfg.skip_trim ? seq_len : s.end) - s0;
    uint8_t* wseq = w + hlen;
    uint8_t* wq = wseq + L + 3u;
    if (lane < 3) wseq[L + lane] = lane == 1 ? '+' : '\n';
    if (lane == 3) wq[L] = '\n';
    if (!s.flip) {
        wave_copy(wseq, bases + b0 + s0, L, lane);
        wave_copy(wq, quals + b0 + s0, L, lane);
    } else {
        wave_copy_rev<true>(wseq, bases + b0 + s0, L, lane);
        wave_copy_rev<false>(wq, quals + b0 + s0, L, lane);
    }
}

int ensure_state(bb_ctx_view& v) {
    if (!*v.trim) *v.trim = new bb_trim_state();
    return BB_OK;
}

}  // namespace

void bb_trim_state_free(bb_trim_state* s) {
    if (!s) return;
    for (void* p : {(void*)s->d_is_flank, (void*)s->d_part_rank, (void*)s->d_cnt, (void*)s->d_base, (void*)s->d_sums, (void*)s->d_err,
                    (void*)s->d_tmp, (void*)s->d_sorted, (void*)s->d_k0, (void*)s->d_k1, (void*)s->d_v0, (void*)s->d_v1, (void*)s->d_sums64,
                    s->d_cub, (void*)s->d_spans, (void*)s->d_nspans})
        if (p) (void)hipFree(p);
    for (void* p : s->d_stage) if (p) (void)hipFree(p);
    if (s->d_sp_host) (void)hipFree(s->d_sp_host);
    for (auto& e : s->ev) if (e) (void)hipEventDestroy(e);
    delete s;
}

extern "C" int bb_trim_set(bb_ctx* ctx, const bb_trim_config* cfg, const uint8_t* label_is_flank, const uint32_t* part_rank,
                           uint32_t n_label_ids) {
    if (!ctx || !cfg || !label_is_flank || !part_rank || n_label_ids == 0) return BB_E_INVALID;
    bb_ctx_view v = bb_ctx_get_view(ctx);
    if (cfg->sort_labels && cfg->only_side != BB_SIDE_NONE) {  // trim.rs:330-334
        *v.last_error = "Cannot enable only keeping left/right label and sorting; this is ambiguous";
        return BB_E_INVALID;
    }
    if (cfg->only_side > BB_SIDE_RIGHT || n_label_ids > 32767) return BB_E_UNSUPPORTED;
    TCHK(v, hipSetDevice(v.device));
    ensure_state(v);
    bb_trim_state* s = *v.trim;
    if (s->d_is_flank) (void)hipFree(s->d_is_flank);
    if (s->d_part_rank) (void)hipFree(s->d_part_rank);
    s->d_is_flank = nullptr; s->d_part_rank = nullptr;
    TCHK(v, hipMalloc((void**)&s->d_is_flank, n_label_ids));
    TCHK(v, hipMalloc((void**)&s->d_part_rank, sizeof(uint32_t) * 2 * n_label_ids));
    TCHK(v, hipMemcpy(s->d_is_flank, label_is_flank, n_label_ids, hipMemcpyHostToDevice));
    TCHK(v, hipMemcpy(s->d_part_rank, part_rank, sizeof(uint32_t) * 2 * n_label_ids, hipMemcpyHostToDevice));
    if (!s->d_err) TCHK(v, hipMalloc((void**)&s->d_err, 4 * sizeof(uint32_t)));
    if (!s->d_nspans) TCHK(v, hipMalloc((void**)&s->d_nspans, 4 * sizeof(uint64_t)));
    s->cfg = *cfg;
    s->n_label_ids = n_label_ids;
    s->set = true;
    return BB_OK;
}

// plan_only: everything but the record text — slices in text order with their offsets and lengths, spans, statuses
static int trim_impl(bb_ctx* ctx, const bb_row* d_rows, const bb_row_verdict* d_ver, uint64_t n_rows, const uint8_t* d_bases,
                     const uint8_t* d_quals, const uint64_t* d_offsets, const bb_headers* h, uint32_t n_reads, uint8_t* d_text,
                     uint64_t text_cap, uint64_t* text_len, bb_slice* d_slices, uint64_t slices_cap, uint64_t* n_slices,
                     bb_label_span* d_spans, uint32_t spans_cap, uint32_t* n_spans, uint8_t* d_status, bool plan_only) {
    if (!ctx || !text_len || !n_slices || !n_spans || !h || (n_reads && (!d_offsets || !d_status)) || (n_rows && (!d_rows || !d_ver)))
        return BB_E_INVALID;
    bb_ctx_view v = bb_ctx_get_view(ctx);
    bb_trim_state* s = *v.trim;
    if (!s || !s->set) { *v.last_error = "bb_trim_set has not been called"; return BB_E_INVALID; }
    if (!v.d_label_ids) { *v.last_error = "bb_filter_set has not been called (label ids)"; return BB_E_INVALID; }
    *text_len = 0; *n_slices = 0; *n_spans = 0;
    TCHK(v, hipSetDevice(v.device));
    hipStream_t st = v.stream;
    hipEvent_t* ev = s->ev;
    if (!ev[0]) for (int i = 0; i < 3; ++i) TCHK(v, hipEventCreate(&ev[i]));
    TCHK(v, hipEventRecord(ev[0], st));
    if (n_reads) TCHK(v, hipMemsetAsync(d_status, BB_TRIM_NONE, n_reads, st));
    uint64_t ns = 0, tl = 0;
    uint32_t nsp = 0;
    int r;
    if (n_rows) {
        if ((r = tgrow(v, s->d_cnt, s->cap_cnt, n_rows))) return r;
        if ((r = tgrow(v, s->d_base, s->cap_base, n_rows))) return r;
        const uint32_t nb = (uint32_t)((n_rows + 2047) / 2048);
        if ((r = tgrow(v, s->d_sums, s->cap_sums, (uint64_t)nb + 1))) return r;
        TCHK(v, hipMemsetAsync(s->d_err, 0, 4 * sizeof(uint32_t), st));
        TCHK(v, hipMemsetAsync(s->d_err + 1, 0xFF, sizeof(uint32_t), st));   // the first read whose rows lie beyond its end (atomicMin)
        const dim3 pg((unsigned)((n_rows + 127) / 128));
        hipLaunchKernelGGL(k_trim_plan<false>, pg, dim3(128), 0, st, d_rows, d_ver, n_rows, v.d_groups, v.d_label_ids,
                           (const uint8_t*)s->d_is_flank, (const uint32_t*)s->d_part_rank, s->cfg, d_offsets, h->hdr_offsets, h->id_len,
                           h->desc_start, n_reads, s->d_cnt, (const uint32_t*)nullptr, (bb_slice*)nullptr, (uint32_t*)nullptr, d_status, s->d_err);
        TCHK(v, bb_scan32(st, (const uint32_t*)s->d_cnt, s->d_base, n_rows, s->d_sums, s->d_err + 2));
        uint32_t herr[4];
        TCHK(v, hipMemcpyAsync(herr, s->d_err, sizeof(herr), hipMemcpyDeviceToHost, st));
        TCHK(v, hipStreamSynchronize(st));
        if (herr[0] & 1u) { *v.last_error = "row with read_idx >= n_reads"; return BB_E_INVALID; }
        if (herr[0] & 2u) { *v.last_error = "more than 32 cuts on one read"; return BB_E_UNSUPPORTED; }
        if (herr[0] & 4u) { *v.last_error = "read " + std::to_string(herr[1]) + ": a row with a cut lies beyond the read's end (rows of other reads?)"; return BB_E_INVALID; }
        ns = herr[2];
    }
    if (ns) {
        const uint32_t n = (uint32_t)ns;
        if (ns > s->cap_sl || !s->d_tmp) {
            for (void** p : {(void**)&s->d_tmp, (void**)&s->d_sorted, (void**)&s->d_k0, (void**)&s->d_k1, (void**)&s->d_v0, (void**)&s->d_v1})
                if (*p) { (void)hipFree(*p); *p = nullptr; }
            const uint64_t cap = ns + ns / 4 + 64;
            TCHK(v, hipMalloc((void**)&s->d_tmp, cap * sizeof(bb_slice)));
            TCHK(v, hipMalloc((void**)&s->d_sorted, cap * sizeof(bb_slice)));
            for (uint32_t** p : {&s->d_k0, &s->d_k1, &s->d_v0, &s->d_v1}) TCHK(v, hipMalloc((void**)p, cap * sizeof(uint32_t)));
            s->cap_sl = cap;
        }
        const dim3 pg((unsigned)((n_rows + 127) / 128));
        hipLaunchKernelGGL(k_trim_plan<true>, pg, dim3(128), 0, st, d_rows, d_ver, n_rows, v.d_groups, v.d_label_ids,
                           (const uint8_t*)s->d_is_flank, (const uint32_t*)s->d_part_rank, s->cfg, d_offsets, h->hdr_offsets, h->id_len,
                           h->desc_start, n_reads, s->d_cnt, (const uint32_t*)s->d_base, s->d_tmp, s->d_k0, d_status, s->d_err);
        hipLaunchKernelGGL(k_iota, dim3((n + 255) / 256), dim3(256), 0, st, s->d_v0, n);
        // stable sort of (label key, record index): the key is (part0 + 1) << 16 | (part1 + 1) with part < 2 * n_label_ids, so only
        // the low digits of its two 16-bit fields can be occupied (SQK-NBD114-96: one 8-bit digit each, two passes)
        uint32_t *k_sorted = nullptr, *v_sorted = nullptr;
        {
            uint32_t shifts[4];
            int n_shifts = 0;
            const uint32_t field_max = 2u * s->n_label_ids + 1u;
            for (uint32_t f = 0; f < 2; ++f)
                for (uint32_t sh = 0; sh < 16u && (field_max >> sh) != 0u; sh += 8u) shifts[n_shifts++] = 16u * f + sh;
            const uint64_t ntiles = (n + BB_RS_TILE - 1) / BB_RS_TILE, hist_words = 256 * ntiles + 2 + (256 * ntiles) / 2048 + 4;
            if (hist_words * 4 > s->cap_cub || !s->d_cub) {
                if (s->d_cub) (void)hipFree(s->d_cub);
                s->d_cub = nullptr;
                TCHK(v, hipMalloc(&s->d_cub, hist_words * 4 + 256));
                s->cap_cub = hist_words * 4 + 256;
            }
            uint32_t* d_hist = (uint32_t*)s->d_cub;
            TCHK(v, bb_radix_sort_pairs(st, s->d_k0, s->d_v0, s->d_k1, s->d_v1, n, shifts, n_shifts, d_hist, d_hist + 256 * ntiles + 2, &k_sorted, &v_sorted));
        }
        const uint32_t gb = (n + 1023) / 1024;
        if ((r = tgrow(v, s->d_sums64, s->cap_sums64, (uint64_t)gb + 1))) return r;
        const uint64_t span_cap_int = std::max<uint64_t>(spans_cap, 65536);
        if ((r = tgrow(v, s->d_spans, s->cap_spans, span_cap_int))) return r;
        TCHK(v, hipMemsetAsync(s->d_nspans, 0, 4 * sizeof(uint64_t), st));
        hipLaunchKernelGGL(k_trim_gather, dim3(gb), dim3(256), 0, st, (const bb_slice*)s->d_tmp, (const uint32_t*)v_sorted, n, s->d_sorted, s->d_sums64);
        hipLaunchKernelGGL(k_scan_sums_t<uint64_t>, dim3(1), dim3(64), 0, st, s->d_sums64, gb, (uint64_t*)s->d_nspans + 1);
        hipLaunchKernelGGL(k_trim_offsets, dim3((n + 255) / 256), dim3(256), 0, st, s->d_sorted, n, (const uint64_t*)s->d_sums64, s->d_spans,
                           (uint32_t)s->cap_spans, s->d_nspans);
        TCHK(v, hipGetLastError());
        uint64_t hs[2];
        TCHK(v, hipMemcpyAsync(hs, s->d_nspans, sizeof(hs), hipMemcpyDeviceToHost, st));
        TCHK(v, hipStreamSynchronize(st));
        nsp = (uint32_t)hs[0];
        tl = hs[1];
        if (nsp > s->cap_spans) { *v.last_error = "more than 65536 output labels in one batch"; return BB_E_UNSUPPORTED; }
    }
    TCHK(v, hipEventRecord(ev[1], st));
    *text_len = tl; *n_slices = ns; *n_spans = nsp;
    if ((!plan_only && (tl > text_cap || (tl && !d_text))) || ns > slices_cap || nsp > spans_cap || (ns && !d_slices) || (nsp && !d_spans)) return BB_E_CAPACITY;
    if (ns) {
        const uint32_t n = (uint32_t)ns;
        // spans: few (one per output label) -> ordered and completed on the host
        std::vector<bb_label_span> sp(nsp);
        TCHK(v, hipMemcpy(sp.data(), s->d_spans, sizeof(bb_label_span) * nsp, hipMemcpyDeviceToHost));
        std::sort(sp.begin(), sp.end(), [](const bb_label_span& a, const bb_label_span& b) { return a.first < b.first; });
        for (uint32_t i = 0; i < nsp; ++i) {
            const uint64_t nf = i + 1 < nsp ? sp[i + 1].first : ns, no = i + 1 < nsp ? sp[i + 1].off : tl;
            sp[i].n_records = (uint32_t)(nf - sp[i].first);
            sp[i].len = no - sp[i].off;
        }
        TCHK(v, hipMemcpyAsync(d_spans, sp.data(), sizeof(bb_label_span) * nsp, hipMemcpyHostToDevice, st));
        TCHK(v, hipMemcpyAsync(d_slices, s->d_sorted, sizeof(bb_slice) * ns, hipMemcpyDeviceToDevice, st));
        if (!plan_only) {
            if (!d_bases || !d_quals) { *v.last_error = "bb_trim_batch_dev needs bases and qualities (a two-line FASTQ block has none: bb_trim_plan_dev)"; return BB_E_INVALID; }
            hipLaunchKernelGGL(k_trim_render, dim3((n + 3) / 4), dim3(256), 0, st, (const bb_slice*)s->d_sorted, n, d_bases, d_quals, d_offsets, h->hdr,
                               h->hdr_offsets, h->id_len, h->desc_start, s->cfg, d_text);
        }
        TCHK(v, hipGetLastError());
    }
    TCHK(v, hipEventRecord(ev[2], st));
    TCHK(v, hipStreamSynchronize(st));
    (void)hipEventElapsedTime(&s->last_ms[0], ev[0], ev[1]);
    (void)hipEventElapsedTime(&s->last_ms[1], ev[1], ev[2]);
    s->last_ms[2] = s->last_ms[0] + s->last_ms[1];
    return BB_OK;
}

extern "C" int bb_trim_batch_dev(bb_ctx* ctx, const bb_row* d_rows, const bb_row_verdict* d_ver, uint64_t n_rows, const uint8_t* d_bases,
                                 const uint8_t* d_quals, const uint64_t* d_offsets, const bb_headers* h, uint32_t n_reads, uint8_t* d_text,
                                 uint64_t text_cap, uint64_t* text_len, bb_slice* d_slices, uint64_t slices_cap, uint64_t* n_slices,
                                 bb_label_span* d_spans, uint32_t spans_cap, uint32_t* n_spans, uint8_t* d_status) {
    return trim_impl(ctx, d_rows, d_ver, n_rows, d_bases, d_quals, d_offsets, h, n_reads, d_text, text_cap, text_len, d_slices, slices_cap, n_slices,
                     d_spans, spans_cap, n_spans, d_status, false);
}

extern "C" int bb_trim_plan_dev(bb_ctx* ctx, const bb_row* d_rows, const bb_row_verdict* d_ver, uint64_t n_rows, const uint64_t* d_offsets,
                                const bb_headers* h, uint32_t n_reads, uint64_t* text_len, bb_slice* d_slices, uint64_t slices_cap,
                                uint64_t* n_slices, bb_label_span* d_spans, uint32_t spans_cap, uint32_t* n_spans, uint8_t* d_status) {
    return trim_impl(ctx, d_rows, d_ver, n_rows, nullptr, nullptr, d_offsets, h, n_reads, nullptr, 0, text_len, d_slices, slices_cap, n_slices,
                     d_spans, spans_cap, n_spans, d_status, true);
}

extern "C" int bb_trim_batch(bb_ctx* ctx, const bb_row* rows, const bb_row_verdict* ver, uint64_t n_rows, const uint8_t* bases,
                             const uint8_t* quals, const uint64_t* offsets, const bb_headers* h, uint32_t n_reads, uint8_t* text,
                             uint64_t text_cap, uint64_t* text_len, bb_slice* slices, uint64_t slices_cap, uint64_t* n_slices,
                             bb_label_span* spans, uint32_t spans_cap, uint32_t* n_spans, uint8_t* read_status) {
    if (!ctx || !text_len || !n_slices || !n_spans || !h || (n_reads && (!offsets || !read_status || !h->hdr_offsets || !h->id_len || !h->desc_start)) ||
        (n_rows && (!rows || !ver)))
        return BB_E_INVALID;
    bb_ctx_view v = bb_ctx_get_view(ctx);
    bb_trim_state* s = *v.trim;
    if (!s || !s->set) { *v.last_error = "bb_trim_set has not been called"; return BB_E_INVALID; }
    TCHK(v, hipSetDevice(v.device));
    const uint64_t n_bases = n_reads ? offsets[n_reads] : 0, n_hdr = n_reads ? h->hdr_offsets[n_reads] : 0;
    // stage: 0 rows 1 verdicts 2 bases 3 quals 4 offsets 5 hdr 6 hdr_offsets 7 id_len 8 desc_start 9 status 10 text 11 slices
    const void* src[9] = {rows, ver, bases, quals, offsets, h->hdr, h->hdr_offsets, h->id_len, h->desc_start};
    const uint64_t bytes[12] = {n_rows * sizeof(bb_row), n_rows * sizeof(bb_row_verdict), n_bases, n_bases, ((uint64_t)n_reads + 1) * 8, n_hdr,
                                ((uint64_t)n_reads + 1) * 8, (uint64_t)n_reads * 4, (uint64_t)n_reads * 4, n_reads, text_cap,
                                slices_cap * sizeof(bb_slice)};
    int r;
    for (int i = 0; i < 12; ++i) {
        uint8_t*& p = (uint8_t*&)s->d_stage[i];
        if ((r = tgrow(v, p, s->cap_stage[i], bytes[i] + 16))) return r;
        if (i < 9 && bytes[i]) {
            if (!src[i]) return BB_E_INVALID;
            TCHK(v, hipMemcpyAsync(p, src[i], bytes[i], hipMemcpyHostToDevice, v.stream));
        }
    }
    if ((r = tgrow(v, s->d_sp_host, s->cap_sp_host, (uint64_t)spans_cap + 1))) return r;
    bb_label_span* d_sp = s->d_sp_host;
    bb_headers dh{(const uint8_t*)s->d_stage[5], (const uint64_t*)s->d_stage[6], (const uint32_t*)s->d_stage[7], (const uint32_t*)s->d_stage[8]};
    r = bb_trim_batch_dev(ctx, (const bb_row*)s->d_stage[0], (const bb_row_verdict*)s->d_stage[1], n_rows, (const uint8_t*)s->d_stage[2],
                          (const uint8_t*)s->d_stage[3], (const uint64_t*)s->d_stage[4], &dh, n_reads, (uint8_t*)s->d_stage[10], text_cap, text_len,
                          (bb_slice*)s->d_stage[11], slices_cap, n_slices, d_sp, spans_cap, n_spans, (uint8_t*)s->d_stage[9]);
    if (r == BB_OK || r == BB_E_CAPACITY) {
        if (n_reads) (void)hipMemcpy(read_status, s->d_stage[9], n_reads, hipMemcpyDeviceToHost);
    }
    if (r == BB_OK) {
        if (*text_len) TCHK(v, hipMemcpy(text, s->d_stage[10], *text_len, hipMemcpyDeviceToHost));
        if (*n_slices) TCHK(v, hipMemcpy(slices, s->d_stage[11], *n_slices * sizeof(bb_slice), hipMemcpyDeviceToHost));
        if (*n_spans) TCHK(v, hipMemcpy(spans, d_sp, (uint64_t)*n_spans * sizeof(bb_label_span), hipMemcpyDeviceToHost));
    }
    return r;
}

extern "C" float bb_trim_last_ms(bb_ctx* ctx, int which) {
    if (!ctx || which < 0 || which > 2) return 0.f;
    bb_ctx_view v = bb_ctx_get_view(ctx);
    return *v.trim ? (*v.trim)->last_ms[which] : 0.f;
}
