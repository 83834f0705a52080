// bb_format.hip — annotation.tsv / filtered.tsv lines rendered on the GPU (include/barbell_amd_format.h).
// Byte-exact with the csv-crate serialisation of BarbellMatch (searcher.rs:31-142, annotator.rs:13-26) as restated by
// barbell_amd/annotate.py::format_rows and host/bb_host.cpp::BarbellMatch::to_tsv (tests/test_format.py).
#include <hip/hip_runtime.h>

#include <string>
#include <vector>

#include "../../include/barbell_amd_format.h"
#include "bb_ctx_view.h"
#include "bb_scan.h"

struct bb_format_state {
    uint8_t* d_blob = nullptr;
    uint32_t* d_off = nullptr;
    uint32_t n_slots = 0;
    uint32_t* d_len = nullptr; uint64_t cap_len = 0;
    uint64_t* d_pos = nullptr; uint64_t cap_pos = 0;
    void* d_cub = nullptr; uint64_t cap_cub = 0;
    uint64_t* d_tot = nullptr;
};

namespace {

#define FCHK(v, call)                                                                  \
    do {                                                                               \
        hipError_t e_ = (call);                                                        \
        if (e_ != hipSuccess) {                                                        \
            *(v).last_error = std::string(#call) + ": " + hipGetErrorString(e_);       \
            return BB_E_HIP;                                                           \
        }                                                                              \
    } while (0)

template <typename T>
int fgrow(bb_ctx_view& v, T*& p, uint64_t& cap, uint64_t need) {
    if (need <= cap && p) return BB_OK;
    if (p) FCHK(v, hipFree(p));
    p = nullptr;
    const uint64_t ncap = need + need / 4 + 64;
    FCHK(v, hipMalloc((void**)&p, ncap * sizeof(T)));
    cap = ncap;
    return BB_OK;
}

__device__ __forceinline__ uint32_t n_digits(uint32_t v) {
    uint32_t d = 1;
    while (v >= 10u) { v /= 10u; ++d; }
    return d;
}
__device__ __forceinline__ uint8_t* put_u32(uint8_t* p, uint32_t v) {
    const uint32_t d = n_digits(v);
    for (uint32_t i = d; i-- > 0;) { p[i] = (uint8_t)('0' + v % 10u); v /= 10u; }
    return p + d;
}
__device__ __forceinline__ uint8_t* put_str(uint8_t* p, const char* s, uint32_t n) {
    for (uint32_t i = 0; i < n; ++i) p[i] = (uint8_t)s[i];
    return p + n;
}
// "After(g):idx" / "Before(g):idx" joined by ',' (searcher.rs:91-106)
__device__ __forceinline__ uint32_t cuts_len(const bb_row_verdict& v) {
    uint32_t n = 0;
    for (uint32_t q = 0; q < v.n_cuts && q < BB_MAX_CUTS; ++q)
        n += (q ? 1u : 0u) + (v.cuts[q].direction == BB_CUT_AFTER ? 6u : 7u) + n_digits(v.cuts[q].group_id) + 2u + n_digits(v.match_idx);
    return n;
}
__device__ __forceinline__ bool selected(int mode, const bb_row_verdict* ver, uint64_t t) {
    return mode == BB_FMT_ALL || (ver[t].pass != 0) == (mode == BB_FMT_KEPT);
}
__device__ __forceinline__ uint32_t slot_of(const bb_group_dev* groups, const bb_row& r) {
    const bb_group_dev& G = groups[r.group_idx];
    return (uint32_t)G.count_off + (r.barcode_idx >= 0 ? (uint32_t)r.barcode_idx : (uint32_t)G.n_seqs);
}

__global__ __launch_bounds__(256) void k_fmt_len(const bb_row* __restrict__ rows, const bb_row_verdict* __restrict__ ver, uint64_t n_rows, int mode,
                                                 const bb_group_dev* __restrict__ groups, const uint32_t* __restrict__ loff, bb_headers h,
                                                 uint32_t* __restrict__ len) {
    const uint64_t t = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    if (t >= n_rows) return;
    if (!selected(mode, ver, t)) { len[t] = 0u; return; }
    const bb_row r = rows[t];
    const uint8_t* id = h.hdr + h.hdr_offsets[r.read_idx];
    const uint32_t idl = h.id_len[r.read_idx];
    uint32_t nq = 0;
    bool quote = false;
    for (uint32_t i = 0; i < idl; ++i) {  // csv QuoteStyle::Necessary: delimiter, quote, CR, LF
        const uint8_t c = id[i];
        nq += c == '"';
        quote = quote || c == '"' || c == '\t' || c == '\n' || c == '\r';
    }
    const uint32_t s = slot_of(groups, r);
    const int32_t rd = r.rel_dist_to_end;
    uint32_t n = idl + (quote ? 2u + nq : 0u) + 15u;  // 14 tabs + newline
    n += n_digits(r.read_len) + (rd < 0 ? 1u : 0u) + n_digits((uint32_t)(rd < 0 ? -(int64_t)rd : rd));
    n += n_digits(r.read_start_bar) + n_digits(r.read_end_bar) + n_digits(r.read_start_flank) + n_digits(r.read_end_flank);
    n += n_digits(r.bar_start) + n_digits(r.bar_end);
    n += r.match_type == BB_FTAG || r.match_type == BB_RTAG ? 4u : 6u;
    n += (r.flank_cost < 0 ? 1u : 0u) + n_digits((uint32_t)(r.flank_cost < 0 ? -r.flank_cost : r.flank_cost));
    n += (r.barcode_cost < 0 ? 1u : 0u) + n_digits((uint32_t)(r.barcode_cost < 0 ? -r.barcode_cost : r.barcode_cost));
    n += loff[s + 1] - loff[s];
    n += r.strand == BB_FWD ? 3u : 2u;
    if (mode != BB_FMT_ALL) n += cuts_len(ver[t]);
    len[t] = n;
}

__global__ __launch_bounds__(256) void k_fmt_render(const bb_row* __restrict__ rows, const bb_row_verdict* __restrict__ ver, uint64_t n_rows, int mode,
                                                    const bb_group_dev* __restrict__ groups, const uint8_t* __restrict__ lblob,
                                                    const uint32_t* __restrict__ loff, bb_headers h, const uint32_t* __restrict__ len,
                                                    const uint64_t* __restrict__ pos, uint8_t* __restrict__ text) {
    const uint64_t t = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    if (t >= n_rows || len[t] == 0u) return;
    const bb_row r = rows[t];
    uint8_t* p = text + pos[t];
    const uint8_t* id = h.hdr + h.hdr_offsets[r.read_idx];
    const uint32_t idl = h.id_len[r.read_idx];
    bool quote = false;
    for (uint32_t i = 0; i < idl; ++i) { const uint8_t c = id[i]; quote = quote || c == '"' || c == '\t' || c == '\n' || c == '\r'; }
    if (quote) *p++ = '"';
    for (uint32_t i = 0; i < idl; ++i) { const uint8_t c = id[i]; *p++ = c; if (quote && c == '"') *p++ = '"'; }
    if (quote) *p++ = '"';
    *p++ = '\t'; p = put_u32(p, r.read_len);
    *p++ = '\t';
    { const int32_t rd = r.rel_dist_to_end; if (rd < 0) *p++ = '-'; p = put_u32(p, (uint32_t)(rd < 0 ? -(int64_t)rd : rd)); }
    *p++ = '\t'; p = put_u32(p, r.read_start_bar);
    *p++ = '\t'; p = put_u32(p, r.read_end_bar);
    *p++ = '\t'; p = put_u32(p, r.read_start_flank);
    *p++ = '\t'; p = put_u32(p, r.read_end_flank);
    *p++ = '\t'; p = put_u32(p, r.bar_start);
    *p++ = '\t'; p = put_u32(p, r.bar_end);
    *p++ = '\t';
    switch (r.match_type) {  // barcodes.rs:25-32
        case BB_FTAG: p = put_str(p, "Ftag", 4); break;
        case BB_RTAG: p = put_str(p, "Rtag", 4); break;
        case BB_FFLANK: p = put_str(p, "Fflank", 6); break;
        default: p = put_str(p, "Rflank", 6); break;
    }
    *p++ = '\t';
    if (r.flank_cost < 0) *p++ = '-';
    p = put_u32(p, (uint32_t)(r.flank_cost < 0 ? -r.flank_cost : r.flank_cost));
    *p++ = '\t';
    if (r.barcode_cost < 0) *p++ = '-';
    p = put_u32(p, (uint32_t)(r.barcode_cost < 0 ? -r.barcode_cost : r.barcode_cost));
    *p++ = '\t';
    { const uint32_t s = slot_of(groups, r); for (uint32_t i = loff[s]; i < loff[s + 1]; ++i) *p++ = lblob[i]; }
    *p++ = '\t';
    if (r.strand == BB_FWD) p = put_str(p, "Fwd", 3); else p = put_str(p, "Rc", 2);  // searcher.rs:67-75
    *p++ = '\t';
    if (mode != BB_FMT_ALL) {
        const bb_row_verdict v = ver[t];
        for (uint32_t q = 0; q < v.n_cuts && q < BB_MAX_CUTS; ++q) {
            if (q) *p++ = ',';
            if (v.cuts[q].direction == BB_CUT_AFTER) p = put_str(p, "After(", 6); else p = put_str(p, "Before(", 7);
            p = put_u32(p, v.cuts[q].group_id);
            *p++ = ')'; *p++ = ':';
            p = put_u32(p, v.match_idx);
        }
    }
    *p++ = '\n';
}

__global__ __launch_bounds__(256) void k_fmt_count(const uint32_t* __restrict__ len, uint64_t n, unsigned long long* __restrict__ cnt) {
    const uint64_t t = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    const unsigned long long m = __ballot(t < n && len[t] != 0u);
    if ((threadIdx.x & 63u) == 0u && m) atomicAdd(cnt, (unsigned long long)__popcll(m));
}

}  // namespace

void bb_format_state_free(bb_format_state* s) {
    if (!s) return;
    for (void* p : {(void*)s->d_blob, (void*)s->d_off, (void*)s->d_len, (void*)s->d_pos, s->d_cub, (void*)s->d_tot})
        if (p) (void)hipFree(p);
    delete s;
}

extern "C" int bb_format_set_labels(bb_ctx* ctx, const uint8_t* blob, const uint32_t* offsets) {
    if (!ctx || !blob || !offsets) return BB_E_INVALID;
    bb_ctx_view v = bb_ctx_get_view(ctx);
    FCHK(v, hipSetDevice(v.device));
    if (!*v.format) *v.format = new bb_format_state();
    bb_format_state* s = *v.format;
    const uint32_t n = bb_counts_len(ctx);
    if (s->d_blob) (void)hipFree(s->d_blob);
    if (s->d_off) (void)hipFree(s->d_off);
    s->d_blob = nullptr; s->d_off = nullptr;
    // labels are installed as the csv crate would write them (QuoteStyle::Necessary, annotator.rs:246-251): a label holding the
    // delimiter, a quote, CR or LF (they come from FASTA headers) is quoted and its quotes doubled — once, here
    std::string q;
    std::vector<uint32_t> qoff((size_t)n + 1, 0u);
    for (uint32_t i = 0; i < n; ++i) {
        qoff[i] = (uint32_t)q.size();
        bool need = false;
        for (uint32_t j = offsets[i]; j < offsets[i + 1]; ++j) need = need || blob[j] == '"' || blob[j] == '\t' || blob[j] == '\n' || blob[j] == '\r';
        if (need) q.push_back('"');
        for (uint32_t j = offsets[i]; j < offsets[i + 1]; ++j) { q.push_back((char)blob[j]); if (need && blob[j] == '"') q.push_back('"'); }
        if (need) q.push_back('"');
    }
    qoff[n] = (uint32_t)q.size();
    FCHK(v, hipMalloc((void**)&s->d_blob, q.size() + 16));
    FCHK(v, hipMalloc((void**)&s->d_off, sizeof(uint32_t) * ((size_t)n + 1)));
    if (!q.empty()) FCHK(v, hipMemcpy(s->d_blob, q.data(), q.size(), hipMemcpyHostToDevice));
    FCHK(v, hipMemcpy(s->d_off, qoff.data(), sizeof(uint32_t) * ((size_t)n + 1), hipMemcpyHostToDevice));
    if (!s->d_tot) FCHK(v, hipMalloc((void**)&s->d_tot, 16));
    s->n_slots = n;
    return BB_OK;
}

extern "C" int bb_format_rows_dev(bb_ctx* ctx, const bb_row* d_rows, const bb_row_verdict* d_ver, uint64_t n_rows, int mode, const bb_headers* h,
                                  uint8_t* d_text, uint64_t text_cap, uint64_t* text_len, uint64_t* n_lines) {
    if (!ctx || !text_len || !h || mode < BB_FMT_ALL || mode > BB_FMT_DROPPED || (n_rows && !d_rows) || (mode != BB_FMT_ALL && n_rows && !d_ver))
        return BB_E_INVALID;
    bb_ctx_view v = bb_ctx_get_view(ctx);
    bb_format_state* s = *v.format;
    if (!s || !s->d_blob) { *v.last_error = "bb_format_set_labels has not been called"; return BB_E_INVALID; }
    *text_len = 0;
    if (n_lines) *n_lines = 0;
    if (n_rows == 0) return BB_OK;
    FCHK(v, hipSetDevice(v.device));
    int r;
    if ((r = fgrow(v, s->d_len, s->cap_len, n_rows))) return r;
    if ((r = fgrow(v, s->d_pos, s->cap_pos, n_rows + 1))) return r;
    const uint32_t nb = (uint32_t)((n_rows + 255) / 256);
    hipLaunchKernelGGL(k_fmt_len, dim3(nb), dim3(256), 0, v.stream, d_rows, d_ver, n_rows, mode, v.d_groups, (const uint32_t*)s->d_off, *h, s->d_len);
    {   // line positions = 64-bit exclusive scan of the lengths (bb_scan.h); its total lands in d_tot[0]
        uint8_t* scr = (uint8_t*)s->d_cub;
        if ((r = fgrow(v, scr, s->cap_cub, ((n_rows + 1023) / 1024 + 2) * sizeof(uint64_t)))) return r;
        s->d_cub = scr;
        FCHK(v, bb_scan64(v.stream, (const uint32_t*)s->d_len, s->d_pos, (uint32_t)n_rows, (uint64_t*)s->d_cub, s->d_tot));
    }
    FCHK(v, hipMemsetAsync(s->d_tot + 1, 0, 8, v.stream));
    hipLaunchKernelGGL(k_fmt_count, dim3(nb), dim3(256), 0, v.stream, (const uint32_t*)s->d_len, n_rows, (unsigned long long*)(s->d_tot + 1));
    uint64_t tot[2] = {0, 0};
    FCHK(v, hipMemcpyAsync(tot, s->d_tot, 16, hipMemcpyDeviceToHost, v.stream));
    FCHK(v, hipStreamSynchronize(v.stream));
    *text_len = tot[0];
    if (n_lines) *n_lines = tot[1];
    if (tot[0] > text_cap || (tot[0] && !d_text)) return BB_E_CAPACITY;
    if (tot[0])
        hipLaunchKernelGGL(k_fmt_render, dim3(nb), dim3(256), 0, v.stream, d_rows, d_ver, n_rows, mode, v.d_groups, (const uint8_t*)s->d_blob,
                           (const uint32_t*)s->d_off, *h, (const uint32_t*)s->d_len, (const uint64_t*)s->d_pos, d_text);
    FCHK(v, hipGetLastError());
    FCHK(v, hipStreamSynchronize(v.stream));
    return BB_OK;
}
