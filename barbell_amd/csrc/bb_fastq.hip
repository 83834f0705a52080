// bb_fastq.hip — FASTQ ingest on the GPU (include/barbell_amd_fastq.h).
//
// One block of raw FASTQ text in HBM -> the packed batch layout of the rest of the library:
//   k_nl_count / scan / k_nl_write   positions of all '\n' (16 text bytes per lane, 4 KB per workgroup); the count
//                                    pass keeps per-lane 16-bit masks so the write pass reads 1/8 of the volume
//   k_fq_records                     lane per record: the 4 lines, '@' / '+' checks, "\r\n", header split
//                                    (id up to the first whitespace, description left-trimmed: io.rs:6-17)
//   k_scan64 (x2)                    64-bit exclusive scans of sequence and header lengths -> offsets
//   k_fq_pack                        wave per record: sequence, qualities and header copied into the packed
//                                    arrays as 16-byte chunks (bb_bytes.h)
// HBM-bound byte work: the text is read twice (newline pass, pack pass) and ~its size written once.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/barbell_amd_fastq.h"
#include "bb_bytes.h"
#include "bb_common.h"
#include "bb_ctx_view.h"
#include "bb_scan.h"

struct bb_fastq_state {
    // text staging (host variant)
    uint8_t* d_text = nullptr; uint64_t cap_text = 0;
    // newline pass
    uint32_t* d_cnt = nullptr; uint64_t* d_cbase = nullptr; uint64_t cap_cnt = 0, cap_cbase = 0;
    uint16_t* d_masks = nullptr; uint64_t cap_masks = 0;
    uint64_t* d_nl = nullptr; uint64_t cap_nl = 0; uint32_t last_lpr = 4;
    uint64_t* d_misc = nullptr;  // [0] newline total, [1] sums scratch total, [2] bases total, [3] hdr total, [4] bad record
    // records
    uint32_t *d_seq_len = nullptr, *d_hdr_len = nullptr, *d_id_len = nullptr, *d_desc = nullptr;
    uint64_t *d_off = nullptr, *d_hoff = nullptr, *d_sums = nullptr;
    uint64_t cap_rec = 0, cap_sums = 0;
    uint8_t *d_bases = nullptr, *d_quals = nullptr, *d_hdr = nullptr;
    uint64_t cap_bases = 0, cap_quals = 0, cap_hdr = 0;
    bb_fastq_info last{};
    float last_ms = 0.f;
    hipEvent_t ev[2] = {nullptr, nullptr};
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

__device__ __forceinline__ uint32_t nl_mask16(const uint8_t* __restrict__ text, uint64_t pos, uint64_t len) {
    if (pos >= len) return 0u;
    const u32x4 v = *(const u32x4*)(text + pos);  // text is 16-byte aligned and padded by the allocator
    uint32_t m = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        const uint32_t x = v[w] ^ 0x0A0A0A0Au;  // zero byte where '\n'; bytes tested one by one (the SWAR zero test has false positives)
        m |= (uint32_t)((x & 0xFFu) == 0) << (4 * w) | (uint32_t)((x & 0xFF00u) == 0) << (4 * w + 1) |
             (uint32_t)((x & 0xFF0000u) == 0) << (4 * w + 2) | (uint32_t)((x & 0xFF000000u) == 0) << (4 * w + 3);
    }
    const uint64_t left = len - pos;
    if (left < 16) m &= (1u << left) - 1u;
    return m;
}

// One pass over the text: per-lane 16-bit newline masks are kept (2 bytes per 16 text bytes) so that the
// position pass below reads 1/8 of the text volume instead of the text again.
__global__ __launch_bounds__(256) void k_nl_count(const uint8_t* __restrict__ text, uint64_t len, uint32_t* __restrict__ cnt,
                                                  uint16_t* __restrict__ masks) {
    __shared__ uint32_t s_w[4];
    const uint64_t chunk = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    const uint64_t pos = chunk * 16u;
    const uint32_t m16 = nl_mask16(text, pos, len);
    masks[chunk] = (uint16_t)m16;
    uint32_t c = __popc(m16);
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) c += __shfl_xor(c, d, 64);
    if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) cnt[blockIdx.x] = s_w[0] + s_w[1] + s_w[2] + s_w[3];
}

__global__ __launch_bounds__(256) void k_nl_write(const uint16_t* __restrict__ masks, const uint64_t* __restrict__ base,
                                                  uint64_t* __restrict__ nl) {
    __shared__ uint32_t s_w[4];
    const uint64_t chunk = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    const uint64_t pos = chunk * 16u;
    uint32_t m = masks[chunk];
    const uint32_t c = __popc(m);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    uint32_t inc = c;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const uint32_t y = __shfl_up(inc, d, 64); if (lane >= d) inc += y; }
    if (lane == 63) s_w[wv] = inc;
    __syncthreads();
    uint64_t o = base[blockIdx.x] + inc - c;
    for (int i = 0; i < wv; ++i) o += s_w[i];
    while (m) {
        const int b = __ffs(m) - 1;
        nl[o++] = pos + (uint64_t)b;
        m &= m - 1u;
    }
}

// byte length of the whitespace character (char::is_whitespace, io.rs:6-17: Unicode White_Space, UTF-8 encoded) that
// starts at p[0]; 0 if there is none.  n = bytes available.
__device__ __forceinline__ uint32_t ws_len(const uint8_t* __restrict__ p, uint32_t n) {
    const uint8_t c = p[0];
    if (c == ' ' || (c >= 9 && c <= 13)) return 1u;
    if (c < 0xC2u) return 0u;
    if (c == 0xC2u) return (n >= 2u && (p[1] == 0x85u || p[1] == 0xA0u)) ? 2u : 0u;          // U+0085, U+00A0
    if (n < 3u) return 0u;
    const uint8_t d = p[1], e = p[2];
    if (c == 0xE1u) return (d == 0x9Au && e == 0x80u) ? 3u : 0u;                               // U+1680
    if (c == 0xE2u) {
        if (d == 0x80u) return (e <= 0x8Au && e >= 0x80u) || e == 0xA8u || e == 0xA9u || e == 0xAFu ? 3u : 0u;  // U+2000-200A, 2028, 2029, 202F
        return (d == 0x81u && e == 0x9Fu) ? 3u : 0u;                                           // U+205F
    }
    if (c == 0xE3u) return (d == 0x80u && e == 0x80u) ? 3u : 0u;                               // U+3000
    return 0u;
}

// line i of the block: [start, end) without the newline and without a trailing '\r'
__device__ __forceinline__ void line_span(const uint8_t* __restrict__ text, const uint64_t* __restrict__ nl, uint64_t i, uint64_t& s, uint64_t& e) {
    s = i ? nl[i - 1] + 1 : 0;
    e = nl[i];
    if (e > s && text[e - 1] == '\r') --e;
}

// lpr = lines per record: 4, or 2 for the compact form (header and sequence lines only: BB_FASTQ_TWO_LINE)
__global__ __launch_bounds__(256) void k_fq_records(const uint8_t* __restrict__ text, const uint64_t* __restrict__ nl, uint32_t n_rec, uint32_t lpr, uint32_t packed,
                                                    uint32_t* __restrict__ seq_len, uint32_t* __restrict__ hdr_len,
                                                    uint32_t* __restrict__ id_len, uint32_t* __restrict__ desc_start,
                                                    unsigned long long* __restrict__ bad) {
    const uint32_t k = blockIdx.x * 256u + threadIdx.x;
    if (k >= n_rec) return;
    uint64_t hs, he, ss, se, ps = 0, pe = 0, qs = 0, qe = 0;
    line_span(text, nl, (uint64_t)lpr * k, hs, he);
    line_span(text, nl, (uint64_t)lpr * k + 1, ss, se);
    if (lpr == 4u) {
        line_span(text, nl, 4ull * k + 2, ps, pe);
        line_span(text, nl, 4ull * k + 3, qs, qe);
    }
    bool ok = he > hs && text[hs] == '@' && (lpr != 4u || (pe > ps && text[ps] == '+' && (se - ss) == (qe - qs)));
    uint32_t sl = (uint32_t)(se - ss);
    if (packed) {  // BB_FASTQ_PACKED: two bases per byte + one terminator byte that carries the parity of the line's length
        const uint8_t term = se > ss ? text[se - 1] : 0;
        ok = ok && (term == 'E' || (term == 'O' && se - ss >= 2));
        sl = ok ? 2u * (uint32_t)(se - ss - 1) - (term == 'O' ? 1u : 0u) : 0u;
    }
    if (!ok) atomicMin(bad, (unsigned long long)k);
    const uint32_t hl = he > hs ? (uint32_t)(he - hs - 1) : 0u;  // header without '@'
    uint32_t idl = hl, ds = hl;
    for (uint32_t p = 0; p < hl; ++p)
        if (ws_len(text + hs + 1 + p, hl - p)) { idl = p; break; }
    if (idl < hl) {
        ds = idl;
        for (uint32_t w; ds < hl && (w = ws_len(text + hs + 1 + ds, hl - ds)) != 0u;) ds += w;
    }
    seq_len[k] = sl;
    hdr_len[k] = hl;
    id_len[k] = idl;
    desc_start[k] = ds;
}

// ---- BB_FASTQ_PACKED: two 4-bit base-set codes per byte -> one canonical character per base --------------------------------------
// four packed bytes -> their eight codes, one per byte, in base order (byte b: code of base 2k = b >> 4, of base 2k+1 = (b & 15) ^ 0xA)
__device__ __forceinline__ unsigned long long fq_codes8(uint32_t v) {
    unsigned long long s = (unsigned long long)v;
    s = (s | (s << 16)) & 0x0000FFFF0000FFFFull;
    s = (s | (s << 8)) & 0x00FF00FF00FF00FFull;                     // packed byte k in the low byte of 16-bit slot k
    return (((s >> 4) & 0x000F000F000F000Full) | ((s & 0x000F000F000F000Full) << 8)) ^ 0x0A000A000A000A00ull;
}
// four codes (one per byte) -> "-ACMGRSVTWYHKDBN"[code]: two v_perm_b32 over the two halves of the table, selected by bit 3
__device__ __forceinline__ uint32_t fq_chars4(uint32_t codes) {
    constexpr uint32_t T0L = '-' | ('A' << 8) | ('C' << 16) | ((uint32_t)'M' << 24), T0H = 'G' | ('R' << 8) | ('S' << 16) | ((uint32_t)'V' << 24);
    constexpr uint32_t T1L = 'T' | ('W' << 8) | ('Y' << 16) | ((uint32_t)'H' << 24), T1H = 'K' | ('D' << 8) | ('B' << 16) | ((uint32_t)'N' << 24);
    const uint32_t sel = codes & 0x07070707u;
    const uint32_t r0 = __builtin_amdgcn_perm(T0H, T0L, sel), r1 = __builtin_amdgcn_perm(T1H, T1L, sel);
    const uint32_t m = ((codes >> 3) & 0x01010101u) * 0xFFu;
    return (r0 & ~m) | (r1 & m);
}
__device__ __forceinline__ uint8_t fq_char_at(const uint8_t* __restrict__ src, uint32_t j) {   // base j of a packed line
    const uint8_t b = src[j >> 1];
    return (uint8_t)"-ACMGRSVTWYHKDBN"[(j & 1u) ? ((b & 15u) ^ 0xAu) : (uint32_t)(b >> 4)];
}
// L bases of a packed line -> dst: byte head up to a 16-byte boundary of dst, then 16 bases per lane and step from 8 (or, when the head
// is odd, 9) packed bytes — the parity is the same for the whole record, a wave-uniform branch —, byte tail
static __device__ __forceinline__ void wave_unpack(uint8_t* __restrict__ dst, const uint8_t* __restrict__ src, uint32_t L, int lane) {
    uint32_t head = (uint32_t)((16u - ((uintptr_t)dst & 15u)) & 15u);
    if (head > L) head = L;
    if ((uint32_t)lane < head) dst[lane] = fq_char_at(src, (uint32_t)lane);
    const uint32_t body = (L - head) >> 4;
    u32x4* d = (u32x4*)(dst + head);
    const bool odd = (head & 1u) != 0u;
    for (uint32_t c = (uint32_t)lane; c < body; c += 64u) {
        const uint32_t j0 = head + (c << 4);              // first base of this lane's 16
        const uint8_t* s = src + (j0 >> 1);
        uint32_t w[2];
        __builtin_memcpy(w, s, 8);
        const unsigned long long c0 = fq_codes8(w[0]), c1 = fq_codes8(w[1]);
        uint32_t e[5] = {fq_chars4((uint32_t)c0), fq_chars4((uint32_t)(c0 >> 32)), fq_chars4((uint32_t)c1), fq_chars4((uint32_t)(c1 >> 32)), 0u};
        u32x4 v;
        if (odd) {   // bases j0 .. j0+15 start at the SECOND code of packed byte j0 >> 1: the 18 characters of 9 packed bytes without the first and the last
            e[4] = (uint32_t)"-ACMGRSVTWYHKDBN"[s[8] >> 4];
#pragma unroll
            for (int q = 0; q < 4; ++q) v[q] = (e[q] >> 8) | (e[q + 1] << 24);
        } else {
#pragma unroll
            for (int q = 0; q < 4; ++q) v[q] = e[q];
        }
        __builtin_nontemporal_store(v, d + c);
    }
    const uint32_t done = head + (body << 4);
    if (done + (uint32_t)lane < L) dst[done + lane] = fq_char_at(src, done + (uint32_t)lane);
}

__global__ __launch_bounds__(256) void k_fq_pack(const uint8_t* __restrict__ text, const uint64_t* __restrict__ nl, uint32_t n_rec, uint32_t lpr, uint32_t packed,
                                                 const uint64_t* __restrict__ off, const uint64_t* __restrict__ hoff,
                                                 uint8_t* __restrict__ bases, uint8_t* __restrict__ quals, uint8_t* __restrict__ hdr) {
    const int lane = threadIdx.x & 63;
    const uint32_t k = blockIdx.x * 4u + (threadIdx.x >> 6);
    if (k >= n_rec) return;
    const uint64_t o = off[k], L = off[k + 1] - o, ho = hoff[k], HL = hoff[k + 1] - ho;
    const uint64_t l0 = (uint64_t)lpr * k;
    const uint64_t hs = (k ? nl[l0 - 1] + 1 : 0) + 1;  // past '@'
    const uint64_t ss = nl[l0] + 1;
    if (packed) wave_unpack(bases + o, text + ss, (uint32_t)L, lane);
    else wave_copy(bases + o, text + ss, (uint32_t)L, lane);
    if (lpr == 4u) wave_copy(quals + o, text + nl[l0 + 2] + 1, (uint32_t)L, lane);
    wave_copy(hdr + ho, text + hs, (uint32_t)HL, lane);
}

// bb_annotate_batch_packed: read k of a batch whose sequences came two bases per byte (each read from a byte of its own), a wave per read
__global__ __launch_bounds__(256) void k_unpack_reads(const uint8_t* __restrict__ packed, const uint64_t* __restrict__ poff, const uint64_t* __restrict__ off,
                                                      uint32_t n_reads, uint8_t* __restrict__ bases) {
    const int lane = threadIdx.x & 63;
    const uint32_t k = blockIdx.x * 4u + (threadIdx.x >> 6);
    if (k >= n_reads) return;
    const uint64_t o = off[k];
    wave_unpack(bases + o, packed + poff[k], (uint32_t)(off[k + 1] - o), lane);
}

int scan64(bb_ctx_view& v, bb_fastq_state* s, const uint32_t* in, uint64_t* out, uint32_t n, uint64_t* d_total) {
    const uint32_t nb = (n + 1023) / 1024;
    int r;
    if ((r = fgrow(v, s->d_sums, s->cap_sums, (uint64_t)nb + 1))) return r;
    FCHK(v, bb_scan64(v.stream, in, out, n, s->d_sums, d_total));
    return BB_OK;
}

}  // namespace

void bb_fastq_state_free(bb_fastq_state* s) {
    if (!s) return;
    for (void* p : {(void*)s->d_text, (void*)s->d_cnt, (void*)s->d_cbase, (void*)s->d_masks, (void*)s->d_nl, (void*)s->d_misc, (void*)s->d_seq_len, (void*)s->d_hdr_len,
                    (void*)s->d_id_len, (void*)s->d_desc, (void*)s->d_off, (void*)s->d_hoff, (void*)s->d_sums, (void*)s->d_bases, (void*)s->d_quals,
                    (void*)s->d_hdr})
        if (p) (void)hipFree(p);
    for (auto& e : s->ev) if (e) (void)hipEventDestroy(e);
    delete s;
}

extern "C" int bb_fastq_ingest_dev(bb_ctx* ctx, const uint8_t* d_text, uint64_t text_len, int final_block, bb_fastq_info* info,
                                   bb_fastq_batch_dev* batch) {
    if (!ctx || !info || !batch || (!d_text && text_len) || ((uintptr_t)d_text & 15u)) return BB_E_INVALID;
    const uint32_t lpr = (final_block & BB_FASTQ_TWO_LINE) ? 2u : 4u;  // lines per record
    const uint32_t packed = (final_block & BB_FASTQ_PACKED) ? 1u : 0u;
    if (packed && lpr != 2u) return BB_E_INVALID;                      // the packed form is a form of the two-line block
    final_block &= BB_FASTQ_FINAL;
    bb_ctx_view v = bb_ctx_get_view(ctx);
    if (!*v.fastq) *v.fastq = new bb_fastq_state();
    bb_fastq_state* s = *v.fastq;
    FCHK(v, hipSetDevice(v.device));
    hipStream_t st = v.stream;
    memset(info, 0, sizeof(*info));
    info->bad_record = -1;
    memset(batch, 0, sizeof(*batch));
    s->last = *info;
    if (!s->d_misc) FCHK(v, hipMalloc((void**)&s->d_misc, 8 * sizeof(uint64_t)));
    hipEvent_t* ev = s->ev;
    if (!ev[0]) for (int i = 0; i < 2; ++i) FCHK(v, hipEventCreate(&ev[i]));
    FCHK(v, hipEventRecord(ev[0], st));
    int r;
    uint64_t n_lines = 0;
    uint8_t last_byte = '\n';
    uint8_t tail[16];                                            // the text's last bytes (trailing blank lines, below)
    const uint64_t tn = std::min<uint64_t>(text_len, sizeof tail);
    if (text_len) {
        const uint32_t nb = (uint32_t)((text_len + 4095) / 4096);
        if ((r = fgrow(v, s->d_cnt, s->cap_cnt, nb))) return r;
        if ((r = fgrow(v, s->d_cbase, s->cap_cbase, (uint64_t)nb + 1))) return r;
        if ((r = fgrow(v, s->d_masks, s->cap_masks, (uint64_t)nb * 256))) return r;
        hipLaunchKernelGGL(k_nl_count, dim3(nb), dim3(256), 0, st, d_text, text_len, s->d_cnt, s->d_masks);
        if ((r = scan64(v, s, s->d_cnt, s->d_cbase, nb, s->d_misc))) return r;
        // one round trip for everything the host decides on: the line count and the text's last bytes
        FCHK(v, hipMemcpyAsync(&n_lines, s->d_misc, 8, hipMemcpyDeviceToHost, st));
        FCHK(v, hipMemcpyAsync(tail, d_text + text_len - tn, tn, hipMemcpyDeviceToHost, st));
        FCHK(v, hipStreamSynchronize(st));
        last_byte = tail[tn - 1];
        if ((r = fgrow(v, s->d_nl, s->cap_nl, n_lines + 2))) return r;
        hipLaunchKernelGGL(k_nl_write, dim3(nb), dim3(256), 0, st, (const uint16_t*)s->d_masks, (const uint64_t*)s->d_cbase, s->d_nl);
        FCHK(v, hipGetLastError());
        if (final_block && last_byte != '\n') {  // last line without a newline: a virtual one at text_len (stream order puts it after k_nl_write)
            FCHK(v, hipMemcpyAsync(s->d_nl + n_lines, &text_len, 8, hipMemcpyHostToDevice, st));
            FCHK(v, hipStreamSynchronize(st));
            ++n_lines;
        }
    }
    if (final_block && n_lines && last_byte == '\n') {
        // blank lines after the last record are not lines of a record.  In the 4-line form up to three of them fall out of n_lines / 4 by
        // themselves; in the two-line forms two of them would make a record of their own: take the trailing blank lines off first.
        uint64_t p = tn, nls = 0;
        while (p > 0) {                                    // the maximal suffix of line ends ("\n" or "\r\n")
            if (tail[p - 1] != '\n') break;
            ++nls; --p;
            if (p > 0 && tail[p - 1] == '\r') --p;
        }
        uint64_t blank = (p == 0 && tn == text_len) ? nls : (nls ? nls - 1 : 0);   // the first of them ends the last real line
        blank = std::min(blank, n_lines);
        // a record's last line may be empty (an empty sequence or quality line): only blank lines that cannot belong to a record go —
        // those beyond the last whole record, then whole records of nothing but blank lines (a record starts with '@')
        const uint64_t r2 = n_lines % lpr;
        if (blank >= r2) { n_lines -= r2; blank -= r2; }
        while (blank >= lpr) { n_lines -= lpr; blank -= lpr; }
    }
    if (n_lines / lpr > 0xFFFFFFF0ull) { *v.last_error = "more than 2^32 records in one block"; return BB_E_UNSUPPORTED; }
    const uint32_t n = (uint32_t)(n_lines / lpr);
    info->n_records = n;
    // `consumed` (the end of the last record's last line) comes back with the record totals below: one round trip less per block
    uint64_t consumed = 0;
    if (n == 0) {
        if (final_block && text_len) {  // what is left may only be blank lines
            std::vector<uint8_t> rest((size_t)std::min<uint64_t>(text_len, 1 << 20));
            FCHK(v, hipMemcpy(rest.data(), d_text, rest.size(), hipMemcpyDeviceToHost));
            bool blank = text_len <= rest.size();
            for (uint8_t ch : rest) if (ch != '\n' && ch != '\r') blank = false;
            if (!blank) {
                info->n_records = 0; info->consumed = 0; info->bad_record = 0;
                *v.last_error = "Input FASTQ parsing failed: the stream ends inside record 0";
                return BB_E_FASTQ;
            }
            consumed = text_len;
        }
        info->consumed = consumed;
    }
    if (n) {
        if (n > s->cap_rec || !s->d_seq_len) {
            for (void** p : {(void**)&s->d_seq_len, (void**)&s->d_hdr_len, (void**)&s->d_id_len, (void**)&s->d_desc, (void**)&s->d_off, (void**)&s->d_hoff})
                if (*p) { (void)hipFree(*p); *p = nullptr; }
            const uint64_t cap = (uint64_t)n + n / 4 + 64;
            for (uint32_t** p : {&s->d_seq_len, &s->d_hdr_len, &s->d_id_len, &s->d_desc}) FCHK(v, hipMalloc((void**)p, cap * 4));
            for (uint64_t** p : {&s->d_off, &s->d_hoff}) FCHK(v, hipMalloc((void**)p, (cap + 1) * 8));
            s->cap_rec = cap;
        }
        const unsigned long long none = ~0ull;
        FCHK(v, hipMemcpyAsync(s->d_misc + 4, &none, 8, hipMemcpyHostToDevice, st));
        hipLaunchKernelGGL(k_fq_records, dim3((n + 255) / 256), dim3(256), 0, st, d_text, (const uint64_t*)s->d_nl, n, lpr, packed, s->d_seq_len, s->d_hdr_len,
                           s->d_id_len, s->d_desc, (unsigned long long*)(s->d_misc + 4));
        if ((r = scan64(v, s, s->d_seq_len, s->d_off, n, s->d_misc + 2))) return r;
        if ((r = scan64(v, s, s->d_hdr_len, s->d_hoff, n, s->d_misc + 3))) return r;
        uint64_t h[3];
        FCHK(v, hipMemcpyAsync(h, s->d_misc + 2, sizeof(h), hipMemcpyDeviceToHost, st));
        FCHK(v, hipMemcpyAsync(&consumed, s->d_nl + ((uint64_t)lpr * n - 1), 8, hipMemcpyDeviceToHost, st));
        FCHK(v, hipStreamSynchronize(st));
        consumed = std::min<uint64_t>(consumed + 1, text_len);
        if (final_block && consumed < text_len) {  // what is left may only be blank lines
            std::vector<uint8_t> rest((size_t)std::min<uint64_t>(text_len - consumed, 1 << 20));
            FCHK(v, hipMemcpy(rest.data(), d_text + consumed, rest.size(), hipMemcpyDeviceToHost));
            bool blank = text_len - consumed <= rest.size();
            for (uint8_t ch : rest) if (ch != '\n' && ch != '\r') blank = false;
            if (!blank) {
                info->n_records = n; info->consumed = consumed; info->bad_record = (int64_t)n;
                *v.last_error = "Input FASTQ parsing failed: the stream ends inside record " + std::to_string(n);
                return BB_E_FASTQ;
            }
            consumed = text_len;
        }
        info->consumed = consumed;
        if (h[2] != none) {
            info->bad_record = (int64_t)h[2];
            *v.last_error = "Input FASTQ parsing failed: record " + std::to_string(h[2]) + " of the block is not a " + std::to_string(lpr) + "-line FASTQ record";
            return BB_E_FASTQ;
        }
        info->n_bases = h[0];
        info->n_hdr = h[1];
        if ((r = fgrow(v, s->d_bases, s->cap_bases, h[0] + 16))) return r;
        if (lpr == 4u && (r = fgrow(v, s->d_quals, s->cap_quals, h[0] + 16))) return r;
        if ((r = fgrow(v, s->d_hdr, s->cap_hdr, h[1] + 16))) return r;
        hipLaunchKernelGGL(k_fq_pack, dim3((n + 3) / 4), dim3(256), 0, st, d_text, (const uint64_t*)s->d_nl, n, lpr, packed, (const uint64_t*)s->d_off,
                           (const uint64_t*)s->d_hoff, s->d_bases, s->d_quals, s->d_hdr);
        FCHK(v, hipGetLastError());
    }
    FCHK(v, hipEventRecord(ev[1], st));
    FCHK(v, hipStreamSynchronize(st));
    (void)hipEventElapsedTime(&s->last_ms, ev[0], ev[1]);
    s->last = *info;
    s->last_lpr = lpr;
    batch->d_bases = s->d_bases; batch->d_quals = lpr == 4u ? s->d_quals : nullptr; batch->d_offsets = s->d_off;
    batch->d_headers = bb_headers{s->d_hdr, s->d_hoff, s->d_id_len, s->d_desc};
    return BB_OK;
}

extern "C" int bb_fastq_ingest(bb_ctx* ctx, const uint8_t* text, uint64_t text_len, int final_block, bb_fastq_info* info,
                               bb_fastq_batch_dev* batch) {
    if (!ctx || !info || !batch || (!text && text_len)) return BB_E_INVALID;
    bb_ctx_view v = bb_ctx_get_view(ctx);
    if (!*v.fastq) *v.fastq = new bb_fastq_state();
    bb_fastq_state* s = *v.fastq;
    FCHK(v, hipSetDevice(v.device));
    int r;
    if ((r = fgrow(v, s->d_text, s->cap_text, text_len + 64))) return r;
    if (text_len) FCHK(v, hipMemcpyAsync(s->d_text, text, text_len, hipMemcpyHostToDevice, v.stream));
    return bb_fastq_ingest_dev(ctx, s->d_text, text_len, final_block, info, batch);
}

extern "C" int bb_fastq_fetch(bb_ctx* ctx, uint64_t* offsets, uint8_t* hdr, uint64_t* hdr_offsets, uint32_t* id_len, uint32_t* desc_start,
                              uint8_t* bases, uint8_t* quals) {
    if (!ctx) return BB_E_INVALID;
    bb_ctx_view v = bb_ctx_get_view(ctx);
    bb_fastq_state* s = *v.fastq;
    if (!s) { *v.last_error = "no FASTQ block has been ingested"; return BB_E_INVALID; }
    FCHK(v, hipSetDevice(v.device));
    const uint64_t n = s->last.n_records;
    if (n == 0) {
        if (offsets) offsets[0] = 0;
        if (hdr_offsets) hdr_offsets[0] = 0;
        return BB_OK;
    }
    hipStream_t st = v.stream;
    if (offsets) FCHK(v, hipMemcpyAsync(offsets, s->d_off, (n + 1) * 8, hipMemcpyDeviceToHost, st));
    if (hdr_offsets) FCHK(v, hipMemcpyAsync(hdr_offsets, s->d_hoff, (n + 1) * 8, hipMemcpyDeviceToHost, st));
    if (id_len) FCHK(v, hipMemcpyAsync(id_len, s->d_id_len, n * 4, hipMemcpyDeviceToHost, st));
    if (desc_start) FCHK(v, hipMemcpyAsync(desc_start, s->d_desc, n * 4, hipMemcpyDeviceToHost, st));
    if (hdr && s->last.n_hdr) FCHK(v, hipMemcpyAsync(hdr, s->d_hdr, s->last.n_hdr, hipMemcpyDeviceToHost, st));
    if (bases && s->last.n_bases) FCHK(v, hipMemcpyAsync(bases, s->d_bases, s->last.n_bases, hipMemcpyDeviceToHost, st));
    if (quals && s->last.n_bases) FCHK(v, hipMemcpyAsync(quals, s->d_quals, s->last.n_bases, hipMemcpyDeviceToHost, st));
    FCHK(v, hipStreamSynchronize(st));
    return BB_OK;
}

extern "C" int bb_fastq_fetch_lines(bb_ctx* ctx, uint64_t* line_ends) {
    if (!ctx) return BB_E_INVALID;
    bb_ctx_view v = bb_ctx_get_view(ctx);
    bb_fastq_state* s = *v.fastq;
    if (!s) { *v.last_error = "no FASTQ block has been ingested"; return BB_E_INVALID; }
    const uint64_t n = s->last.n_records * s->last_lpr;
    if (n == 0) return BB_OK;
    if (!line_ends) return BB_E_INVALID;
    FCHK(v, hipSetDevice(v.device));
    FCHK(v, hipMemcpyAsync(line_ends, s->d_nl, n * 8, hipMemcpyDeviceToHost, v.stream));
    FCHK(v, hipStreamSynchronize(v.stream));
    return BB_OK;
}

void bb_launch_unpack_reads(hipStream_t st, const uint8_t* d_packed, const uint64_t* d_poff, const uint64_t* d_off, uint32_t n, uint8_t* d_bases) {
    hipLaunchKernelGGL(k_unpack_reads, dim3((n + 3u) / 4u), dim3(256), 0, st, d_packed, d_poff, d_off, n, d_bases);
}

extern "C" float bb_fastq_last_ms(bb_ctx* ctx) {
    if (!ctx) return 0.f;
    bb_ctx_view v = bb_ctx_get_view(ctx);
    return *v.fastq ? (*v.fastq)->last_ms : 0.f;
}
