// bb_k_trace.h — k_flank_trace: one lane per raw flank hit; (m+k)-column DP with move bits, traceback, get_matching_region
// (cigar_parse.rs:71-82) and the padded barcode window (searcher.rs:453-456).
#pragma once
#include "bb_myers.h"

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
// every 16 columns in LDS; the walk then goes back block by block — the lane recomputes the 16 columns of a block from its
// checkpoint with their move bits and walks through its part of the block — so the DP is computed twice, and nothing lives
// in private memory.  Round 6: the block's move bits stay in REGISTERS (2 x W words per column, the walk's column loop unrolled so
// that every index is static) instead of an LDS window, and checkpoints are 16 columns apart instead of 8: 10.7 KB of LDS per
// 64-lane block for the rapid kits' 90-nt flank at k = 20 instead of 33.8 KB — twelve waves share a CU instead of four, and
// the kernel, which waits on memory most of its time (one lane per hit), had 52 % of its wave time parked at one wave per SIMD:
// 6.8 -> 2.9 ms per 2 M-read step.  The price: with three times the lanes in flight the text lines of the first pass are often gone
// from L2 when the walk asks for them again (1.95 -> 3.65 GB of HBM traffic per step).  Keeping the text in LDS as well (7 KB more per
// block: nine waves per CU) brought the traffic back to 1.87 GB and the time to 3.8 ms — measured, not kept: the path is not HBM-bound.
#define BB_TRACE_CKB 16
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
    const int m = G.m, k = G.flank_k;
    // get_matching_region's row range (cigar_parse.rs:71-82 with bar_region of barcodes.rs:192), in the rows of THIS walk: policy [H5] — the
    // pattern indices Match::to_path() yields for a Strand::Rc match are either those of the forward flank (rc = complement(flank) against the
    // reversed text: the same rows) or mirrored (rc = reverse complement against the forward text: path index i <-> row m - 1 - i)
    const bool mirror = h.strand != 0 && groups[0].pol_rc_mirror != 0;
    const int bar_lo = mirror ? m - 1 - G.bar_hi : G.bar_lo, bar_hi = mirror ? m - 1 - G.bar_lo : G.bar_hi;
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
    constexpr bool KEEP_TEXT = (MODE == 2 || MODE == 4) && W <= 2;  // the first 64 scan positions of the DP's text stay in registers
    constexpr int NCH = 4;  // 64 columns at once; the rest (m + k > 64: two-word flanks of more than 58 characters) one by one
    uint32_t buf[KEEP_TEXT ? NCH : 1][4];
    if constexpr (KEEP_TEXT) {
        // band variants (k <= 6: at most 32 W + 6 columns): every chunk of the window's text requested before the first column. Its
        // 50-70 bytes lie in one or two lines; fetched a chunk at a time as the DP got there, a line was often gone from L2 again by
        // the next request once 24 waves per CU were in flight (1.2 -> 1.7 GB of HBM reads per step with the 8-row band).
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
        static_assert(BB_TRACE_CKB == 16, "a block's text is one 16-byte chunk");
        uint32_t tq[4], tn[4] = {0u, 0u, 0u, 0u};            // the block's text; the next (lower) block's, requested a block ahead
        {
            const int32_t cl = ((w - 1) / BB_TRACE_CKB) * BB_TRACE_CKB;  // this lane's last block
            if (w > 0) load16(s0 + cl, tn);
        }
        for (int blk = (mk_max - 1) / BB_TRACE_CKB; blk >= 0; --blk) {  // wave-uniform: the largest m + k of the launch's groups
            const int32_t c0 = blk * BB_TRACE_CKB;           // the block holds columns c0+1 .. c0+16
            if (c0 < w) {
#pragma unroll
                for (int q = 0; q < 4; ++q) tq[q] = tn[q];
                if (c0 >= BB_TRACE_CKB) load16(s0 + c0 - BB_TRACE_CKB, tn);
            }
            if (c0 < w && j > 0 && i > c0) {
                uint32_t wl[BB_TRACE_CKB][W], wh[BB_TRACE_CKB][W];   // the block's move bits: registers (every index below is a compile-time constant)
#pragma unroll
                for (int x = 0; x < W; ++x) { pv[x] = s_moves[((blk * 2 * W) + x) * 64 + threadIdx.x]; mv[x] = s_moves[((blk * 2 * W) + W + x) * 64 + threadIdx.x]; }
#pragma unroll
                for (int b = 0; b < BB_TRACE_CKB; ++b) {
                    const uint32_t ch = (tq[b >> 2] >> (8 * (b & 3))) & 0xFFu;
                    uint32_t eq[W], d0[W], ph[W], mh[W];
                    load_eq<W, S>(peq, ch, eq);
                    myers_step<W>(pv, mv, eq, d0, ph, mh);       // (columns beyond w: computed on whatever the chunk holds, never walked)
                    move_bits_prio<W>(prio, eq, d0, ph, pv, wl[b], wh[b]);
                }
#pragma unroll
                for (int cc = BB_TRACE_CKB - 1; cc >= 0; --cc) {
                    while (j > 0 && i == c0 + cc + 1) {          // ops that stay in the column (rows deleted) repeat here; the others leave it
                        const int bit = j - 1, q = bit >> 5;
                        uint32_t lw = wl[cc][0], hw = wh[cc][0];
#pragma unroll
                        for (int x = 1; x < W; ++x) { lw = q == x ? wl[cc][x] : lw; hw = q == x ? wh[cc][x] : hw; }
                        take(((lw >> (bit & 31)) & 1u) | (((hw >> (bit & 31)) & 1u) << 1));
                    }
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
    if (wn <= 64) {  // window codes for the barcode kernels: the window's bytes, then the base-set LUT
        u32x4_t tv[4];
        // The window lies inside the flank match plus 10 positions either side: almost always inside the 64 scan positions the DP's
        // text was loaded for.  Those bytes are still in registers: they go through the lane's staging row (LDS) and come back at the
        // window's byte offset — asking HBM for them again cost as much as the first time (the lines are long gone from L2: 50 MB of
        // them are in flight), 0.45 GB per 2 M-read step.  Windows that stick out of the buffer take the loads.
        bool from_regs = false;
        if constexpr (KEEP_TEXT) {
            // offset of the window's first forward byte in the buffer, in FORWARD byte order: the forward strand's buffer is in that
            // order from s0; the rc strand's holds scan position p = n - 1 - f, so reversed (words swapped, bytes swapped) it holds
            // forward position f = n - 1 - s0 - 63 + x at x
            const int32_t a = h.strand ? ws - (n - 1 - s0 - 63) : ws - s0;
            const int32_t ext = min(64, 16 * ((w + 15) / 16));   // scan positions [0, ext) of the buffer were loaded (whole 16-byte chunks up to the hit's end)
            from_regs = h.strand ? (a >= 64 - ext && a + wn <= 64) : (a >= 0 && a + wn <= ext);
            if (__any(from_regs)) {
                uint32_t* st = orec + 8;  // 16 words of the lane's staging row (the record's window area: written with the codes below)
#pragma unroll
                for (int x = 0; x < 16; ++x) st[x] = h.strand ? __builtin_bswap32(buf[(15 - x) >> 2][(15 - x) & 3]) : buf[x >> 2][x & 3];
                const int32_t q = from_regs ? a >> 2 : 0, sh = from_regs ? a & 3 : 0;
                uint32_t w[17];
#pragma unroll
                for (int x = 0; x < 17; ++x) w[x] = st[min(q + x, 15)];
#pragma unroll
                for (int x = 0; x < 16; ++x)
                    if (from_regs) tv[x >> 2][x & 3] = __builtin_amdgcn_alignbyte(w[x + 1], w[x], (uint32_t)sh);
            }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (from_regs) continue;
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
                                                    const uint32_t* __restrict__ slot_base, bb_hit* __restrict__ hits, uint32_t* __restrict__ hit_meta,
                                                    uint32_t gmask, int mk_max, const uint32_t* __restrict__ n_hits_dev) {
    extern __shared__ uint32_t s_dyn[];
    __shared__ uint32_t s_slot[64];
    BB_HITS_ON_DEVICE(n_hits, n_hits_dev, 64u);
    static_assert(sizeof(bb_hit) == 96, "six 16-byte pieces");
    // the staged records reuse the move bits' LDS (>= 64 * BB_TRACE_REC_STRIDE words, launch_trace): the block is one wave,
    // a lane writes its record after every lane's walk is over, and LDS operations of a wave execute in order
    uint32_t* s_rec = s_dyn;
    s_slot[threadIdx.x] = flank_trace_lane<W, MODE>(bases, offsets, tables, groups, n_groups, raw, n_hits, slot_base, gmask, mk_max,
                                                    s_rec + threadIdx.x * BB_TRACE_REC_STRIDE, s_dyn);
    {   // what k_hit_lists and the barcode kernels' lane assignment need of a hit, 4 bytes instead of its 96-byte record (bb_hit_meta)
        const uint32_t slot = s_slot[threadIdx.x];
        if (slot != 0xFFFFFFFFu) {
            const uint32_t* r = s_rec + threadIdx.x * BB_TRACE_REC_STRIDE;
            hit_meta[slot] = bb_hit_meta(r[6] & 0xFFu, (r[5] >> 16) & 0xFFu, r[5] >> 24, r[4] - r[3]);
        }
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < 64u * 6u; i += 64u) {
        const uint32_t hl = i / 6u, pc = i - hl * 6u, slot = s_slot[hl];
        if (slot != 0xFFFFFFFFu) {
            const uint32_t* r = s_rec + hl * BB_TRACE_REC_STRIDE + 4u * pc;
            reinterpret_cast<uint4*>(hits + slot)[pc] = make_uint4(r[0], r[1], r[2], r[3]);
        }
    }
}

