// bb_k_bar_prefix.h — the leading shared rows of the padded barcodes once per flank hit (the row split's prefix records, read by
// k_barcode_pfx): k_bar_prefix over every hit, k_bar_prefix_list over the hits of a list.  The traceback order is a run-time value here.
#pragma once
#include "bb_k_bar_common.h"

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
                                                    const bb_hit* __restrict__ hits, uint32_t n_hits, bb_hit_pfx* __restrict__ out, uint32_t n_groups,
                                                    const uint32_t* __restrict__ n_hits_dev) {
    BB_HITS_ON_DEVICE(n_hits, n_hits_dev, 128u);
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
            const uint32_t prio = (uint32_t)G.pol_prio;
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
                    uint32_t hp, hm;
                    shared_rows_column<BB_PRIO_RT>(prio, eq, P, pv, mv, hp, hm, shw);  // row r <-> bit P - r; top boundary row: D[0][c] = 0, no horizontal delta
                    PH |= (unsigned long long)hp << c;
                    MH |= (unsigned long long)hm << c;
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

// The prefix records of the hits on a list (the hits k_rows left undecided: k_barcode_pfx's exact variant reads them), where no
// k_bar_prefix has run over every hit because k_barcode_lane computes its own.  A lane per listed hit, records written in place: the
// lists are a few per cent of the hits, coalescing does not matter here.
__global__ __launch_bounds__(128) void k_bar_prefix_list(const uint8_t* __restrict__ tables, const bb_group_dev* __restrict__ groups,
                                                         const bb_hit* __restrict__ hits, const uint32_t* __restrict__ list,
                                                         const uint32_t* __restrict__ cnt, bb_hit_pfx* __restrict__ out) {
    const uint32_t n = *cnt;
    for (uint32_t i = blockIdx.x * 128u + threadIdx.x; i < n; i += gridDim.x * 128u) {
        const uint32_t idx = list[i];
        const bb_hit& H = hits[idx];
        const uint32_t strand = H.strand & 1u;
        const bb_group_dev& G = groups[H.group];
        const int32_t wn = (int32_t)(H.we - H.ws);
        bb_hit_pfx R;
        R.ph = R.mh = 0ull;
#pragma unroll
        for (int q = 0; q < BB_MAX_TAIL; ++q) R.teq[q] = 0ull;
        if (H.valid && G.split[strand] && wn <= 64) {
            const int P = G.pfx[strand], T = G.tail[strand];
            const uint32_t* eqt = reinterpret_cast<const uint32_t*>(tables + G.off_peq_pfx[strand]);
            const uint8_t* tlut = tables + G.off_tail_lut[strand];
            uint32_t pv = P ? (P >= 32 ? 0xFFFFFFFFu : (1u << P) - 1u) : 0u, mv = 0u;
            for (int c = 0; c < 64; ++c) {
                uint32_t shw = 0u;
                if (c < wn) {
                    const uint32_t code = H.win[c] & 0xFu;
                    const uint32_t tb = tlut[code];
                    for (int q = 0; q < T; ++q) R.teq[q] |= (unsigned long long)((tb >> q) & 1u) << c;
                    if (P > 0) {
                        uint32_t hp, hm;
                        shared_rows_column<BB_PRIO_RT>((uint32_t)G.pol_prio, eqt[code], P, pv, mv, hp, hm, shw);
                        R.ph |= (unsigned long long)hp << c; R.mh |= (unsigned long long)hm << c;
                    }
                }
                out[idx].sh[c] = shw;
            }
        } else {
            for (int c = 0; c < 64; ++c) out[idx].sh[c] = 0u;
        }
        out[idx].ph = R.ph; out[idx].mh = R.mh;
#pragma unroll
        for (int q = 0; q < BB_MAX_TAIL; ++q) out[idx].teq[q] = R.teq[q];
    }
}
