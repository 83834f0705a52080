// bb_len.h — reads of differing lengths (real nanopore runs: most reads a few hundred to a few thousand nt, a tail to 100 kb and beyond; the
// benchmark's reads all have 4000).  The scans give a lane one read, so a wave runs at the pace of its longest lane and a batch at the pace
// of its longest read: measured on a heavy-tailed mix, the scan stage took 13 x (SQK-NBD114-96) the time of the same bases in equal reads.
//   k_len_hist     one pass over the offsets: line counts of the reads (128-byte lines as the scans stream them), their minimum / maximum,
//                  and a histogram of SEGMENT line counts — a read of more than split_above lines is cut into segments of seg_lines lines.
//                  The host reads it in the round trip it makes for the batch's byte span anyway; batches
//                  of (nearly) equal reads stop here and the scans run as they always did.
//   k_len_scatter  otherwise: the segments (read, index) into `vtab` by FALLING length (counting sort; the bins' start positions come from
//                  the host) — k_flank_filter and the full scan (k_flank_scan_seg) take their lanes' work from it; the segments of cut
//                  reads also get a cell each (contiguous per read) for the full scan's hit counts.
// Which lane scans what never shows in the results: flags are addressed by text position, hits by (read, ordinal).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "bb_lenstat.h"

__device__ __forceinline__ uint32_t bb_len_lines(const uint8_t* bases, uint64_t off, uint32_t n) {
    return n ? (uint32_t)((((uint64_t)(uintptr_t)(bases + off) & 127u) + n + 127u) >> 7) : 0u;
}

// (grid-stride over at most 1024 blocks, a wave of equal reads counted with one LDS atomic and the extremes reduced across the wave first:
// with an atomic per read and 8 K blocks the batch of 2 M equal reads — every lane on the same cell — took 187 us)
__global__ __launch_bounds__(256) void k_len_hist(const uint8_t* __restrict__ bases, const uint64_t* __restrict__ offsets, uint32_t n_reads,
                                                  uint32_t seg_lines, uint32_t split_above, bb_lenstat* __restrict__ st) {
    __shared__ uint32_t s_seg[BB_LEN_SEG_BINS], s_mm[2];
    for (uint32_t i = threadIdx.x; i < BB_LEN_SEG_BINS; i += 256u) s_seg[i] = 0u;
    if (threadIdx.x == 0) { s_mm[0] = 0xFFFFFFFFu; s_mm[1] = 0u; }
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t lo = 0xFFFFFFFFu, hi = 0u;
    for (uint32_t base = blockIdx.x * 256u; base < n_reads; base += gridDim.x * 256u) {   // (block-uniform trip count)
        const uint32_t read = base + threadIdx.x;
        const bool live = read < n_reads;
        uint32_t nl = 0u;
        if (live) {
            const uint64_t off = offsets[read];
            nl = bb_len_lines(bases, off, (uint32_t)(offsets[read + 1] - off));
            lo = min(lo, nl); hi = max(hi, nl);
        }
        // whole reads (empty ones too: the full scan closes them — overhang positions, count): one LDS atomic per distinct line count of the wave
        unsigned long long rest = __ballot(live && nl <= split_above);
        while (rest) {
            const uint32_t v = (uint32_t)__shfl((int)nl, __ffsll((long long)rest) - 1, 64);
            const unsigned long long same = __ballot(nl == v) & rest;
            if (lane == 0u && v < BB_LEN_SEG_BINS) atomicAdd(&s_seg[v], (uint32_t)__popcll(same));   // (seg_lines = 0: no bins are read)
            rest &= ~same;
        }
        if (live && nl > split_above) {   // cut reads (few)
            const uint32_t nseg = (nl + seg_lines - 1u) / seg_lines;
            atomicAdd(&s_seg[seg_lines], nseg - 1u);
            atomicAdd(&s_seg[nl - (nseg - 1u) * seg_lines], 1u);
            atomicAdd(&st->n_cut_reads, 1u); atomicAdd(&st->n_cut_segs, nseg);
        }
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { lo = min(lo, (uint32_t)__shfl_xor((int)lo, d, 64)); hi = max(hi, (uint32_t)__shfl_xor((int)hi, d, 64)); }
    if (lane == 0u) { atomicMin(&s_mm[0], lo); atomicMax(&s_mm[1], hi); }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < BB_LEN_SEG_BINS; i += 256u) if (s_seg[i]) atomicAdd(&st->seg[i], s_seg[i]);
    if (threadIdx.x == 0) {
        atomicMin(&st->min_nl, s_mm[0]); atomicMax(&st->max_nl, s_mm[1]);
        if (blockIdx.x == 0) { st->off0 = offsets[0]; st->off1 = offsets[n_reads]; }
    }
}

__global__ __launch_bounds__(256) void k_len_scatter(const uint8_t* __restrict__ bases, const uint64_t* __restrict__ offsets, uint32_t n_reads,
                                                     uint32_t seg_lines, uint32_t split_above, bb_lencur* __restrict__ cur,
                                                     uint2* __restrict__ vtab,
                                                     uint32_t* __restrict__ vcut /* per vtab entry: its cell among the cut reads' segments */,
                                                     uint32_t* __restrict__ cutread /* per cell: the read */, uint4* __restrict__ cutlist /* per cut read: read, first cell, segments */) {
    __shared__ uint32_t s_seg[BB_LEN_SEG_BINS];
    for (uint32_t i = threadIdx.x; i < BB_LEN_SEG_BINS; i += 256u) s_seg[i] = 0u;
    __syncthreads();
    const uint32_t read = blockIdx.x * 256u + threadIdx.x;
    uint32_t nl = 0u, nseg = 0u, last_lines = 0u, slot_full = 0u, slot_last = 0u;
    if (read < n_reads) {
        const uint64_t off = offsets[read];
        nl = bb_len_lines(bases, off, (uint32_t)(offsets[read + 1] - off));
        if (nl > split_above) {
            nseg = (nl + seg_lines - 1u) / seg_lines;
            last_lines = nl - (nseg - 1u) * seg_lines;
            slot_full = atomicAdd(&s_seg[seg_lines], nseg - 1u);
        } else { nseg = 1u; last_lines = nl; }
        slot_last = atomicAdd(&s_seg[last_lines], 1u);
    }
    __syncthreads();
    // the block's share of every bin, reserved with one atomic per bin in use; the LDS cell then holds where the share begins
    for (uint32_t i = threadIdx.x; i < BB_LEN_SEG_BINS; i += 256u) if (s_seg[i]) s_seg[i] = atomicAdd(&cur->seg[i], s_seg[i]);
    __syncthreads();
    if (read < n_reads) {
        // (a cut read's last segment may have seg_lines lines as well: its slot and the full segments' slots are distinct draws of one cell)
        uint32_t cell0 = 0xFFFFFFFFu;
        if (nseg > 1u) {
            cell0 = atomicAdd(&cur->cut_segs, nseg);
            cutlist[atomicAdd(&cur->cut_reads, 1u)] = make_uint4(read, cell0, nseg, 0u);
            for (uint32_t t = 0; t < nseg; ++t) cutread[cell0 + t] = read;
        }
        for (uint32_t t = 0; t + 1u < nseg; ++t) {
            const uint32_t v = s_seg[seg_lines] + slot_full + t;
            vtab[v] = make_uint2(read, t); vcut[v] = cell0 + t;
        }
        const uint32_t v = s_seg[last_lines] + slot_last;
        vtab[v] = make_uint2(read, nseg - 1u); vcut[v] = nseg > 1u ? cell0 + nseg - 1u : 0xFFFFFFFFu;
    }
}
