// bb_lenstat.h — what k_len_hist (bb_len.h) leaves for the host: the batch's byte span, the reads' line counts (128-byte lines as the scans
// stream them) and the histogram of segment line counts.  The host-pointer form of a small batch fills the same record from the offsets it
// holds (bb_host_lenstat, barbell_amd.hip) and spares the kernel and its round trip.
#pragma once
#include <stdint.h>

#define BB_LEN_SEG_BINS 130u   // segment line counts 0 .. 129 (split_above <= 128)
struct bb_lenstat {
    unsigned long long off0, off1;   // offsets[0], offsets[n]
    uint32_t min_nl, max_nl;
    uint32_t n_cut_reads, n_cut_segs;   // reads cut into segments, and their segments
    uint32_t seg[BB_LEN_SEG_BINS];
};
struct bb_lencur { uint32_t seg[BB_LEN_SEG_BINS]; uint32_t cut_reads, cut_segs; };   // next free position of every bin / list
