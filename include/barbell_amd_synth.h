/*
 * barbell_amd_synth.h — deterministic synthetic read generator (bench / test input only; not part
 * of the drop-in boundary).  Read i of a stream is a pure function of (seed, i): any shard can be
 * generated on any GPU or on the host bit-identically.  The design
 * follows the reference's simulator (benchmarks/src/simulations/sim_data.rs:163-447): random ACGT
 * body; 80 % reads carry <lead 0..60><front+barcode+rear> at the 5' end, half of those also the
 * construct near the 3' end (reverse complement for a single group, the second group's construct
 * for dual-end query sets); 10 % no adapter; 5 % 5'-truncated adapter (1..20 nt missing);
 * 5 % a second, different barcode mid-read; 2 % sub / 1 % ins / 1 % del inside constructs.
 */
#ifndef BARBELL_AMD_SYNTH_H
#define BARBELL_AMD_SYNTH_H
#include "barbell_amd.h"
#ifdef __cplusplus
extern "C" {
#endif
/* n+1 byte offsets of reads first_read .. first_read+n-1 (length uniform in [len_min, len_max]) */
int bb_synth_offsets(uint64_t seed, uint32_t len_min, uint32_t len_max, uint64_t first_read, uint32_t n, uint64_t* offsets);
/* host generation (no GPU needed) */
int bb_synth_reads_host(const bb_group_desc* groups, uint32_t n_groups, uint64_t seed, uint32_t len_min, uint32_t len_max,
                        uint64_t first_read, uint32_t n, const uint64_t* offsets, uint8_t* bases);
/* device generation straight into HBM (d_offsets / d_bases are device pointers) */
int bb_synth_reads_dev(bb_ctx* ctx, uint64_t seed, uint32_t len_min, uint32_t len_max, uint64_t first_read, uint32_t n,
                       const uint64_t* d_offsets, uint8_t* d_bases);
#ifdef __cplusplus
}
#endif
#endif
