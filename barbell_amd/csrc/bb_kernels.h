// bb_kernels.h — hand-written HIP kernels (gfx950 / CDNA4, wave64) of the annotate hot path.
//
// Pipeline per batch, all integer bit-twiddling, no MFMA:
//   flank scan     sassy's search of the N-masked flank over the whole read, both strands (searcher.rs:438): either
//                  k_flank_filter (15 or 31 rows of the flank, both strands in one pass over the text, flags per 16 bytes)
//                  + k_flank_verify (the full-height scan around flagged columns only), or k_flank_scan2 (the full-height
//                  scan of every column, one lane per (read, strand)).
//                  Local-minimum ends <= k are the raw flank hits.
//   scan           exclusive scan of per-(read,group,strand) hit counts -> deterministic slots.
//   k_flank_trace  one lane per flank hit: (m+k)-column DP with move bits, traceback,
//                  get_matching_region + window padding (cigar_parse.rs:71-82, searcher.rs:453-456).
//   barcode stage  one lane per (flank hit, barcode) (searcher.rs:267-426): k_bar_prefix (shared leading rows, once per
//                  hit), k_barcode_pfx (one Myers word per lane; fast variant: score bounds, top-2 per hit), k_rows (exact
//                  score of the best-bounded path, decision), the exact variant for undecided hits; k_barcode_reg /
//                  k_barcode for geometries outside the row split.
//   k_collapse     one lane per read: collapse_overlapping_matches (interval.rs:4-79).
//   scan + k_emit  compaction of surviving rows in read order + per-barcode histogram.
#pragma once
#include "bb_myers.h"
#include "bb_k_scan.h"
#include "bb_k_trace.h"
#include "bb_k_bar_common.h"
#include "bb_k_bar_generic.h"
#include "bb_k_bar_pfx.h"
#include "bb_k_rows.h"
#include "bb_k_misc.h"
