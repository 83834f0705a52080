/*
 * barbell_amd_filter.h — C-ABI of the row filter, the first "next" row of SURVEY.md §8(f): the
 * reference's `filter` step (src/filter/filter.rs:10-119, `check_filter_pass` :183-214,
 * `match_pattern` src/filter/pattern.rs:205-240) applied to the rows of a batch while they are still
 * in HBM, instead of a TSV round trip.  Pattern STRINGS (the `pattern_from_str!` language,
 * pattern.rs:242-383) are parsed on the host (barbell_amd/filter.py, csrc/host); what crosses the
 * boundary is the parsed pattern with label constraints resolved to the histogram slot space
 * (per group: n_seqs barcode slots then one "flank" slot — the layout of bb_counts).
 */
#ifndef BARBELL_AMD_FILTER_H
#define BARBELL_AMD_FILTER_H
#include "barbell_amd.h"
#ifdef __cplusplus
extern "C" {
#endif

/* Limits of the filter kernel the reference does not have (its `cuts: Option<Vec<Cut>>` and placeholder map are
 * unbounded, pattern.rs:21-30,128-145); bb_filter_set and both host parsers reject what exceeds them with
 * BB_E_UNSUPPORTED instead of truncating: <= BB_MAX_CUTS cut markers per pattern element, cut group ids <= 65535,
 * <= 16 distinct ?N placeholders per pattern.  The trim planner takes <= 32 cut entries per read (bb_trim_batch
 * returns BB_E_UNSUPPORTED beyond).  None of the kit pattern sets comes near them.                         */
#define BB_MAX_CUTS 3          /* cut markers per pattern element / per row                      */
#define BB_CUT_BEFORE 0        /* "<<" : cut at match start (pattern.rs:9-12)                    */
#define BB_CUT_AFTER  1        /* ">>" : cut at match end                                        */
#define BB_REL_NONE 0
#define BB_REL_LEFT 1          /* @left(a..b)      pattern.rs:157-163                            */
#define BB_REL_RIGHT 2         /* @right(a..b)     pattern.rs:164-171                            */
#define BB_REL_PREV_LEFT 3     /* @prev_left(a..b) pattern.rs:172-181                            */

typedef struct { uint8_t direction; uint8_t _pad; uint16_t group_id; } bb_cut;

/* One PatternElement (pattern.rs:21-30). */
typedef struct {
    uint8_t  match_type;       /* BB_FTAG / BB_RTAG / BB_FFLANK / BB_RFLANK                       */
    int8_t   orientation;      /* -1 = any, BB_FWD, BB_RC                                         */
    uint8_t  relative_to;      /* BB_REL_*                                                        */
    uint8_t  n_cuts;
    int32_t  placeholder;      /* ?N ; -1 = none                                                  */
    int64_t  range_lo, range_hi;
    const uint8_t* label_ok;   /* NULL = any label ("*"); else one byte per histogram slot        */
    bb_cut   cuts[BB_MAX_CUTS];
} bb_pattern_elem;

typedef struct { const bb_pattern_elem* elems; uint32_t n_elems; } bb_pattern;

/* Verdict for one row: whether its read passes (check_filter_pass: the longest matching pattern
 * consumed every row of the read) and the cuts the winning pattern attaches to it — what the
 * reference serialises into the `cuts` column as "After(g):idx,Before(g):idx" (searcher.rs:91-106),
 * idx = match_idx = position of the row inside its read.                                         */
typedef struct {
    uint8_t  pass;
    uint8_t  n_cuts;
    uint16_t match_idx;
    bb_cut   cuts[BB_MAX_CUTS];
} bb_row_verdict;              /* 16 bytes */

/* Installs the pattern set (replaces any previous one).  label_ids: one id per histogram slot;
 * equal label STRINGS must get equal ids (placeholders compare labels, pattern.rs:128-145).     */
int bb_filter_set(bb_ctx* ctx, const bb_pattern* patterns, uint32_t n_patterns, const uint32_t* label_ids);

/* rows as returned by bb_annotate_batch (grouped by read_idx).  Host / device pointer variants.  */
int bb_filter_rows(bb_ctx* ctx, const bb_row* rows, uint64_t n_rows, bb_row_verdict* out);
int bb_filter_rows_dev(bb_ctx* ctx, const bb_row* d_rows, uint64_t n_rows, bb_row_verdict* d_out);

#ifdef __cplusplus
}
#endif
#endif
