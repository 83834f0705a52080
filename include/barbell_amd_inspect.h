/*
 * barbell_amd_inspect.h — C-ABI of the inspect step, part of the fourth "next" row of SURVEY.md §8(f):
 * `get_group_structure` (src/inspect/inspect.rs:15-117) on the rows of a batch in HBM.  Every row
 * becomes one pattern element "{match_type}[{fw|rc}, *{cut}, {position tag}]"; the GPU computes the
 * element's fields (which needs the previous row of the same read), the host joins the elements of a
 * read with "__" into the pattern string of pattern_per_read.tsv and counts patterns (inspect.rs:128-200).
 */
#ifndef BARBELL_AMD_INSPECT_H
#define BARBELL_AMD_INSPECT_H
#include "barbell_amd.h"
#include "barbell_amd_filter.h"
#ifdef __cplusplus
extern "C" {
#endif

/* One pattern element. tag: BB_REL_LEFT "@left(lo..hi)", BB_REL_RIGHT "@right(lo..hi)", BB_REL_PREV_LEFT
 * "@prev_left(lo..hi)" (inspect.rs:41-70); has_cut: the row carries cuts -> ", <<" on Fwd, ", >>" on Rc
 * (inspect.rs:72-85).  first != 0 on the first row of a read.                                        */
typedef struct {
    uint8_t  match_type, strand, has_cut, tag;
    uint32_t lo, hi;
    uint32_t first;
} bb_inspect_elem;             /* 16 bytes */

/* verdicts may be NULL (inspect of annotation.tsv: no cuts).  bucket_size: `-s`, default 250. */
int bb_inspect_rows(bb_ctx* ctx, const bb_row* rows, const bb_row_verdict* verdicts, uint64_t n_rows, uint32_t bucket_size,
                    bb_inspect_elem* out);
int bb_inspect_rows_dev(bb_ctx* ctx, const bb_row* d_rows, const bb_row_verdict* d_verdicts, uint64_t n_rows, uint32_t bucket_size,
                        bb_inspect_elem* d_out);

#ifdef __cplusplus
}
#endif
#endif
