/*
 * barbell_amd_format.h — C-ABI of the annotation.tsv renderer.
 *
 * The reference serialises every `BarbellMatch` with the csv crate (tab-delimited, `\n`, quoting only where needed;
 * `write_annotation_batch`, src/annotate/annotator.rs:13-26; field order and the `cuts` / `match_type` / `strand`
 * spellings: src/annotate/searcher.rs:31-142).  At the rates of this library that host-side formatting is the
 * bottleneck of the CLI, so the rows are rendered where they already are: one kernel pass turns the 48-byte rows of a
 * batch (plus, for filtered.tsv / dropped.tsv, their 16-byte filter verdicts, src/filter/filter.rs:87-119) into the
 * exact bytes of the TSV lines, read ids taken from the header buffer FASTQ ingest left in HBM.  The header line is
 * the caller's (the csv writer emits it with the first record only, so an empty run writes an empty file).
 */
#ifndef BARBELL_AMD_FORMAT_H
#define BARBELL_AMD_FORMAT_H
#include "barbell_amd.h"
#include "barbell_amd_filter.h"
#include "barbell_amd_trim.h"
#ifdef __cplusplus
extern "C" {
#endif

#define BB_FMT_ALL      0   /* every row, empty `cuts` column: annotation.tsv                                  */
#define BB_FMT_KEPT     1   /* rows of reads that pass the filter, `cuts` filled: filtered.tsv                 */
#define BB_FMT_DROPPED  2   /* rows of reads that fail the filter, `cuts` filled: the --dropped file           */

/* Label strings never cross bb_create; the renderer needs them.  One string per histogram slot (the order of
 * bb_counts: for each group its n_seqs labels, then "flank"): blob + bb_counts_len()+1 byte offsets.        */
int bb_format_set_labels(bb_ctx* ctx, const uint8_t* blob, const uint32_t* offsets);

/* Renders d_rows[0, n_rows) (device pointers; d_verdicts may be NULL for BB_FMT_ALL) into d_text.  `d_headers` are
 * the four device arrays of the ingested batch (read ids).  On return *text_len is the number of bytes written;
 * BB_E_CAPACITY with the needed size in *text_len when text_cap is too small.  *n_lines = rows rendered.     */
int bb_format_rows_dev(bb_ctx* ctx, const bb_row* d_rows, const bb_row_verdict* d_verdicts, uint64_t n_rows, int mode,
                       const bb_headers* d_headers, uint8_t* d_text, uint64_t text_cap, uint64_t* text_len, uint64_t* n_lines);

#ifdef __cplusplus
}
#endif
#endif
