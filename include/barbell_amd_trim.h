/*
 * barbell_amd_trim.h — C-ABI of the trim/split step, the second "next" row of SURVEY.md §8(f): the
 * reference's `trim` (src/trim/trim.rs) applied to the reads of a batch while reads, rows and filter
 * verdicts are still in HBM.  For every read that passed the filter:
 *   preprocess_cuts        trim.rs:127-254   cut groups -> CompleteSlice {start, end, annotations}
 *   process_read_and_anno  trim.rs:256-300   slice / skip_trim / flip (reverse complement) / label / suffix
 *   the FASTQ record text  trim.rs:447-460   "@{id}{suffix}[ {desc}]\n{seq}\n+\n{qual}\n"
 * The GPU renders the records of the whole batch into ONE text buffer grouped by output label (records
 * of one label contiguous and in read order), so the host side of `trim_matches` (trim.rs:317-480)
 * reduces to one write() per label file.  Label STRINGS stay on the host: a record's label is the key
 * (part0+1) << 16 | (part1+1), part = label_id * 2 + strand bit, which the host formats with
 * LabelConfig::create_label's join rules (trim.rs:58-105).
 */
#ifndef BARBELL_AMD_TRIM_H
#define BARBELL_AMD_TRIM_H
#include "barbell_amd.h"
#include "barbell_amd_filter.h"
#ifdef __cplusplus
extern "C" {
#endif

#define BB_SIDE_NONE  0
#define BB_SIDE_LEFT  1        /* LabelSide::Left  (trim.rs:24-28) */
#define BB_SIDE_RIGHT 2

/* TrimConfig (config.rs:19-32) minus the host-only fields (failed writer path, verbose, gzip). */
typedef struct {
    uint8_t add_labels, add_orientation, add_flank, sort_labels;
    uint8_t only_side;         /* BB_SIDE_* ; sort_labels with a side is rejected like trim.rs:330-334 */
    uint8_t write_full_header; /* "@id{suffix} desc" when the record has a description               */
    uint8_t skip_trim;         /* write the whole read for every surviving slice                     */
    uint8_t flip;              /* reverse-complement slices that carry an Ftag matched on Rc          */
} bb_trim_config;

/* One surviving CompleteSlice = one output record. 32 bytes. */
typedef struct {
    uint32_t read_idx;
    uint32_t start, end;       /* slice of the read, end exclusive (also set when skip_trim)         */
    uint32_t label_key;        /* 0 = "none"; else (part0+1)<<16 | (part1+1), low half 0 = one part */
    uint16_t suffix;           /* slice_count of trim.rs:271: 0 = no suffix, n = "_n"                */
    uint8_t  flip;
    uint8_t  _pad;
    uint32_t rec_len;          /* bytes of the rendered record                                       */
    uint64_t out_off;          /* byte offset of the record in the text buffer                       */
} bb_slice;

/* Records of one label: text[off, off+len), slices[first, first+n_records). */
typedef struct {
    uint32_t label_key;
    uint32_t n_records;
    uint64_t first;
    uint64_t off, len;
} bb_label_span;

#define BB_TRIM_NONE    0      /* read has no passing rows                                            */
#define BB_TRIM_TRIMMED 1      /* >= 1 record written  (TRIMMED_IDX, trim.rs:417)                     */
#define BB_TRIM_FAILED  2      /* passed the filter but produced no record (FAILED_IDX, trim.rs:419)  */

/* Read headers of a batch: hdr = all header lines (without '@') back to back, hdr_offsets[n+1];
 * id_len[i] = bytes of the read id (up to the first whitespace), desc_start[i] = offset inside the
 * header at which the description starts (split_fastq_header, io.rs:6-17), == header length if none. */
typedef struct {
    const uint8_t*  hdr;
    const uint64_t* hdr_offsets;
    const uint32_t* id_len;
    const uint32_t* desc_start;
} bb_headers;

/* Installs the trim configuration.  Tables are indexed by the label ids given to bb_filter_set
 * (n_label_ids = max id + 1): label_is_flank[id] != 0 when the label string contains "flank"
 * (trim.rs:66-68); part_rank[id*2 + strand] = rank of the formatted part ("BC01_fw", or "BC01" without
 * orientation) in lexicographic order, used by sort_labels (trim.rs:95-97).                          */
int bb_trim_set(bb_ctx* ctx, const bb_trim_config* cfg, const uint8_t* label_is_flank, const uint32_t* part_rank,
                uint32_t n_label_ids);

/* Host-pointer variant.  rows/verdicts as returned by bb_annotate_batch / bb_filter_rows for the batch
 * whose reads are bases/quals/offsets (quals share the offsets).  On return
 *   text[0, *text_len)        all records, grouped by label_key ascending, read order inside a label
 *   slices[0, *n_slices)      one per record, in text order
 *   spans[0, *n_spans)        one per label, in text order
 *   read_status[n_reads]      BB_TRIM_*
 * BB_E_CAPACITY with the needed sizes in *text_len / *n_slices / *n_spans when a buffer is too small. */
int bb_trim_batch(bb_ctx* ctx, const bb_row* rows, const bb_row_verdict* verdicts, uint64_t n_rows,
                  const uint8_t* bases, const uint8_t* quals, const uint64_t* offsets, const bb_headers* headers,
                  uint32_t n_reads, uint8_t* text, uint64_t text_cap, uint64_t* text_len, bb_slice* slices,
                  uint64_t slices_cap, uint64_t* n_slices, bb_label_span* spans, uint32_t spans_cap, uint32_t* n_spans,
                  uint8_t* read_status);

/* Device-pointer variant: every array (including the four of `d_headers`) lives in HBM; the three
 * counts come back on the host.  spans are returned in text order.                                  */
int bb_trim_batch_dev(bb_ctx* ctx, const bb_row* d_rows, const bb_row_verdict* d_verdicts, uint64_t n_rows,
                      const uint8_t* d_bases, const uint8_t* d_quals, const uint64_t* d_offsets, const bb_headers* d_headers,
                      uint32_t n_reads, uint8_t* d_text, uint64_t text_cap, uint64_t* text_len, bb_slice* d_slices,
                      uint64_t slices_cap, uint64_t* n_slices, bb_label_span* d_spans, uint32_t spans_cap, uint32_t* n_spans,
                      uint8_t* d_read_status);

/* The plan without the text: slices (text order: grouped by label_key ascending, read order inside a label; out_off / rec_len as
 * bb_trim_batch_dev would lay the records out), spans, statuses and *text_len — everything `trim_matches` (trim.rs:317-480) decides,
 * nothing it copies.  For callers whose reads are in host memory anyway (a FASTQ file staged for upload): the records are then cut
 * out of that text by the threads that write the per-label files (host/bb_writers.cpp), and neither the qualities nor the rendered records
 * cross PCIe.  Needs no bases or qualities on the device: works on a BB_FASTQ_TWO_LINE block.                                    */
int bb_trim_plan_dev(bb_ctx* ctx, const bb_row* d_rows, const bb_row_verdict* d_verdicts, uint64_t n_rows, const uint64_t* d_offsets,
                     const bb_headers* d_headers, uint32_t n_reads, uint64_t* text_len, bb_slice* d_slices, uint64_t slices_cap,
                     uint64_t* n_slices, bb_label_span* d_spans, uint32_t spans_cap, uint32_t* n_spans, uint8_t* d_read_status);

/* GPU milliseconds of the last bb_trim_batch_dev: which = 0 plan + sort, 1 render, 2 both. */
float bb_trim_last_ms(bb_ctx* ctx, int which);

#ifdef __cplusplus
}
#endif
#endif
