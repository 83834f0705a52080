/*
 * bb_oracle.h — CPU restatement (plain C) of Barbell's annotate hot path.  TEST INFRASTRUCTURE.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may link or call this; the
 * product library (barbell_amd/csrc) never does.  See oracle/README.md for the pinning status:
 * the arithmetic of this path lives in un-vendored crates (sassy 0.2.1, cigar-lodhi-rs 0.1.0,
 * pa-types 1.2.0) whose sources are absent from /root/reference, so everything beyond the
 * reference's own known-answer tests is "parity unpinned" and isolated in the policy functions
 * marked H1..H9 in bb_oracle.c.
 */
#ifndef BB_ORACLE_H
#define BB_ORACLE_H

#include "../include/barbell_amd.h"
#include "../include/barbell_amd_policy.h"
#include "../include/barbell_amd_filter.h"
#include "../include/barbell_amd_trim.h"
#include "../include/barbell_amd_inspect.h"
#include "../include/barbell_amd_fastq.h"

#ifdef __cplusplus
extern "C" {
#endif

/* unit CIGAR ops, pa-types naming (Pos(i,j): i = pattern, j = text) */
#define BBO_MATCH 0  /* +(1,1) cost 0 */
#define BBO_SUB   1  /* +(1,1) cost 1 */
#define BBO_INS   2  /* +(0,1) text char not in pattern */
#define BBO_DEL   3  /* +(1,0) pattern char not in text */

/* restatement of sassy::Match (fields Barbell reads: SURVEY Appendix A) */
typedef struct {
    int32_t text_start, text_end;        /* forward text coordinates, end exclusive          */
    int32_t pattern_start, pattern_end;  /* part of the pattern inside the text (overhang)   */
    int32_t cost;
    int32_t strand;                      /* BB_FWD / BB_RC                                   */
    int32_t pattern_idx;
    int32_t n_ops;
    uint8_t* ops;                        /* unit ops in pattern order (malloc'd)             */
    int32_t rc_text_len;                 /* text length, needed by to_path for Rc matches    */
    int32_t rc_mirror_len;               /* policy [H5]: 0, or the pattern length m of an Rc match whose path indices are m-1-i */
} bbo_match;

typedef struct { int32_t i, j; } bbo_pos;   /* Pos(pattern idx, text idx) */

/* sassy Searcher::<Iupac>::search restated: alpha < 0 -> no overhang; rc != 0 -> also search the
 * reverse complement.  Returns the number of matches, *out is malloc'd (free with bbo_free_matches). */
int  bbo_search(const uint8_t* pat, int m, const uint8_t* text, int n, int k, float alpha, int rc,
                bbo_match** out);
void bbo_free_matches(bbo_match* ms, int n);
/* Match::to_path: one Pos per unit op, path[t] = cell at which op t is applied. returns n_ops */
int  bbo_to_path(const bbo_match* m, bbo_pos* path /* n_ops entries */);

/* cigar_parse.rs:6-45 / :71-82 */
int  bbo_map_pat_to_text_with_cost(const bbo_match* m, int p_start, int p_end,
                                   int* pat_lo, int* pat_hi, int* txt_lo, int* txt_hi, int* cost);
int  bbo_get_matching_region(const bbo_match* m, int start, int end, int* lo, int* hi);

/* cigar-lodhi-rs Lodhi::new(3, 0.5).compute restated (H8) on a unit-op string */
double bbo_lodhi(const uint8_t* ops, int n_ops);
/* edit_model.rs:2-11 */
int    bbo_edit_cut_off(int l);
/* searcher.rs:183-199 */
int    bbo_rel_dist_to_end(long pos, long read_len);
/* interval.rs:4-79; collapses `rows` (n) in place, returns the new count */
int    bbo_collapse(bb_row* rows, int n, float filter_overlap);
/* IUPAC profile: 4-bit code, 0xFF = invalid */
uint8_t bbo_iupac_code(uint8_t c);

/* whole-path API with the same shape as the product's C-ABI */
typedef struct bbo_ctx bbo_ctx;
int  bbo_create(const bb_group_desc* groups, uint32_t n_groups, const bb_params* params, bbo_ctx** out);
/* the same under an explicit policy (include/barbell_amd_policy.h; NULL = default); bbo_create uses the policy last set
 * with bbo_set_policy, which also governs the stand-alone bbo_search / bbo_lodhi (NULL resets it to the default) */
int  bbo_create_policy(const bb_group_desc* groups, uint32_t n_groups, const bb_params* params, const bb_policy* policy, bbo_ctx** out);
int  bbo_set_policy(const bb_policy* policy);
void bbo_destroy(bbo_ctx* ctx);
int  bbo_group_get_info(const bbo_ctx* ctx, uint32_t group, bb_group_info* info);
int  bbo_group_get_flank(const bbo_ctx* ctx, uint32_t group, uint8_t* out);
int  bbo_group_get_pattern(const bbo_ctx* ctx, uint32_t group, uint32_t idx, int rc, uint8_t* out);
/* Demuxer::demux over a batch; n_threads > 1 uses OpenMP over reads (one scratch per thread,
 * like the reference's one Demuxer per paraseq worker, annotator.rs:88-101). */
int  bbo_annotate_batch(bbo_ctx* ctx, const uint8_t* bases, const uint64_t* offsets, uint32_t n_reads,
                        bb_row* rows, uint64_t rows_cap, uint64_t* n_rows, int n_threads);
/* The same rows from bit-parallel scans (64-bit Myers / Hyyro words instead of scalar DP cells).  TIMING ONLY: bench.py's
 * cpu_baseline reports it so the CPU figure is not a scalar loop against a reference that runs AVX2 sassy; the parity tests
 * compare the GPU with bbo_annotate_batch, and tests/test_oracle_fast.py compares this function with it. */
void bbo_set_pin_threads(int on);   /* default 1: the batch entry points run one pinned OpenMP worker per allowed CPU */
int  bbo_fast_is_simd(void);   /* the timing path runs its AVX-512 forms (bb_oracle_simd.h) on this CPU; BBO_NO_SIMD=1 switches them off */
int  bbo_annotate_batch_fast(bbo_ctx* ctx, const uint8_t* bases, const uint64_t* offsets, uint32_t n_reads,
                             bb_row* rows, uint64_t rows_cap, uint64_t* n_rows, int n_threads);

/* tools/policy_feasible.py: the same per-read procedure with the reference's own invariants COUNTED instead of aborting, and the barcode
 * windows compared with where the constructs were planted.  truth: per read BBO_TRUTH_PER_READ records of BBO_TRUTH_FIELDS int32 —
 * group, strand, construct_lo, construct_hi, bar_lo, bar_hi (text coordinates, hi exclusive), barcode idx; group = -1: none. */
#define BBO_TRUTH_PER_READ 2
#define BBO_TRUTH_FIELDS   7
typedef struct {
    uint64_t flank_matches;       /* matches the flank search returned (searcher.rs:438)                                  */
    uint64_t region_none;         /* get_matching_region -> None (searcher.rs:445-449: the reference skips the match)      */
    uint64_t slice_panic;         /* window start beyond its end: `read[ws..we]` would panic (searcher.rs:456)             */
    uint64_t subpath_none;        /* map_pat_to_text_with_cost -> None: expect("No barcode match region found") (:388)    */
    uint64_t on_target;           /* flank matches lying on a planted construct of their group and strand                  */
    uint64_t window_overlaps;     /* ... whose barcode window overlaps the planted barcode                                 */
    uint64_t window_covers;       /* ... whose barcode window contains it                                                  */
    uint64_t tag_rows_on_target;  /* ... that became a tag row (before collapse)                                           */
    uint64_t tag_rows_correct;    /* ... with the planted barcode                                                          */
    uint64_t rows;                /* rows after collapse                                                                   */
} bbo_diag;
int  bbo_annotate_diag(bbo_ctx* ctx, const uint8_t* bases, const uint64_t* offsets, uint32_t n_reads, const int32_t* truth, int n_threads,
                       int fast, bbo_diag* out);
/* filter step on a row stream (filter.rs:183-214 check_filter_pass, pattern.rs:96-240 match_pattern);
 * rows grouped by consecutive read_idx like the reference groups by consecutive read_id (filter.rs:54-85) */
int  bbo_filter_rows(const bbo_ctx* ctx, const bb_pattern* patterns, uint32_t n_patterns, const uint32_t* label_ids,
                     const bb_row* rows, uint64_t n_rows, bb_row_verdict* out);
/* trim/split step on a batch (trim.rs:127-300 preprocess_cuts + process_read_and_anno, record text
 * trim.rs:447-460); same outputs as bb_trim_batch, configuration passed directly */
int  bbo_trim_batch(const bbo_ctx* ctx, const bb_trim_config* cfg, const uint8_t* label_is_flank, const uint32_t* part_rank,
                    const uint32_t* label_ids, const bb_row* rows, const bb_row_verdict* verdicts, uint64_t n_rows,
                    const uint8_t* bases, const uint8_t* quals, const uint64_t* offsets, const bb_headers* headers, uint32_t n_reads,
                    uint8_t* text, uint64_t text_cap, uint64_t* text_len, bb_slice* slices, uint64_t slices_cap, uint64_t* n_slices,
                    bb_label_span* spans, uint32_t spans_cap, uint32_t* n_spans, uint8_t* read_status);
/* inspect step: get_group_structure (inspect.rs:15-117) per row */
int  bbo_inspect_rows(const bb_row* rows, const bb_row_verdict* verdicts, uint64_t n_rows, uint32_t bucket_size, bb_inspect_elem* out);
/* FASTQ ingest restated: same outputs as bb_fastq_ingest + bb_fastq_fetch; with any array NULL only `info`
 * is computed (sizing call) */
int  bbo_fastq_parse(const uint8_t* text, uint64_t len, int final_block, bb_fastq_info* info, uint64_t* offsets, uint8_t* bases,
                     uint8_t* quals, uint8_t* hdr, uint64_t* hdr_offsets, uint32_t* id_len, uint32_t* desc_start);
/* test hook: 1 = trace flank matches on the full DP matrix instead of the (m+k)-column window */
void bbo_set_full_trace(int on);

#ifdef __cplusplus
}
#endif
#endif
