/*
 * barbell_amd.h — C-ABI of the MI355X-native Barbell annotate hot path.
 *
 * This is the drop-in boundary for ONE path of rickbeeloo/barbell: the per-read match loop
 * `Demuxer::demux` (reference src/annotate/searcher.rs:430-490) together with its constructor
 * (`Demuxer::new` :202-217, `add_query_group` :220-226) and the one-time query preparation
 * (`BarcodeGroup::new`, src/annotate/barcodes.rs:105-197).  The reference has no FFI of its own;
 * the narrowest seam is `DemuxProcessor::process_record` -> `Demuxer::demux`
 * (src/annotate/annotator.rs:123-135).  A per-read call is the wrong granularity for a GPU, so the
 * boundary is per BATCH and maps 1:1 onto paraseq's hooks: `process_record` appends the read to a
 * staging buffer, `on_batch_complete` (annotator.rs:137-139) makes one `bb_annotate_batch` call and
 * turns the returned rows into `BarbellMatch` values for the existing `write_annotation_batch`
 * (annotator.rs:13-26).  INTEGRATION.md shows that Rust-side stub.
 *
 * Conventions: plain C, caller owns every buffer, nothing throws across the boundary, every
 * reference `panic!` on this path becomes a negative error code.  Strings never cross: the caller
 * maps `read_idx` back to its read id and `barcode_idx` back to its label.  All entry points need
 * a gfx950 GPU; there is no CPU fallback in this library (the CPU restatement lives in oracle/ and
 * is test infrastructure only).
 */
#ifndef BARBELL_AMD_H
#define BARBELL_AMD_H

#include <stdint.h>
#include <stddef.h>

#include "barbell_amd_policy.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- error codes (negative; 0 = ok) ------------------------------------------------------- */
#define BB_OK               0
#define BB_E_INVALID       -1  /* null pointer / bad argument                                        */
#define BB_E_ONE_QUERY     -2  /* group with a single query sequence (barcodes.rs:113-117 panic)     */
#define BB_E_UNEQUAL_LEN   -3  /* sequences of a group differ in length (barcodes.rs:325-328)        */
#define BB_E_NO_BARCODE    -4  /* prefix+suffix cover the whole sequence (barcodes.rs:124-128)       */
#define BB_E_NO_FLANK      -5  /* neither shared prefix nor suffix (barcodes.rs:131-133)             */
#define BB_E_NOT_IUPAC     -6  /* non-IUPAC character in a query (barcodes.rs:45-47)                 */
#define BB_E_CAPACITY      -7  /* caller's row buffer too small; *n_rows holds the required count    */
#define BB_E_NO_DEVICE     -8  /* no gfx950 device / HIP runtime failure at create                   */
#define BB_E_HIP           -9  /* HIP runtime error during a batch (see bb_last_error)               */
#define BB_E_UNSUPPORTED  -10  /* geometry outside what the kernels were built for (table below)    */
#define BB_E_NOMEM        -11
#define BB_E_FASTQ        -12  /* malformed FASTQ record (barbell_amd_fastq.h)                         */

/* ---- limits (the reference's BarcodeGroup::new, barcodes.rs:105-197, has none; every shipped kit fits) --------
 *   query groups per context                      <= 32
 *   sequences per group                           <= 1024
 *   flank = prefix + barcode mask + suffix        <= 256 nt (<= 128 nt run the tuned scan/trace instantiations)
 *   padded barcode pattern (10 + barcode + 10)    <= 128 nt (<= 48 nt and windows <= 64 columns: register-resident kernels;
 *                                                            beyond that the any-geometry kernel: correct, several times slower)
 *   flank error budget (--flank-max-errors)       <= 127 (the automatic cutoff of a 256-nt flank is 103)
 *   barcode window = barcode + flank errors + 20  <= 256 columns
 *   filter: cut markers per pattern element <= 3, cut group id <= 65535, distinct ?N placeholders per pattern <= 16
 *   trim: cut entries per read <= 32
 * bb_create returns BB_E_UNSUPPORTED beyond them and leaves the reason for bb_last_error(NULL).                */

/* ---- match_type / strand encodings (searcher.rs:31-75, barcodes.rs:8-33) ------------------ */
#define BB_FTAG   0
#define BB_RTAG   1
#define BB_FFLANK 2
#define BB_RFLANK 3
#define BB_FWD    0
#define BB_RC     1

/* One query group = one `BarcodeGroup` (barcodes.rs:57-72): N equally long sequences
 * <shared prefix><barcode><shared suffix>.  `flank_k` is `--flank-max-errors`
 * (bin/main.rs:90-91); a negative value selects the automatic cutoff
 * `get_edit_cut_off(prefix_len + suffix_len)` (edit_model.rs:2-11, annotator.rs:219-226).      */
typedef struct {
    const uint8_t* const* seqs;      /* n_seqs pointers, ASCII IUPAC                              */
    const uint32_t*       seq_lens;  /* n_seqs lengths; must all be equal (barcodes.rs:325-328)   */
    uint32_t              n_seqs;
    uint8_t               type;      /* BB_FTAG or BB_RTAG                                        */
    int32_t               flank_k;   /* <0 = auto                                                 */
} bb_group_desc;

/* `Demuxer::new(alpha, verbose, min_score_frac, min_score_diff_frac)` (searcher.rs:202) with the
 * CLI defaults of bin/main.rs:98-111: alpha 0.4, min_score 0.2, min_score_diff 0.1.            */
typedef struct {
    float   alpha;
    double  min_score;
    double  min_score_diff;
    int32_t device;                  /* HIP device ordinal                                        */
} bb_params;

/* One `BarbellMatch` (searcher.rs:31-64) without its strings.  48 bytes, POD.                  */
typedef struct {
    uint32_t read_idx;               /* index of the read inside the batch                        */
    uint32_t read_len;
    int32_t  rel_dist_to_end;        /* searcher.rs:183-199                                       */
    uint32_t read_start_bar, read_end_bar;
    uint32_t read_start_flank, read_end_flank;
    uint32_t bar_start, bar_end;
    int16_t  flank_cost, barcode_cost;
    int16_t  barcode_idx;            /* index into the group's sequences; -1 = label "flank"      */
    uint8_t  group_idx;
    uint8_t  match_type;             /* BB_FTAG / BB_RTAG / BB_FFLANK / BB_RFLANK                 */
    uint8_t  strand;                 /* BB_FWD / BB_RC                                            */
    uint8_t  _pad[3];
} bb_row;

/* Geometry of a prepared group, for callers that want to display it like
 * `BarcodeGroup::display` (barcodes.rs:199-248) or test it like barcodes.rs:488-546.           */
typedef struct {
    uint32_t flank_len, prefix_len, suffix_len, mask_len;
    uint32_t bar_lo, bar_hi;         /* bar_region, inclusive end (barcodes.rs:192)               */
    uint32_t pad_lo, pad_hi;         /* pad_region, pad_hi unclamped (barcodes.rs:160-163)        */
    uint32_t pattern_len;            /* length of the padded barcode patterns                     */
    int32_t  flank_k;                /* resolved flank cutoff                                     */
    int32_t  bar_k1, bar_k2;         /* barcode cutoffs of the two passes (searcher.rs:458-460)   */
    double   perfect_score;          /* searcher.rs:229-239                                       */
} bb_group_info;

typedef struct bb_ctx bb_ctx;        /* opaque; one per host thread / GPU stream; NOT thread-safe */

/* Replaces Demuxer::new + add_query_group + BarcodeGroup::new.  Returns BB_OK or an error code. */
int bb_create(const bb_group_desc* groups, uint32_t n_groups, const bb_params* params, bb_ctx** out);
void bb_destroy(bb_ctx* ctx);
/* The same under an explicit policy (barbell_amd_policy.h: what sassy 0.2.1 / cigar-lodhi-rs 0.1.0 are ASSUMED to do where
 * Barbell's own code does not pin it; searcher.rs:209-211,282-301,364-396,438).  NULL = the default.  bb_create takes
 * the policy from the environment variable BARBELL_AMD_POLICY (text form) when it is set, the default otherwise.      */
int bb_create_policy(const bb_group_desc* groups, uint32_t n_groups, const bb_params* params, const bb_policy* policy, bb_ctx** out);
int bb_get_policy(const bb_ctx* ctx, bb_policy* out);

int bb_n_groups(const bb_ctx* ctx);
int bb_group_get_info(const bb_ctx* ctx, uint32_t group, bb_group_info* info);
/* copies the N-masked flank (flank_len bytes) / one padded pattern (pattern_len bytes) */
int bb_group_get_flank(const bb_ctx* ctx, uint32_t group, uint8_t* out);
int bb_group_get_pattern(const bb_ctx* ctx, uint32_t group, uint32_t idx, int rc, uint8_t* out);

/* Replaces the per-read loop `for record in batch { demuxer.demux(id, seq) }`
 * (annotator.rs:123-135).  `bases` = concatenated read sequences exactly as they appear in the
 * FASTQ (no normalisation, annotator.rs:126-127), `offsets` = n_reads+1 byte offsets.
 * Rows come back ordered by (read_idx, read_start_flank) — within a read that is the order
 * `collapse_overlapping_matches` returns (interval.rs:12,27) — so the TSV is deterministic.
 * Host-pointer variant: copies the batch to the GPU and the rows back.                         */
int bb_annotate_batch(bb_ctx* ctx, const uint8_t* bases, const uint64_t* offsets, uint32_t n_reads,
                      bb_row* rows, uint64_t rows_cap, uint64_t* n_rows);

/* The same with the sequences TWO BASES PER BYTE: what crosses PCIe is ~2 KB per 4-kb read instead of 4 (the host-pointer form is PCIe-bound from a
 * few thousand reads per call on: INTEGRATION.md, "Batch size").  Every kernel looks at a read character only through its IUPAC base set — a 4-bit
 * code: A=1 C=2 G=4 T/U=8, their unions, 0 = not an IUPAC letter (matches nothing) — so rows are those of the original text.  Read i's bases
 * 2j, 2j+1 are byte packed[packed_offsets[i] + j] = (code << 4) | (code' ^ 0xA), an odd last base paired with anything; every read begins at a byte of
 * its own; `offsets` are the reads' offsets in BASES (n_reads + 1, as for bb_annotate_batch), `packed_offsets` in bytes of `packed`.
 * bb_pack_bases writes the packed form of `n` characters (AVX2 / AVX-512 where the CPU has them; a binding calls it per record as it collects
 * a batch, annotator.rs:123-135) and returns the bytes written, (n + 1) / 2.  `out` must not overlap `bases`.                                    */
uint64_t bb_pack_bases(const uint8_t* bases, uint64_t n, uint8_t* out);
int bb_annotate_batch_packed(bb_ctx* ctx, const uint8_t* packed, const uint64_t* packed_offsets, const uint64_t* offsets, uint32_t n_reads,
                             bb_row* rows, uint64_t rows_cap, uint64_t* n_rows);

/* Device-pointer variant: `d_bases`, `d_offsets` and `d_rows` are HIP device pointers on the
 * context's device.  The call enqueues on the context's stream and returns after the row count is
 * known (one small D2H of 8 bytes); rows stay in HBM.                                          */
int bb_annotate_batch_dev(bb_ctx* ctx, const uint8_t* d_bases, const uint64_t* d_offsets,
                          uint32_t n_reads, bb_row* d_rows, uint64_t rows_cap, uint64_t* n_rows);

/* Histogram of emitted rows per (group, barcode): for each group n_seqs tag counters followed by
 * one flank-only counter; groups concatenated.  Accumulates over batches until bb_counts_reset.
 * `bb_counts_dev` returns the device pointer (uint64, bb_counts_len entries) so a multi-GPU
 * driver can all-reduce it with RCCL without a host round trip.                                */
uint32_t  bb_counts_len(const bb_ctx* ctx);
int       bb_counts(bb_ctx* ctx, uint64_t* out);
uint64_t* bb_counts_dev(bb_ctx* ctx);
int       bb_counts_reset(bb_ctx* ctx);

/* Timing of the last batch: per-kernel device time in ms measured with HIP events on the
 * context's stream (k = 0..bb_n_kernels()-1), and its name.                                    */
int         bb_n_kernels(void);
const char* bb_kernel_name(int k);
float       bb_last_kernel_ms(const bb_ctx* ctx, int k);
void        bb_set_timing(bb_ctx* ctx, int enable);
/* With timing on: the longest single kernel launch of the last batch's barcode stage (searcher.rs:267-426) — its name as rocprofv3
 * prints it, e.g. "k_barcode_lane<48, false, 216u>", and its duration between two events on the stream it was launched on (launches
 * on the second stream overlap it; the stage's time is bb_last_kernel_ms's "k_barcode").  Empty name: no such launch.             */
int         bb_last_dominant_kernel(const bb_ctx* ctx, char* name, size_t name_cap, float* ms);

/* How the flank scan (searcher.rs:438) of group `group` ran on the last batch: kind 0 = full-height scan of every column, 1 = 15/31-row
 * filter + full-height verification around the flagged 16-byte pieces, 2 = the filter flagged more than the break-even
 * fraction of the batch's pieces (low-complexity text, adapter-like decoys), so the full scan did the batch, 3 = the full scan
 * without a filter pass: the group's last probed batch was of kind 2 (sixteen batches, then the group is probed again;
 * the counts stay those of the probed batch).  Results are the same whichever ran; the counts tell a caller how far its data is from the
 * synthetic benchmark's.                                                                                                  */
int bb_last_scan_stats(const bb_ctx* ctx, uint32_t group, uint64_t* flagged_pieces, uint64_t* total_pieces, int* kind);

/* Twin filter windows.  The right-hand query group of a dual-end kit is (give or take a few bases at the ends) the reverse complement of
 * the left-hand one: searching group B on the forward strand and group A on the reverse-complement strand look at the same text for the
 * same rows.  Where bb_finalize finds R rows of B's flank that are A's filter window reverse-complemented, it lays B's window there, and a
 * batch in which both groups take the filter pass runs ONE pass for the two (B's verification reads A's flags with the strands swapped).
 * The same holds, without the swap, for groups of one kit that share most of their flank (SQK-RBK114-96 --use-extended: two groups that
 * differ in their first 16 nt) when their windows can lie on the same rows.
 * *twin_of = that group A or -1; *shared = 1 if the last batch's scan of `group` read its twin's flags with the strands swapped, 2 if as
 * they are, 0 if the group ran its own pass (or none).  Results do not depend on it.                                                  */
int bb_filter_twin(const bb_ctx* ctx, uint32_t group, int* twin_of, int* shared);

/* The read lengths of the last batch as the scans saw them: the shortest and the longest read in 128-byte lines, and the number of work
 * items the scans' lanes drew — the number of reads for a batch of (nearly) equal reads (lanes take reads in file order), otherwise the
 * number of SEGMENTS: reads of more than 64 lines are cut into segments of 32 for the filter pass and lanes take segments / reads by
 * falling length, so that a wave's lanes finish together and a 100 kb read is not one lane's work (real runs are heavy-tailed; the
 * reference's threads take reads one by one, annotator.rs:123-135, and never meet the problem).  Results do not depend on it.     */
int bb_last_length_stats(const bb_ctx* ctx, uint32_t* min_lines, uint32_t* max_lines, uint32_t* work_items);

/* How often the host WAITED for the device inside the last bb_annotate_batch / bb_annotate_batch_dev call (stream synchronisations and blocking
 * copies): the part of a call's cost that does not shrink with the batch.  A caller that hands over small batches (paraseq's ~1 k records,
 * annotator.rs:278-280) pays it per call — INTEGRATION.md, "Batch size".                                                                   */
int bb_last_host_syncs(const bb_ctx* ctx);
/* Where the host-pointer form spends its wall time (diagnostics; off by default).  `enable` switches the clock on or off for the calls that follow;
 * ms[0..BB_N_HOST_PHASES) and *calls (either may be NULL) receive what was summed since the last call of this function, then the sums start over:
 * 0 upload enqueued (incl. HIP's copy of pageable memory), 1 the whole device pipeline, 2 rows back; inside 1: 3 read lengths, 4 flank scans up to
 * the hit count's round trip, 5 traceback .. barcode stage .. collapse up to the row count's round trip, 6 rows emitted and the stream drained.   */
#define BB_N_HOST_PHASES 7
int bb_host_phases(bb_ctx* ctx, int enable, double* ms, uint64_t* calls);

/* The barcode stage of the last batch, per (group, strand): flank hits listed for it, how many of them the fast kernel's bounds left
 * undecided (those are scored exactly by the second pass), and whether the pair's NEXT batch takes the one-lane-per-hit kernel
 * (k_barcode_lane: its bound assumes the shared pad rows match; above 20 % undecided the pair goes back to k_barcode_pfx for 32
 * batches; a batch of up to 4 096 reads takes one lane per (hit, barcode) whatever this says: one lane per hit is a long walk that only pays
 * when there are hits enough to fill the GPU with it).  Counts are only collected in the default kernel choice (BARBELL_AMD_LANE=1).
 * Rows do not depend on any of this.                                                                                                */
int bb_last_barcode_stats(const bb_ctx* ctx, uint32_t group, uint32_t strand, uint64_t* hits, uint64_t* undecided, int* lane_kernel);

/* Device buffers for callers of the *_dev entry points that have no HIP binding of their own (the Rust
 * or C++ host): memory on the context's GPU, and copies ordered after the context's stream.         */
int  bb_dev_malloc(bb_ctx* ctx, uint64_t bytes, void** d_ptr);
void bb_dev_free(bb_ctx* ctx, void* d_ptr);
int  bb_dev_download(bb_ctx* ctx, void* dst_host, const void* d_src, uint64_t bytes);
int  bb_dev_upload(bb_ctx* ctx, void* d_dst, const void* src_host, uint64_t bytes);
/* Page-locked host memory: buffers handed to the host-pointer entry points (FASTQ text blocks, rendered
 * records) move over PCIe at full rate when they come from here.                                    */
int  bb_host_malloc(bb_ctx* ctx, uint64_t bytes, void** ptr);
void bb_host_free(bb_ctx* ctx, void* ptr);
/* The same without a context (a host that stages its first FASTQ blocks while its contexts are still being created): page-locked memory
 * for uploads to `device`.  BB_E_NO_DEVICE without a usable GPU. */
int  bb_host_malloc_on(int device, uint64_t bytes, void** ptr);
void bb_host_free_on(int device, void* ptr);

/* Which classes of traceback orders (policy field `trace`, barbell_amd_policy.h) this build holds fast barcode kernels for: bit i = class i of
 * barbell_amd/csrc/bb_prio.h (bit 0 = the default order M,I,S,D).  The default build holds the classes the reference's own known-answer tests
 * (cigar_parse.rs:163-176) leave open; a context created under another order computes the same rows with the kernels that read the order at run
 * time and says so in bb_last_error(ctx).  Needs no GPU.                                                                                   */
uint32_t    bb_build_trace_classes(void);

const char* bb_strerror(int code);
const char* bb_last_error(const bb_ctx* ctx);   /* detail of the last BB_E_HIP / BB_E_UNSUPPORTED; ctx == NULL: of the
                                                   last failed bb_create on this thread                          */

#ifdef __cplusplus
}
#endif
#endif /* BARBELL_AMD_H */
