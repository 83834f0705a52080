// bb_host.hpp — C++ host side above the C-ABI, mirroring the reference's annotate interface
// (names, argument meaning, defaults, error behaviour) for the hot path only:
//   BarcodeType / BarcodeGroup        src/annotate/barcodes.rs:8-33, 57-72, 251-320
//   Demuxer::new / add_query_group    src/annotate/searcher.rs:202-226
//   BarbellMatch + TSV serialisation  src/annotate/searcher.rs:31-142, annotator.rs:13-26
//   annotate / annotate_with_kit / annotate_with_files / annotate_with_groups   annotator.rs:155-285
//   kit presets                       src/kits/kits.rs:635-816, 1074-1103 (data in kits_data.inc)
//   Cut / PatternElement / Pattern / pattern_from_str!   src/filter/pattern.rs:9-30, 69-95, 242-383
//   filter pattern files, kit default patterns            src/filter/filter.rs:141-181, kits.rs:175-236
//   TrimConfig / LabelSide / label + file naming          src/config.rs:19-32, src/trim/trim.rs:24-105, 317-480
//   get_group_structure strings, inspect summary          src/inspect/inspect.rs:15-208
//   demux_using_kit                                        src/kits/use_kit.rs:11-109
//   FASTQ record loop -> GPU block ingest                  src/io/io.rs:6-33, annotator.rs:245-262, trim.rs:364-384
// The per-read call `Demuxer::demux(read_id, read)` (searcher.rs:430) becomes `demux_batch`: one
// bb_annotate_batch per batch.  All arithmetic of the path runs in the HIP kernels of
// libbarbell_amd.so; nothing here computes alignments.
// Units (round 6): bb_host.cpp (kits, patterns, Demuxer), bb_feed.{hpp,cpp} (FASTQ files -> blocks), bb_inflate.{hpp,cpp} (gzip), bb_writers.{hpp,cpp}
// (per-label FASTQ files), bb_annotate.cpp (the block pipeline, annotate*, demux_using_kit), bb_rccl.cpp + bb_rendezvous.{hpp,cpp}, bb_steps.cpp.
#pragma once
#include <cstdint>
#include <map>
#include <mutex>
#include <memory>
#include <mutex>
#include <optional>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../../include/barbell_amd.h"
#include "../../../include/barbell_amd_fastq.h"
#include "../../../include/barbell_amd_filter.h"
#include "../../../include/barbell_amd_format.h"
#include "../../../include/barbell_amd_inspect.h"
#include "../../../include/barbell_amd_trim.h"

namespace barbell {

enum class BarcodeType { Ftag, Rtag, Fflank, Rflank };
const char* as_str(BarcodeType t);  // barcodes.rs:25-32

struct BarbellError : std::runtime_error {
    int code;
    BarbellError(int c, const std::string& what) : std::runtime_error(what), code(c) {}
};

// What BarcodeGroup::new receives (barcodes.rs:106-110); geometry is derived inside bb_create.
struct BarcodeGroup {
    std::vector<std::string> seqs, labels;
    BarcodeType barcode_type = BarcodeType::Ftag;
    std::optional<size_t> k_cutoff;
    void set_flank_threshold(size_t k) { k_cutoff = k; }                                  // barcodes.rs:318-320
    static std::vector<BarcodeGroup> new_from_kit(const std::string& kit, bool also_use_extended);  // barcodes.rs:251-299
    static BarcodeGroup new_from_fasta(const std::string& fasta_file, BarcodeType bar_type);         // barcodes.rs:302-315
};

// kits.rs:741-816, 1074-1103
std::vector<std::string> get_barcodes(const std::string& from_label, const std::string& to_label, bool use_12a_flag);
const char* lookup_barcode_seq(const std::string& label);
std::vector<std::string> supported_kits();

// ---- filter patterns (src/filter/pattern.rs) ------------------------------------------------------
struct Cut {  // pattern.rs:15-19
    size_t group_id = 0;
    bool after = false;                                                  // CutDirection::After / Before
    static std::optional<Cut> from_pattern_string(const std::string& s);  // ">>", "<<", ">>3"  (pattern.rs:69-85)
    std::string to_string() const;                                       // "After(0)"  (pattern.rs:88-95)
};
struct PatternElement {  // pattern.rs:21-30
    BarcodeType match_type = BarcodeType::Ftag;
    int orientation = -1;                      // -1 any, BB_FWD, BB_RC
    std::optional<std::string> label;          // exact, or "~substr"
    int placeholder = -1;                      // ?N
    long range_lo = 0, range_hi = 0;
    int relative_to = BB_REL_NONE;             // @left / @right / @prev_left
    std::vector<Cut> cuts;
};
struct Pattern { std::vector<PatternElement> elements; };
Pattern pattern_from_str(const std::string& s);                                   // pattern_from_str! (pattern.rs:242-383)
std::vector<Pattern> patterns_from_files(const std::vector<std::string>& paths);   // filter.rs:137-181
std::vector<Pattern> kit_patterns(const std::string& kit, bool maximize);         // kits.rs:175-236 (safe unless maximize)

// ---- trim (src/trim/trim.rs, src/config.rs) -------------------------------------------------------
enum class LabelSide { Left, Right };
struct TrimConfig {  // config.rs:19-32, CLI defaults bin/main.rs:136-186
    bool add_labels = true, add_orientation = true, add_flank = true, sort_labels = false;
    std::optional<LabelSide> only_side;
    std::optional<std::string> failed_trimmed_writer;
    bool write_full_header = true, skip_trim = false, flip = false, verbose = false, gzip = false;
    static TrimConfig for_kit(std::optional<std::string> failed_out, bool gzip);  // use_kit.rs:87-99
};
struct TrimBatch {  // what bb_trim_batch returns for one batch
    std::vector<uint8_t> text;        // trim_last_batch: the rendered records
    const uint8_t* text_ptr = nullptr; // trim_ingested: the records in a page-locked buffer of the demuxer's pool
    std::shared_ptr<void> text_hold;   // ... which goes back to the pool when the last holder (the file writers) lets go
    uint64_t text_len = 0;
    const uint8_t* data() const { return text_ptr ? text_ptr : text.data(); }
    std::vector<bb_slice> slices;
    std::vector<bb_label_span> spans;
    std::vector<uint8_t> status;  // BB_TRIM_* per read
};
// trim_plan_ingested: everything the trim step decides, nothing it copies — for a caller that still holds the block's FASTQ text
// (the CLI: the page-locked slot the block was uploaded from).  render_trim_record cuts one record out of that text.
struct TrimPlan {
    std::vector<bb_slice> slices;       // text order: grouped by label, read order inside a label; out_off / rec_len set
    std::vector<bb_label_span> spans;
    std::vector<uint8_t> status;        // BB_TRIM_* per read
    std::vector<uint64_t> line_ends;    // bb_fastq_fetch_lines: 4 per record
    std::vector<uint32_t> id_len, desc_start;
    uint64_t text_len = 0;              // bytes of all records
};
// the record of `s` exactly as bb_trim_batch renders it (trim.rs:447-460), cut out of the block's text; returns its length (== s.rec_len)
size_t render_trim_record(uint8_t* dst, const uint8_t* block_text, const TrimPlan& plan, const bb_slice& s, const bb_trim_config& cfg);

struct FastqBatch {  // one batch of records as the C-ABI wants them
    std::vector<std::string> ids;
    std::vector<uint8_t> bases, quals, hdr;
    std::vector<uint64_t> offsets{0}, hdr_offsets{0};
    std::vector<uint32_t> id_len, desc_start;
    void clear();
};

struct BarbellMatch {  // searcher.rs:31-64
    std::string read_id;
    size_t read_len;
    long rel_dist_to_end;
    size_t read_start_bar, read_end_bar, read_start_flank, read_end_flank, bar_start, bar_end;
    BarcodeType match_type;
    int flank_cost, barcode_cost;
    std::string label;
    bool strand_rc;
    std::string cuts;             // "After(0):1,Before(0):2" once the filter step ran (searcher.rs:91-106), else ""
    std::string to_tsv() const;   // csv-crate row, tab-delimited
};
extern const char* const TSV_HEADER;

// A device allocation owned through the C-ABI (bb_dev_malloc / bb_dev_free); grows, never shrinks.
struct DevBuf {
    bb_ctx* ctx = nullptr;
    void* p = nullptr;
    uint64_t cap = 0;
    void ensure(bb_ctx* c, uint64_t bytes);
    void release();
    ~DevBuf() { release(); }
};

class Demuxer {
public:
    Demuxer(float alpha, bool verbose, double min_score_frac, double min_score_diff_frac, int device = 0);  // searcher.rs:202
    ~Demuxer();
    Demuxer(const Demuxer&) = delete;
    Demuxer& operator=(const Demuxer&) = delete;
    Demuxer& add_query_group(BarcodeGroup g);                                                               // searcher.rs:220
    // rows of all reads of the batch, in input order (rows of one read contiguous, sorted by flank start)
    std::vector<BarbellMatch> demux_batch(const std::vector<std::string>& read_ids, const std::vector<uint8_t>& bases,
                                          const std::vector<uint64_t>& offsets);
    // filter step on the rows of the LAST demux_batch (filter.rs:183-214 on the GPU): one verdict per row
    void set_filter(const std::vector<Pattern>& patterns);
    std::vector<bb_row_verdict> filter_last_batch();
    // trim step (trim.rs:127-300 on the GPU) on the rows + verdicts of the LAST batch; needs set_filter
    void set_trim(const TrimConfig& cfg);
    TrimBatch trim_last_batch(const std::vector<bb_row_verdict>& verdicts, const FastqBatch& batch);
    // rows parsed back from an annotation file take the place of the last batch's (bb_steps.cpp: the steps on files)
    void load_rows(const std::vector<bb_row>& rows);
    std::string label_of_key(uint32_t key) const;  // LabelConfig::create_label's string for a bb_slice.label_key
    // inspect step (inspect.rs:15-117) on the rows of the LAST batch: one pattern string per read with rows
    std::vector<std::pair<uint32_t, std::string>> inspect_last_batch(const std::vector<bb_row_verdict>* verdicts, uint32_t bucket_size);
    // ---- device-resident path: one block of raw FASTQ text -> everything else happens in HBM ----------
    // ingest() parses the block on the GPU (bb_fastq_ingest) and fetches only the headers; the *_ingested
    // calls run on the batch it left in HBM and download rows / verdicts / rendered text.
    struct Ingested { bb_fastq_info info{}; std::vector<std::string> ids; };
    Ingested ingest(const uint8_t* text, uint64_t len, bool final_block, bool want_ids = true, bool two_line = false, bool packed = false);
    // annotate the ingested batch; rows stay in HBM, their POD copy is rows() (no strings are built)
    uint64_t annotate_ingested();
    const bb_row* rows() const { return rows_.data(); }
    uint64_t n_rows() const { return n_rows_; }
    // TSV lines of the last batch's rows rendered on the GPU (bb_format_rows_dev): appended to `out`, returns the line count
    uint64_t format_ingested(int mode, std::vector<uint8_t>& out);
    // per-(group, barcode | flank) histogram of this context (bb_counts) and its labels, slot order
    std::vector<uint64_t> counts();
    std::vector<std::string> slot_labels() const;
    int device() const { return device_; }
    bb_ctx* ctx() { ensure_ctx(); return ctx_; }  // for page-locked block buffers (bb_host_malloc)
    std::vector<BarbellMatch> demux_ingested();
    std::vector<bb_row_verdict> filter_ingested();
    TrimBatch trim_ingested();
    TrimPlan trim_plan_ingested();   // bb_trim_plan_dev + the block's line ends: no record text crosses PCIe
    bb_trim_config trim_config_pod() const;
    std::vector<std::pair<uint32_t, std::string>> inspect_ingested(bool with_verdicts, uint32_t bucket_size);
    void inspect_ingested_interned(bool with_verdicts, uint32_t bucket_size, std::vector<std::string>& patterns,
                                   std::vector<std::pair<uint32_t, uint32_t>>& per_read);  // per_read: (read, index into patterns)
    bb_group_info group_info(size_t g);
    const std::vector<BarcodeGroup>& queries() const { return queries_; }

private:
    void ensure_ctx();
    float alpha_;
    bool verbose_;
    double min_score_, min_score_diff_;
    int device_;
    std::vector<BarcodeGroup> queries_;
    bb_ctx* ctx_ = nullptr;
    std::vector<bb_row> rows_;
    uint64_t n_rows_ = 0;
    bool has_filter_ = false, has_trim_ = false;
    TrimConfig trim_cfg_;
    std::vector<std::string> label_strings_;  // by label id (ids as handed to bb_filter_set)
    std::string part_str(uint32_t part) const;
    std::vector<std::pair<uint32_t, std::string>> elems_to_patterns(const std::vector<bb_inspect_elem>& el) const;
    std::vector<BarbellMatch> rows_to_matches(const std::vector<std::string>& read_ids) const;
    bb_fastq_batch_dev batch_{};
    Ingested ing_;
    DevBuf d_rows_, d_ver_, d_elems_, d_text_, d_slices_, d_spans_, d_status_, d_tsv_;
    // page-locked landing buffers of the rendered records (bb_host_malloc), used alternately so a writer thread
    // can still be flushing one batch's records while the next batch is downloaded
    // A pool: the writer threads may still be flushing several batches' records while the next one is downloaded; a buffer returns
    // when its last holder lets go (no copy of the rendered text on the host).
    struct TextPool {
        std::mutex mu;
        std::vector<std::pair<uint8_t*, uint64_t>> free_;
        std::vector<uint8_t*> all;
    };
    std::shared_ptr<TextPool> text_pool_ = std::make_shared<TextPool>();
};

struct AnnotateConfig {  // config.rs:3-12 with the CLI defaults of bin/main.rs:64-112
    std::optional<size_t> max_flank_errors;
    float alpha = 0.4f;
    unsigned n_threads = 0;   // -t: reader / inflater threads of the input feeder (the reference's worker threads, default 10 there); 0 = as many as the
                              // process may use, at most 32 (measured on the 16-CPU box, 4 M reads: -t 4 7.7, 10 11–12, 16 14.6, 32 13.5 M reads/s)
    bool verbose = false;
    double min_score = 0.2, min_score_diff = 0.1;
    bool use_extended = false;
    size_t batch_reads = 0;               // if set: block_bytes = batch_reads * 4096 (kept for CLI compatibility)
    size_t block_bytes = 256u << 20;      // raw FASTQ text handed to the GPU per ingest call (round 5: 256 MiB — a block costs ~2.5 ms of fixed work on its
                                          // context whatever its size; 128 MiB blocks held the pipeline at 11.1 M reads/s, 256 MiB at 11.8 M).  Page-locked memory: 3 slots per context + 2 (twice
                                          // that when the quality lines are dropped on the host), each block_bytes + 16 MiB or the largest input
                                          // file if that is smaller — 1.2 GiB at the defaults (2 contexts), 6 GiB at --streams 3 --block-bytes 256Mi;
                                          // with the trim step and host_cut six slots more (blocks wait in them for the file writers)
    bool compact_upload = true;           // without the trim step: drop the '+' and quality lines on the host (half the PCIe bytes); --no-compact
    bool pack_upload = true;              // ... and stage the sequence lines two bases per byte (BB_FASTQ_PACKED: a quarter of the PCIe bytes; the kernels
                                          // only look at a character's IUPAC base set); --no-pack.  Falls back by itself where the form cannot hold the input
    bool host_cut = true;                 // trim step: the GPU plans (slices, labels, offsets), the threads that write the per-label files cut the
                                          // records out of the block's own page-locked text — the rendered records (as many bytes as went up) do
                                          // not come back over PCIe.  false (--gpu-render): bb_trim_batch_dev renders them in HBM and they are downloaded
    int device = 0;
    // One FASTQ stream over several contexts: block i of the stream goes to context i mod G, rows are merged in
    // block order, the per-barcode histogram is all-reduced (RCCL when the devices are distinct).  Empty = {device}
    // repeated streams_per_device times; a device may appear more than once (several contexts on one GPU overlap the
    // upload of one block with the kernels of another).
    std::vector<int> devices;
    unsigned streams_per_device = 2;
    std::string counts_file;              // label <TAB> count of the all-reduced histogram
    // one process per GPU (`--shard R/W`): with rccl_id set, the W processes all-reduce their histograms (ncclCommInitRank, the unique id
    // published through files next to rccl_id; through the files alone where processes share a device) and shard 0 writes counts_file;
    // without it every process keeps its own counts (a warning says so)
    uint32_t shard_rank = 0, shard_world = 1;
    std::string rccl_id;
    bool shard_by_bytes = false;          // --shard-by bytes: shard R of W takes the records that start in the R-th of W equal byte ranges of EVERY (plain)
                                          // input file — one big FASTQ over W processes; false: the files with index R modulo W
    // fused filter step: when filter_patterns is non-empty, rows of passing / failing reads go to
    // filtered_file / dropped_file with their cuts column (what `barbell filter -o/--dropped` writes)
    std::vector<Pattern> filter_patterns;
    std::string filtered_file, dropped_file;
    // fused trim step (needs the filter): per-label '{trim_folder}/{label}.trimmed.fastq[.gz]' (trim.rs:427-446)
    std::optional<TrimConfig> trim;
    std::string trim_folder;
    // fused inspect step: pattern_per_read.tsv + pattern counts of the annotation rows (inspect.rs:119-208)
    bool inspect = false;
    std::string read_pattern_out;
    uint32_t bucket_size = 250;
    // the caller's process ends right after this call (the CLI): the block buffers are left to the OS instead of being un-page-locked one by
    // one (0.4-0.7 s for the default 2-6 GB)
    bool process_exits_after = false;
};
struct AnnotateStats {
    size_t total = 0, found = 0, rows = 0, kept = 0, dropped = 0, trimmed = 0, trimmed_split = 0, trim_failed = 0;
    std::vector<std::pair<std::string, size_t>> patterns;  // inspect: pattern -> count, most common first
    std::vector<std::pair<std::string, uint64_t>> counts;  // (label, rows) summed over all contexts, slot order
    std::string counts_reduce;                              // "rccl" | "host" | "single": how the histogram was summed
    double seconds_pipeline = 0;                            // first block read .. last block committed (steady state, no start-up)
};
unsigned effective_cpus();   // CPUs the process can keep busy: affinity mask cut by the cgroup's CPU quota (BARBELL_AMD_CPUS overrides)
int stage_blocks(const std::vector<std::string>& read_files, size_t block_bytes, unsigned n_threads, bool two_line, bool pack, const std::string& out_path,
                 size_t& n_blocks, uint32_t byte_shard_rank = 0, uint32_t byte_shard_world = 1);   // `barbell-amd stage`: what the host stages for upload, to a file (no GPU); returns 4 / 2 / 1 (packed)
void shard_rendezvous_reset(const std::string& rccl_id, uint32_t rank);   // bb_rendezvous.cpp (Rendezvous::hello); call at program start of a --shard R/W --rccl-id run
// the histogram summed over the W processes through the rendezvous files alone, no GPU (`barbell-amd rendezvous`: tests/test_rendezvous.py)
std::vector<uint64_t> rendezvous_sum_counts(const std::string& rccl_id, uint32_t rank, uint32_t world, const std::string& bus, const std::vector<uint64_t>& local,
                                            bool* shared_device = nullptr);
std::vector<std::string> inspect_summary(const AnnotateStats& st, size_t top_n);  // the lines of inspect.rs:186-205

struct KitConfig {  // config.rs:34-48, CLI defaults bin/main.rs:208-262
    std::string kit_name, output_folder;
    size_t threads = 0;   // AnnotateConfig::n_threads
    bool maximize = false, verbose = false;
    double min_score = 0.2, min_score_diff = 0.1;
    std::optional<size_t> max_flank_errors;
    std::optional<std::string> failed_out;
    bool use_extended = false;
    float alpha = 0.4f;
    bool gzip = false;
    bool host_cut = true;   // AnnotateConfig::host_cut
    size_t batch_reads = 0;
    int device = 0;
    std::vector<int> devices;
    unsigned streams_per_device = 2;
    std::string counts_file;
    uint32_t shard_rank = 0, shard_world = 1;   // AnnotateConfig::shard_rank / shard_world / rccl_id / shard_by_bytes
    std::string rccl_id;
    bool shard_by_bytes = false;
    bool process_exits_after = false;   // AnnotateConfig::process_exits_after
};
AnnotateStats demux_using_kit(const std::vector<std::string>& fastq_files, const KitConfig& config);  // use_kit.rs:11-109

AnnotateStats annotate(const std::vector<std::string>& read_files, const std::string& out_file,
                       std::vector<BarcodeGroup> query_groups, const AnnotateConfig& config);          // annotator.rs:233-285
AnnotateStats annotate_with_groups(const std::vector<std::string>& read_files, const std::string& out_file,
                                   std::vector<BarcodeGroup> query_groups, const AnnotateConfig& config);  // :207-231
AnnotateStats annotate_with_kit(const std::vector<std::string>& read_files, const std::string& out_file,
                                const std::string& kit, const AnnotateConfig& config);                     // :196-204
AnnotateStats annotate_with_files(const std::vector<std::string>& read_files, const std::vector<std::string>& query_files,
                                  const std::vector<BarcodeType>& query_types, const std::string& out_file,
                                  const AnnotateConfig& config);                                            // :155-193

// ---- the stand-alone steps on files (bb_steps.cpp): an annotation.tsv written earlier back through the same kernels ----
struct StepStats { size_t total = 0, kept = 0, dropped = 0, split = 0; };   // reads; trim: kept = trimmed, dropped = failed
StepStats filter_file(const std::string& annotated_file, const std::string& output_file, const std::optional<std::string>& dropped_out_file,
                      const std::vector<Pattern>& filters, int device = 0, size_t batch_rows = 1u << 18);                       // filter.rs:10-119
StepStats inspect_file(const std::string& annotated_file, const std::optional<std::string>& read_pattern_out, uint32_t bucket_size,
                       AnnotateStats& patterns, int device = 0, size_t batch_rows = 1u << 18);                                   // inspect.rs:119-208
StepStats trim_file(const std::string& filtered_match_file, const std::vector<std::string>& read_fastq_files, const std::string& output_folder,
                    const TrimConfig& config, int device = 0, size_t batch_reads = 20000);                                       // trim.rs:317-480

// what --verbose leaves behind (progress.rs:96-144): '{log_dir}/{step}.{unix ms}.log', "step\tmetric\tcount" + a line per counter; returns the path
std::string write_progress_log(const std::string& step, const std::string& log_dir, const std::vector<std::pair<std::string, size_t>>& counts);
std::string parent_dir(const std::string& file);

}  // namespace barbell
