// bb_host.hpp — C++ host side above the C-ABI, mirroring the reference's annotate interface
// (names, argument meaning, defaults, error behaviour) for the hot path only:
//   BarcodeType / BarcodeGroup        src/annotate/barcodes.rs:8-33, 57-72, 251-320
//   Demuxer::new / add_query_group    src/annotate/searcher.rs:202-226
//   BarbellMatch + TSV serialisation  src/annotate/searcher.rs:31-142, annotator.rs:13-26
//   annotate / annotate_with_kit / annotate_with_files / annotate_with_groups   annotator.rs:155-285
//   kit presets                       src/kits/kits.rs:635-816, 1074-1103 (data in kits_data.inc)
//   Cut / PatternElement / Pattern / pattern_from_str!   src/filter/pattern.rs:9-30, 69-95, 242-383
//   filter pattern files, kit default patterns            src/filter/filter.rs:141-181, kits.rs:175-236
// The per-read call `Demuxer::demux(read_id, read)` (searcher.rs:430) becomes `demux_batch`: one
// bb_annotate_batch per batch.  All arithmetic of the path runs in the HIP kernels of
// libbarbell_amd.so; nothing here computes alignments.
#pragma once
#include <cstdint>
#include <optional>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../../include/barbell_amd.h"
#include "../../../include/barbell_amd_filter.h"

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
    bool has_filter_ = false;
};

struct AnnotateConfig {  // config.rs:3-12 with the CLI defaults of bin/main.rs:64-112
    std::optional<size_t> max_flank_errors;
    float alpha = 0.4f;
    unsigned n_threads = 10;  // accepted for CLI compatibility; the GPU path uses one host thread per context
    bool verbose = false;
    double min_score = 0.2, min_score_diff = 0.1;
    bool use_extended = false;
    size_t batch_reads = 65536;
    int device = 0;
    // fused filter step: when filter_patterns is non-empty, rows of passing / failing reads go to
    // filtered_file / dropped_file with their cuts column (what `barbell filter -o/--dropped` writes)
    std::vector<Pattern> filter_patterns;
    std::string filtered_file, dropped_file;
};
struct AnnotateStats { size_t total = 0, found = 0, rows = 0, kept = 0, dropped = 0; };

AnnotateStats annotate(const std::vector<std::string>& read_files, const std::string& out_file,
                       std::vector<BarcodeGroup> query_groups, const AnnotateConfig& config);          // annotator.rs:233-285
AnnotateStats annotate_with_groups(const std::vector<std::string>& read_files, const std::string& out_file,
                                   std::vector<BarcodeGroup> query_groups, const AnnotateConfig& config);  // :207-231
AnnotateStats annotate_with_kit(const std::vector<std::string>& read_files, const std::string& out_file,
                                const std::string& kit, const AnnotateConfig& config);                     // :196-204
AnnotateStats annotate_with_files(const std::vector<std::string>& read_files, const std::vector<std::string>& query_files,
                                  const std::vector<BarcodeType>& query_types, const std::string& out_file,
                                  const AnnotateConfig& config);                                            // :155-193

}  // namespace barbell
