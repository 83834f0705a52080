// bb_steps.cpp — the reference's stand-alone steps on FILES: `barbell filter` (src/filter/filter.rs:10-119), `barbell inspect`
// (src/inspect/inspect.rs:119-208), `barbell trim` (src/trim/trim.rs:317-480) on an annotation.tsv written earlier, by Barbell or by
// this host.  The reference documents four separate commands with files in between (README "In-depth inspection", "Custom experiment
// with mixed sequences": two filters on one annotation file, a trim per filtered file); the fused run (annotate with --kit-filter /
// --trim-output, `kit`) stays the fast way to process a run.
//
// The TSV is parsed back into bb_rows (searcher.rs:31-142 is the schema; the `cuts` column back into verdicts, searcher.rs:108-140), the
// same kernels decide through the host-buffer entry points (bb_filter_rows, bb_inspect_rows, bb_trim_batch), nothing here matches a
// pattern or cuts a read.  Labels map to histogram slots through stand-in query groups that only carry the file's label strings: these
// kernels read a row's slot and label id, never the query sequences (the reference's filter / inspect / trim need no queries either).
// Python twin: barbell_amd/steps.py (same behaviour; tests/test_steps.py compares the two and the fused run byte for byte).
#include <zlib.h>

#include <algorithm>
#include <cerrno>
#include <cstdio>
#include <cstring>
#include <ctime>
#include <filesystem>
#include <fstream>
#include <unordered_map>

#include "bb_host.hpp"

namespace barbell {
namespace {

const char* const COLS[15] = {"read_id", "read_len", "rel_dist_to_end", "read_start_bar", "read_end_bar", "read_start_flank", "read_end_flank",
                              "bar_start", "bar_end", "match_type", "flank_cost", "barcode_cost", "label", "strand", "cuts"};
enum { C_ID = 0, C_LEN, C_REL, C_SB, C_EB, C_SF, C_EF, C_BS, C_BE, C_MT, C_FC, C_BC, C_LAB, C_STRAND, C_CUTS };
constexpr size_t GROUP_MAX_LABELS = 1022;   // 1024 sequences per group (include/barbell_amd.h) minus the two stand-ins that pin the barcode region

[[noreturn]] void fail(const std::string& what) { throw BarbellError(BB_E_INVALID, what); }

// One record of a csv-crate file (tab-delimited; a field holding a tab, a quote, CR or LF is quoted, quotes doubled: annotator.rs:246-251).
// Returns false at the end of the file.
bool read_record(std::istream& in, std::vector<std::string>& rec, const std::string& path) {
    std::string line;
    if (!std::getline(in, line)) return false;
    while (std::count(line.begin(), line.end(), '"') % 2) {   // a quoted field with a line break in it
        std::string more;
        if (!std::getline(in, more)) fail(path + ": unterminated quoted field");
        line += "\n" + more;
    }
    if (!line.empty() && line.back() == '\r') line.pop_back();
    rec.clear();
    std::string cur;
    bool quoted = false;
    for (size_t i = 0; i < line.size(); ++i) {
        const char ch = line[i];
        if (quoted) {
            if (ch == '"') { if (i + 1 < line.size() && line[i + 1] == '"') { cur += '"'; ++i; } else quoted = false; }
            else cur += ch;
        } else if (ch == '"' && cur.empty()) quoted = true;
        else if (ch == '\t') { rec.push_back(std::move(cur)); cur.clear(); }
        else cur += ch;
    }
    rec.push_back(std::move(cur));
    return true;
}

struct Tsv {
    std::ifstream in;
    std::string path;
    int col[15];
    size_t ncol = 0;
    bool empty = false;   // the csv writer emits the header with the first record only: a run without rows leaves an EMPTY file
    explicit Tsv(const std::string& p) : in(p), path(p) {
        if (!in) fail("cannot open " + p);
        std::vector<std::string> h;
        if (!read_record(in, h, p)) { empty = true; return; }
        ncol = h.size();
        for (int c = 0; c < 15; ++c) {
            const auto it = std::find(h.begin(), h.end(), COLS[c]);
            if (it == h.end()) fail(p + ": column missing from the header: " + COLS[c]);
            col[c] = (int)(it - h.begin());
        }
    }
    // fields in the schema's order; false at the end
    bool next(std::vector<std::string>& f) {
        if (empty) return false;
        std::vector<std::string> rec;
        for (;;) {
            if (!read_record(in, rec, path)) return false;
            if (rec.size() == 1 && rec[0].empty()) continue;
            if (rec.size() != ncol) fail(path + ": a record of " + std::to_string(rec.size()) + " fields under a header of " + std::to_string(ncol));
            f.resize(15);
            for (int c = 0; c < 15; ++c) f[c] = rec[(size_t)col[c]];
            return true;
        }
    }
};

int match_type_of(const std::string& s, const std::string& path) {
    for (int t = 0; t < 4; ++t) if (s == as_str((BarcodeType)t)) return t;
    fail(path + ": match_type '" + s + "'");
}
inline int kind_of(int mt) { return mt == (int)BarcodeType::Rtag || mt == (int)BarcodeType::Rflank; }   // 0 = Ftag side, 1 = Rtag side

long long to_int(const std::string& s, const char* what, const std::string& path) {
    if (s.empty()) fail(path + ": " + what + ": empty");
    char* end = nullptr;
    errno = 0;
    const long long v = strtoll(s.c_str(), &end, 10);
    if (*end || errno == ERANGE) fail(path + ": " + what + ": '" + s + "' is not an integer");
    return v;
}

// "After(0):1,Before(2):1" -> the row's verdict (n_cuts, cuts, match_idx); searcher.rs:108-140, pattern.rs:40-66
void parse_cuts(const std::string& s, bb_row_verdict& v) {
    if (s.empty()) return;
    long pos = -1;
    size_t a = 0;
    while (a <= s.size()) {
        size_t b = s.find(',', a);
        if (b == std::string::npos) b = s.size();
        const std::string part = s.substr(a, b - a);
        const size_t colon = part.find(':');
        if (colon == std::string::npos) fail("Invalid cut format: missing position part");
        const std::string cut = part.substr(0, colon), p = part.substr(colon + 1);
        int dir;
        std::string gid;
        if (cut.rfind("Before(", 0) == 0 && cut.back() == ')') { dir = BB_CUT_BEFORE; gid = cut.substr(7, cut.size() - 8); }
        else if (cut.rfind("After(", 0) == 0 && cut.back() == ')') { dir = BB_CUT_AFTER; gid = cut.substr(6, cut.size() - 7); }
        else fail("Invalid cut string: " + cut);
        if (gid.empty() || p.empty() || gid.find_first_not_of("0123456789") != std::string::npos || p.find_first_not_of("0123456789") != std::string::npos)
            fail("Invalid cut string: " + part);
        if (p.size() > 5 || atol(p.c_str()) > 0xFFFF) fail("cut position above 65535: " + part);
        const long q = atol(p.c_str());
        if (pos >= 0 && q != pos) fail("cuts of one row with different positions (" + s + "): not what the filter step writes");
        pos = q;
        if (v.n_cuts >= BB_MAX_CUTS) fail("more than 3 cuts on one row (kernel limit, include/barbell_amd_filter.h)");
        if (atol(gid.c_str()) > 0xFFFF) fail("cut group id above 65535 (kernel limit, include/barbell_amd_filter.h)");
        v.cuts[v.n_cuts++] = bb_cut{(uint8_t)dir, 0, (uint16_t)atol(gid.c_str())};
        a = b + 1;
    }
    v.match_idx = (uint16_t)pos;
}

std::string format_cuts(const bb_row_verdict& v) {   // searcher.rs:91-106
    std::string s;
    for (int q = 0; q < v.n_cuts; ++q) {
        if (q) s += ',';
        s += (v.cuts[q].direction == BB_CUT_AFTER ? "After(" : "Before(") + std::to_string(v.cuts[q].group_id) + "):" + std::to_string(v.match_idx);
    }
    return s;
}

// ---- labels <-> slots --------------------------------------------------------------------------------------------------
// A 46-nt sequence of SQK-NBD114-96's shape (14 / 24 / 8) whose 24-nt barcode encodes i: never searched, only there so that bb_create
// has a well-formed group to hang the labels on.
std::string standin_seq(size_t i) {
    std::string bc = "G";
    for (int j = 0; j < 22; ++j) bc += "ACGT"[(i >> (2 * j)) & 3];
    return "AAGGTTAACACAAA" + bc + "C" + "CAGCACCT";
}

struct LabelSpace {
    std::vector<BarcodeGroup> groups;
    std::unordered_map<std::string, std::pair<int, int>> slot[2];   // per tag kind: label -> (group, barcode)
    int flank_group[2] = {-1, -1};

    // the distinct (match_type, label) pairs of the file, in order of first appearance
    static LabelSpace from_file(const std::string& path) {
        std::vector<std::string> labs[2];
        bool any[2] = {false, false};
        std::unordered_map<std::string, char> seen[2];
        Tsv t(path);
        std::vector<std::string> f;
        while (t.next(f)) {
            const int mt = match_type_of(f[C_MT], path), k = kind_of(mt);
            any[k] = true;
            if (mt == (int)BarcodeType::Fflank || mt == (int)BarcodeType::Rflank) continue;
            if (seen[k].emplace(f[C_LAB], 1).second) labs[k].push_back(f[C_LAB]);
        }
        LabelSpace sp;
        for (int k = 0; k < 2; ++k) {
            if (!any[k]) continue;
            sp.flank_group[k] = (int)sp.groups.size();
            for (size_t a = 0; a < std::max<size_t>(1, labs[k].size()); a += GROUP_MAX_LABELS) {
                BarcodeGroup g;
                g.barcode_type = k ? BarcodeType::Rtag : BarcodeType::Ftag;
                g.set_flank_threshold(0);
                const size_t e = std::min(labs[k].size(), a + GROUP_MAX_LABELS);
                for (size_t i = a; i < e; ++i) {
                    sp.slot[k].emplace(labs[k][i], std::make_pair((int)sp.groups.size(), (int)(i - a)));
                    g.seqs.push_back(standin_seq(i - a));
                    g.labels.push_back(labs[k][i]);
                }
                g.seqs.push_back("AAGGTTAACACAAA" + std::string(24, 'A') + "CAGCACCT"); g.labels.push_back(std::string("\0barbell-amd stand-in A", 23));
                g.seqs.push_back("AAGGTTAACACAAA" + std::string(24, 'T') + "CAGCACCT"); g.labels.push_back(std::string("\0barbell-amd stand-in T", 23));
                sp.groups.push_back(std::move(g));
            }
        }
        if (sp.groups.size() > 32)
            fail(std::to_string(labs[0].size() + labs[1].size()) + " distinct labels need " + std::to_string(sp.groups.size()) +
                 " stand-in groups; a context holds 32 (include/barbell_amd.h)");
        return sp;
    }
    std::pair<int, int> lookup(int mt, const std::string& label, const std::string& path) const {
        const int k = kind_of(mt);
        if (mt == (int)BarcodeType::Fflank || mt == (int)BarcodeType::Rflank) return {flank_group[k], -1};
        const auto it = slot[k].find(label);
        if (it == slot[k].end()) fail(path + ": label '" + label + "' not seen by the label scan");
        return it->second;
    }
};

// a column's value inside its field of the row record (the Python twin raises TsvError "outside the range of the row record")
long long in_range(const std::string& text, const char* what, const std::string& path, long long lo, long long hi) {
    const long long v = to_int(text, what, path);
    if (v < lo || v > hi) fail(path + ": " + what + ": " + text + " is outside the range of the row record");
    return v;
}
bb_row make_row(const std::vector<std::string>& f, uint32_t read_idx, const LabelSpace& sp, const std::string& path) {
    bb_row r{};
    const long long U32 = 0xFFFFFFFFll;
    r.read_idx = read_idx;
    r.read_len = (uint32_t)in_range(f[C_LEN], "read_len", path, 0, U32);
    r.rel_dist_to_end = (int32_t)in_range(f[C_REL], "rel_dist_to_end", path, -0x80000000ll, 0x7FFFFFFFll);
    r.read_start_bar = (uint32_t)in_range(f[C_SB], "read_start_bar", path, 0, U32); r.read_end_bar = (uint32_t)in_range(f[C_EB], "read_end_bar", path, 0, U32);
    r.read_start_flank = (uint32_t)in_range(f[C_SF], "read_start_flank", path, 0, U32); r.read_end_flank = (uint32_t)in_range(f[C_EF], "read_end_flank", path, 0, U32);
    r.bar_start = (uint32_t)in_range(f[C_BS], "bar_start", path, 0, U32); r.bar_end = (uint32_t)in_range(f[C_BE], "bar_end", path, 0, U32);
    r.flank_cost = (int16_t)in_range(f[C_FC], "flank_cost", path, -32768, 32767); r.barcode_cost = (int16_t)in_range(f[C_BC], "barcode_cost", path, -32768, 32767);
    const int mt = match_type_of(f[C_MT], path);
    const auto s = sp.lookup(mt, f[C_LAB], path);
    r.group_idx = (decltype(r.group_idx))s.first; r.barcode_idx = (int16_t)s.second; r.match_type = (uint8_t)mt;
    if (f[C_STRAND] == "Fwd") r.strand = BB_FWD;
    else if (f[C_STRAND] == "Rc") r.strand = BB_RC;
    else fail(path + ": Invalid strand: " + f[C_STRAND]);
    return r;
}

std::string csv_quote(const std::string& x) {  // csv crate, QuoteStyle::Necessary
    if (x.find_first_of("\t\"\n\r") == std::string::npos) return x;
    std::string q = "\"";
    for (char c : x) { if (c == '"') q += '"'; q += c; }
    return q + "\"";
}
// the record as the csv writer serialises it again: the fields as they stand in the file, the cuts replaced
std::string line_of(const std::vector<std::string>& f, const std::string& cuts) {
    std::string s = csv_quote(f[C_ID]);
    for (int c = 1; c < 14; ++c) { s += '\t'; s += c == C_LAB ? csv_quote(f[c]) : f[c]; }
    s += '\t'; s += cuts; s += '\n';
    return s;
}

std::unique_ptr<Demuxer> context_for(const LabelSpace& sp, int device, const std::vector<Pattern>& patterns) {
    auto dm = std::make_unique<Demuxer>(0.4f, false, 0.2, 0.1, device);
    for (const auto& g : sp.groups) dm->add_query_group(g);
    dm->set_filter(patterns);   // also for inspect / trim: the label ids of the trim step come from it
    return dm;
}

// Batches of consecutive reads (consecutive lines of one read_id: filter.rs:52-85), a read never split over two batches.
struct BatchReader {
    Tsv t;
    const LabelSpace& sp;
    size_t batch_rows;
    std::vector<std::string> pending;   // the first record of the next batch
    bool has_pending = false, done = false;
    BatchReader(const std::string& path, const LabelSpace& s, size_t rows) : t(path), sp(s), batch_rows(rows) {}
    bool next(std::vector<bb_row>& rows, std::vector<bb_row_verdict>& ver, std::vector<std::vector<std::string>>& fields, std::vector<std::string>& ids, bool& has_cuts) {
        rows.clear(); ver.clear(); fields.clear(); ids.clear(); has_cuts = false;
        if (done) return false;
        std::vector<std::string> f;
        for (;;) {
            if (has_pending) { f = std::move(pending); has_pending = false; }
            else if (!t.next(f)) { done = true; break; }
            if (rows.size() >= batch_rows && f[C_ID] != ids.back()) { pending = std::move(f); has_pending = true; break; }
            if (ids.empty() || f[C_ID] != ids.back()) ids.push_back(f[C_ID]);
            rows.push_back(make_row(f, (uint32_t)(ids.size() - 1), sp, t.path));
            bb_row_verdict v{};
            v.pass = 1;
            parse_cuts(f[C_CUTS], v);
            has_cuts |= v.n_cuts != 0;
            ver.push_back(v);
            fields.push_back(f);
        }
        // the position of a row without cuts: its index among its read's rows (what k_filter writes for every row)
        for (size_t i = 0, start = 0; i < rows.size(); ++i) {
            if (i && rows[i].read_idx != rows[i - 1].read_idx) start = i;
            if (!ver[i].n_cuts) ver[i].match_idx = (uint16_t)(i - start);
        }
        return !rows.empty();
    }
};

// id up to the first whitespace, description left-trimmed (io.rs:6-17; char::is_whitespace over UTF-8)
size_t ws_len(const uint8_t* p, size_t n) {
    if (!n) return 0;
    if (p[0] == ' ' || (p[0] >= 9 && p[0] <= 13)) return 1;
    if (n >= 2 && p[0] == 0xC2 && (p[1] == 0x85 || p[1] == 0xA0)) return 2;
    if (n >= 3 && p[0] == 0xE1 && p[1] == 0x9A && p[2] == 0x80) return 3;                                      // U+1680
    if (n >= 3 && p[0] == 0xE2 && p[1] == 0x80 && ((p[2] >= 0x80 && p[2] <= 0x8A) || p[2] == 0xA8 || p[2] == 0xA9 || p[2] == 0xAF)) return 3;   // U+2000-200A, 2028, 2029, 202F
    if (n >= 3 && p[0] == 0xE2 && p[1] == 0x81 && p[2] == 0x9F) return 3;                                      // U+205F
    if (n >= 3 && p[0] == 0xE3 && p[1] == 0x80 && p[2] == 0x80) return 3;                                      // U+3000
    return 0;
}
void split_header(const std::string& h, uint32_t& id_len, uint32_t& desc_start) {
    const uint8_t* p = (const uint8_t*)h.data();
    size_t i = 0;
    while (i < h.size() && !ws_len(p + i, h.size() - i)) ++i;
    id_len = (uint32_t)i;
    for (size_t w; i < h.size() && (w = ws_len(p + i, h.size() - i)) != 0; i += w) {}
    desc_start = (uint32_t)i;
}

// plain or gzip FASTQ, 4-line records (gzread passes plain files through)
struct FastqReader {
    gzFile f;
    std::string path;
    std::vector<char> buf;
    explicit FastqReader(const std::string& p) : f(gzopen(p.c_str(), "rb")), path(p), buf(1 << 20) {
        if (!f) fail("Failed to open FASTQ file '" + p + "'");
        gzbuffer(f, 1 << 20);
    }
    ~FastqReader() { if (f) gzclose(f); }
    bool line(std::string& s) {
        s.clear();
        for (;;) {
            if (!gzgets(f, buf.data(), (int)buf.size())) return !s.empty();
            s += buf.data();
            if (!s.empty() && s.back() == '\n') break;
            if (gzeof(f)) break;
        }
        while (!s.empty() && (s.back() == '\n' || s.back() == '\r')) s.pop_back();
        return true;
    }
    bool next(std::string& h, std::string& seq, std::string& qual) {
        std::string plus;
        do { if (!line(h)) return false; } while (h.empty());
        if (h[0] != '@') fail("Error reading FASTQ file '" + path + "': record does not start with '@'");
        h.erase(0, 1);
        if (!line(seq) || !line(plus) || !line(qual)) fail("Error reading FASTQ file '" + path + "': truncated record");
        return true;
    }
};

struct OutFile {   // '{folder}/{label}.trimmed.fastq[.gz]' (trim.rs:428-446)
    FILE* fp = nullptr;
    gzFile gz = nullptr;
    void write(const uint8_t* p, size_t n) {
        if (gz) { for (size_t o = 0; o < n;) { const unsigned c = (unsigned)std::min<size_t>(n - o, 1u << 30); if (gzwrite(gz, p + o, c) <= 0) fail("gzwrite failed"); o += c; } }
        else if (fwrite(p, 1, n, fp) != n) fail("write failed");
    }
    void close() { if (gz) gzclose(gz); if (fp) fclose(fp); gz = nullptr; fp = nullptr; }
};

}  // namespace

// ProgressTracker::new_with_logging + ProgressLog::write (progress.rs:96-144, 188-195): '{log_dir}/{step}.{unix ms}.log' holding
// "step\tmetric\tcount" and one line per counter — what --verbose leaves behind besides the progress bars (which stay out of scope)
std::string write_progress_log(const std::string& step, const std::string& log_dir, const std::vector<std::pair<std::string, size_t>>& counts) {
    struct timespec ts;
    clock_gettime(CLOCK_REALTIME, &ts);
    const unsigned long long ms = (unsigned long long)ts.tv_sec * 1000ull + (unsigned long long)(ts.tv_nsec / 1000000);
    const std::string path = (log_dir.empty() ? std::string(".") : log_dir) + "/" + step + "." + std::to_string(ms) + ".log";
    FILE* f = fopen(path.c_str(), "wb");
    if (!f) { fprintf(stderr, "Failed to create log file '%s'\n", path.c_str()); return ""; }
    fputs("step\tmetric\tcount\n", f);
    for (const auto& c : counts) fprintf(f, "%s\t%s\t%zu\n", step.c_str(), c.first.c_str(), c.second);
    fclose(f);
    return path;
}
std::string parent_dir(const std::string& file) {   // Path::parent() of an output file, "." where it has none
    const size_t q = file.find_last_of('/');
    return q == std::string::npos ? std::string(".") : (q == 0 ? std::string("/") : file.substr(0, q));
}

void Demuxer::load_rows(const std::vector<bb_row>& rows) {
    ensure_ctx();
    rows_ = rows;
    n_rows_ = rows.size();
}

StepStats filter_file(const std::string& annotated_file, const std::string& output_file, const std::optional<std::string>& dropped_out_file,
                      const std::vector<Pattern>& filters, int device, size_t batch_rows) {
    StepStats st;
    const LabelSpace sp = LabelSpace::from_file(annotated_file);
    FILE* out[2] = {fopen(output_file.c_str(), "wb"), dropped_out_file ? fopen(dropped_out_file->c_str(), "wb") : nullptr};
    if (!out[0] || (dropped_out_file && !out[1])) fail("cannot open the output file");
    bool wrote[2] = {false, false};
    if (!sp.groups.empty()) {
        auto dm = context_for(sp, device, filters);
        BatchReader br(annotated_file, sp, batch_rows);
        std::vector<bb_row> rows;
        std::vector<bb_row_verdict> ver;
        std::vector<std::vector<std::string>> fields;
        std::vector<std::string> ids;
        bool has_cuts;
        while (br.next(rows, ver, fields, ids, has_cuts)) {
            dm->load_rows(rows);
            const std::vector<bb_row_verdict> v = dm->filter_last_batch();
            for (size_t i = 0; i < rows.size(); ++i) {
                const int w = v[i].pass ? 0 : 1;
                if (!i || rows[i].read_idx != rows[i - 1].read_idx) { ++st.total; ++(w ? st.dropped : st.kept); }
                if (!out[w]) continue;
                if (!wrote[w]) { fprintf(out[w], "%s\n", TSV_HEADER); wrote[w] = true; }   // the csv writer emits the header with the first record only
                // the cuts the row came with stay, the new ones follow (filter.rs:204-209: existing_cuts.push) — a filtered.tsv filtered again,
                // e.g. with a sub-selecting pattern without cut markers, keeps what `trim` needs
                if ((size_t)ver[i].n_cuts + v[i].n_cuts > BB_MAX_CUTS)
                    fail(annotated_file + ": read '" + ids[rows[i].read_idx] + "': more than 3 cuts on one row after this filter (kernel limit, include/barbell_amd_filter.h)");
                std::string cuts = ver[i].n_cuts ? fields[i][C_CUTS] : std::string();
                if (v[i].n_cuts) { if (!cuts.empty()) cuts += ','; cuts += format_cuts(v[i]); }
                const std::string l = line_of(fields[i], cuts);
                fwrite(l.data(), 1, l.size(), out[w]);
            }
        }
    }
    for (FILE* f : out) if (f) fclose(f);
    return st;
}

StepStats inspect_file(const std::string& annotated_file, const std::optional<std::string>& read_pattern_out, uint32_t bucket_size, AnnotateStats& patterns,
                       int device, size_t batch_rows) {
    StepStats st;
    const LabelSpace sp = LabelSpace::from_file(annotated_file);
    FILE* ppr = read_pattern_out ? fopen(read_pattern_out->c_str(), "wb") : nullptr;
    if (read_pattern_out && !ppr) fail("cannot open " + *read_pattern_out);
    std::unordered_map<std::string, size_t> counts;
    std::vector<std::string> order;   // first appearance: ties of the count keep it (a stable sort below)
    if (!sp.groups.empty()) {
        auto dm = context_for(sp, device, {});
        BatchReader br(annotated_file, sp, batch_rows);
        std::vector<bb_row> rows;
        std::vector<bb_row_verdict> ver;
        std::vector<std::vector<std::string>> fields;
        std::vector<std::string> ids;
        bool has_cuts;
        while (br.next(rows, ver, fields, ids, has_cuts)) {
            dm->load_rows(rows);
            for (const auto& pr : dm->inspect_last_batch(has_cuts ? &ver : nullptr, bucket_size)) {
                ++st.total;
                if (ppr) fprintf(ppr, "%s\t%s\n", ids[pr.first].c_str(), pr.second.c_str());
                const auto it = counts.find(pr.second);
                if (it == counts.end()) { counts.emplace(pr.second, 1); order.push_back(pr.second); }
                else ++it->second;
            }
        }
    }
    if (ppr) fclose(ppr);
    patterns.patterns.clear();
    for (const auto& p : order) patterns.patterns.emplace_back(p, counts[p]);
    std::stable_sort(patterns.patterns.begin(), patterns.patterns.end(), [](const auto& a, const auto& b) { return a.second > b.second; });
    return st;
}

StepStats trim_file(const std::string& filtered_match_file, const std::vector<std::string>& read_fastq_files, const std::string& output_folder,
                    const TrimConfig& cfg, int device, size_t batch_reads) {
    if (cfg.sort_labels && cfg.only_side) fail("Cannot enable only keeping left/right label and sorting; this is ambiguous");   // trim.rs:331-335
    StepStats st;
    {
        std::error_code ec;
        std::filesystem::create_directories(output_folder, ec);
        if (ec) fail("Failed to create output folder " + output_folder);
    }
    const LabelSpace sp = LabelSpace::from_file(filtered_match_file);
    // annotations by read id, the whole filtered file in memory as the reference holds it (trim.rs:337-358): a read's rows in file order
    std::unordered_map<std::string, uint32_t> by_id;
    std::vector<std::vector<std::pair<bb_row, bb_row_verdict>>> anno;
    {
        Tsv t(filtered_match_file);
        std::vector<std::string> f;
        while (t.next(f)) {
            const auto ins = by_id.emplace(f[C_ID], (uint32_t)anno.size());
            if (ins.second) anno.emplace_back();
            bb_row_verdict v{};
            v.pass = 1;
            parse_cuts(f[C_CUTS], v);
            auto& rows = anno[ins.first->second];
            if (!v.n_cuts) v.match_idx = (uint16_t)rows.size();
            rows.emplace_back(make_row(f, 0, sp, filtered_match_file), v);
        }
    }
    std::unique_ptr<Demuxer> dm;
    if (!sp.groups.empty()) { dm = context_for(sp, device, {}); dm->set_trim(cfg); }
    std::map<std::string, OutFile> writers;
    FILE* failed = cfg.failed_trimmed_writer ? fopen(cfg.failed_trimmed_writer->c_str(), "wb") : nullptr;
    if (cfg.failed_trimmed_writer && !failed) fail("Failed to create " + *cfg.failed_trimmed_writer);   // (the reference unwraps the open: trim.rs:364-370)
    FastqBatch b;
    std::vector<bb_row> rows;
    std::vector<bb_row_verdict> ver;
    auto flush = [&]() {
        if (b.ids.empty()) return;
        dm->load_rows(rows);
        const TrimBatch t = dm->trim_last_batch(ver, b);
        for (const auto& sp_ : t.spans) {
            const std::string label = dm->label_of_key(sp_.label_key);
            auto it = writers.find(label);
            if (it == writers.end()) {
                const std::string path = output_folder + "/" + label + (cfg.gzip ? ".trimmed.fastq.gz" : ".trimmed.fastq");
                OutFile o;
                if (cfg.gzip) o.gz = gzopen(path.c_str(), "wb"); else o.fp = fopen(path.c_str(), "wb");
                if (!o.gz && !o.fp) fail("Failed to create output file " + path);
                it = writers.emplace(label, o).first;
            }
            it->second.write(t.data() + sp_.off, (size_t)sp_.len);
        }
        std::vector<uint32_t> per_read(b.ids.size(), 0);
        for (const auto& s : t.slices) ++per_read[s.read_idx];
        for (size_t i = 0; i < b.ids.size(); ++i) {
            if (t.status[i] == BB_TRIM_TRIMMED) ++st.kept;
            else if (t.status[i] == BB_TRIM_FAILED) { ++st.dropped; if (failed) fprintf(failed, "%s\n", b.ids[i].c_str()); }
            if (per_read[i] > 1) ++st.split;
        }
        b.clear(); rows.clear(); ver.clear();
    };
    for (const auto& path : read_fastq_files) {
        FastqReader fq(path);
        std::string h, seq, qual;
        while (fq.next(h, seq, qual)) {
            ++st.total;
            uint32_t id_len, desc_start;
            split_header(h, id_len, desc_start);
            const auto it = by_id.find(h.substr(0, id_len));
            if (it == by_id.end()) continue;
            if (qual.size() != seq.size()) fail("FASTQ record '" + h.substr(0, id_len) + "': " + std::to_string(seq.size()) + " bases, " + std::to_string(qual.size()) + " qualities");
            const uint32_t ridx = (uint32_t)b.ids.size();
            for (const auto& rv : anno[it->second]) {
                // the annotation must be THIS record's (another FASTQ with the same ids, re-basecalled reads, edited rows): the reference panics on
                // seq[start..end]; unchecked, the kernels would copy a neighbour's bases
                const bb_row& a = rv.first;
                if (a.read_len != seq.size() || a.read_start_flank > seq.size() || a.read_end_flank > seq.size() || a.read_start_flank > a.read_end_flank)
                    fail(filtered_match_file + ": read '" + h.substr(0, id_len) + "': the annotation says read_len " + std::to_string(a.read_len) + ", flank " +
                         std::to_string(a.read_start_flank) + ".." + std::to_string(a.read_end_flank) + ", the FASTQ record has " + std::to_string(seq.size()) +
                         " bases (an annotation file of other reads?)");
                rows.push_back(a); rows.back().read_idx = ridx; ver.push_back(rv.second);
            }
            b.ids.push_back(h.substr(0, id_len));
            b.bases.insert(b.bases.end(), seq.begin(), seq.end());
            b.quals.insert(b.quals.end(), qual.begin(), qual.end());
            b.offsets.push_back(b.bases.size());
            b.hdr.insert(b.hdr.end(), h.begin(), h.end());
            b.hdr_offsets.push_back(b.hdr.size());
            b.id_len.push_back(id_len);
            b.desc_start.push_back(desc_start);
            if (b.ids.size() >= batch_reads) flush();
        }
    }
    flush();
    for (auto& w : writers) w.second.close();
    if (failed) fclose(failed);
    return st;
}

}  // namespace barbell
