// bb_host.cpp — see bb_host.hpp.  Host plumbing only: kit/FASTA loading, FASTQ batching, C-ABI
// calls, annotation.tsv.  No alignment arithmetic lives here.
#include "bb_host.hpp"
#include "../bb_pack.h"

#include <zlib.h>

#include <algorithm>
#include <array>
#include <atomic>
#include <unordered_map>
#include <cctype>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <map>
#include <condition_variable>
#include <deque>
#include <memory>
#include <mutex>
#include <thread>

#include <dlfcn.h>
#include <sched.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <fcntl.h>
#include <unistd.h>
#include <cerrno>
#include <chrono>

#include "kits_data.inc"

namespace barbell {

const char* as_str(BarcodeType t) {
    switch (t) {
        case BarcodeType::Ftag: return "Ftag";
        case BarcodeType::Rtag: return "Rtag";
        case BarcodeType::Fflank: return "Fflank";
        default: return "Rflank";
    }
}

const char* const TSV_HEADER =
    "read_id\tread_len\trel_dist_to_end\tread_start_bar\tread_end_bar\tread_start_flank\tread_end_flank\t"
    "bar_start\tbar_end\tmatch_type\tflank_cost\tbarcode_cost\tlabel\tstrand\tcuts";

// ---- kit presets (kits.rs) ----------------------------------------------------------------------
namespace {
struct Label { std::string prefix; size_t number; bool a_flag; };
Label parse_label_simple(const std::string& label) {  // kits.rs:710-739
    Label l{"", 0, false};
    size_t i = 0;
    while (i < label.size() && std::isalpha((unsigned char)label[i])) l.prefix.push_back((char)std::toupper((unsigned char)label[i++]));
    std::string num;
    while (i < label.size() && std::isdigit((unsigned char)label[i])) num.push_back(label[i++]);
    if (num.empty()) throw BarbellError(BB_E_INVALID, "Invalid numeric part in label: " + label);
    l.number = std::stoul(num);
    l.a_flag = i < label.size() && std::toupper((unsigned char)label[i]) == 'A';
    return l;
}
std::string two(size_t n) { char b[16]; snprintf(b, sizeof b, "%02zu", n); return b; }
}  // namespace

std::vector<std::string> get_barcodes(const std::string& from_label, const std::string& to_label, bool use_12a_flag) {
    const Label f = parse_label_simple(from_label), t = parse_label_simple(to_label);
    if (f.prefix != t.prefix) throw BarbellError(BB_E_INVALID, "Mismatched label prefixes: " + f.prefix + " vs " + t.prefix);
    const size_t start = std::min(f.number, t.number), end = std::max(f.number, t.number);
    std::vector<std::string> out;
    for (size_t i = start; i <= end; ++i) out.push_back((f.prefix == "AB" ? "AB" : "BC") + two(i));
    if (f.prefix == "AB") return out;
    const bool use_12a = use_12a_flag || ((f.a_flag || t.a_flag) && start <= 12 && 12 <= end);
    for (auto& s : out) {
        if (use_12a && s == "BC12") s = "BC12A";
        if (f.prefix == "NB" && s.rfind("BC", 0) == 0) s = "NB" + s.substr(2);
        if (f.prefix == "RBK" && s.rfind("BC", 0) == 0 && s.size() >= 4) {
            const int n = std::atoi(s.substr(2, 2).c_str());
            if (n == 26 || n == 39 || n == 40 || n == 48 || n == 54 || n == 60) s = "RBK" + s.substr(2);
        }
    }
    return out;
}

const char* lookup_barcode_seq(const std::string& label) {  // kits.rs:1074-1103
    const Label l = parse_label_simple(label);
    const size_t idx = l.number ? l.number - 1 : 0;
    auto get = [&](const char* const* tab, int n) -> const char* { return idx < (size_t)n ? tab[idx] : nullptr; };
    if (l.prefix == "BC" || l.prefix == "NB") {
        if (l.a_flag && l.number == 12) return BC12A_SEQ;
        return l.prefix == "BC" ? get(BC_SEQS, BC_SEQS_N) : get(NB_SEQS, NB_SEQS_N);
    }
    if (l.prefix == "AB") return get(AB_SEQS, AB_SEQS_N);
    if (l.prefix == "BP") return get(BP_SEQS, BP_SEQS_N);
    if (l.prefix == "RBK") {
        for (const auto& s : RBK_SPECIAL)
            if ((size_t)s.number == l.number) return s.seq;
        return get(BC_SEQS, BC_SEQS_N);
    }
    return nullptr;
}

std::vector<std::string> supported_kits() {
    std::vector<std::string> v;
    for (const auto& k : KITS) v.push_back(k.kit);
    return v;
}

std::vector<BarcodeGroup> BarcodeGroup::new_from_kit(const std::string& kit_in, bool also_use_extended) {
    std::string kit = kit_in;
    const TemplateSpecData* specs = nullptr;
    int n = 0;
    for (int attempt = 0; attempt < 2 && !specs; ++attempt) {
        for (const auto& k : KITS)
            if (kit == k.kit) { specs = k.specs; n = k.n; }
        if (!specs) std::replace(kit.begin(), kit.end(), '.', '-');  // kits.rs:694-702
    }
    if (!specs) throw BarbellError(BB_E_INVALID, "Unknown or unsupported kit: " + kit_in + ", please raise an issue");
    std::vector<BarcodeGroup> groups;
    for (int i = 0; i < n; ++i) {
        const TemplateSpecData& t = specs[i];
        if (t.extended && !also_use_extended) continue;  // barcodes.rs:258-262
        BarcodeGroup g;
        g.labels = get_barcodes(t.from, t.to, t.use_12a);
        for (const auto& lab : g.labels) {
            const char* bar = lookup_barcode_seq(lab);
            if (!bar) throw BarbellError(BB_E_INVALID, "Barcode not found - odd - raise issue");
            g.seqs.push_back(std::string(t.front) + bar + t.rear);
        }
        g.barcode_type = t.right ? BarcodeType::Rtag : BarcodeType::Ftag;
        groups.push_back(std::move(g));
    }
    return groups;
}

BarcodeGroup BarcodeGroup::new_from_fasta(const std::string& fasta_file, BarcodeType bar_type) {
    std::ifstream f(fasta_file);
    if (!f) throw BarbellError(BB_E_INVALID, "Query file not found: " + fasta_file);
    BarcodeGroup g;
    g.barcode_type = bar_type;
    std::string line, cur;
    auto flush = [&]() { if (!cur.empty()) { g.seqs.push_back(cur); cur.clear(); } };
    while (std::getline(f, line)) {
        while (!line.empty() && (line.back() == '\r' || line.back() == '\n')) line.pop_back();
        if (line.empty()) continue;
        if (line[0] == '>') {
            flush();
            g.labels.push_back(line.substr(1, line.find_first_of(" \t", 1) == std::string::npos ? std::string::npos : line.find_first_of(" \t", 1) - 1));
        } else {
            for (char& c : line) c = (char)std::toupper((unsigned char)c);  // needletail normalize(true)
            cur += line;
        }
    }
    flush();
    return g;
}

// ---- BarbellMatch -------------------------------------------------------------------------------
static std::string csv_field(const std::string& x) {  // csv crate, QuoteStyle::Necessary (annotator.rs:246-251)
    if (x.find_first_of("\t\"\n\r") == std::string::npos) return x;
    std::string q = "\"";
    for (char c : x) { if (c == '"') q += '"'; q += c; }
    return q + "\"";
}
std::string BarbellMatch::to_tsv() const {
    const std::string id = csv_field(read_id);
    char buf[512];
    snprintf(buf, sizeof buf, "\t%zu\t%ld\t%zu\t%zu\t%zu\t%zu\t%zu\t%zu\t%s\t%d\t%d\t", read_len, rel_dist_to_end, read_start_bar,
             read_end_bar, read_start_flank, read_end_flank, bar_start, bar_end, as_str(match_type), flank_cost, barcode_cost);
    return id + buf + csv_field(label) + "\t" + (strand_rc ? "Rc" : "Fwd") + "\t" + cuts;
}

// ---- filter patterns (pattern.rs) -----------------------------------------------------------------
namespace {
std::string trim(const std::string& s) {
    size_t a = 0, b = s.size();
    while (a < b && isspace((unsigned char)s[a])) ++a;
    while (b > a && isspace((unsigned char)s[b - 1])) --b;
    return s.substr(a, b - a);
}
std::vector<std::string> split(const std::string& s, const std::string& sep) {
    std::vector<std::string> out;
    size_t pos = 0;
    for (;;) {
        const size_t q = s.find(sep, pos);
        out.push_back(s.substr(pos, q == std::string::npos ? std::string::npos : q - pos));
        if (q == std::string::npos) return out;
        pos = q + sep.size();
    }
}
bool parse_int(const std::string& s, long& v) {  // str::parse::<isize>: optional sign, digits only
    if (s.empty()) return false;
    size_t i = (s[0] == '-' || s[0] == '+') ? 1 : 0;
    if (i == s.size()) return false;
    for (size_t j = i; j < s.size(); ++j) if (!isdigit((unsigned char)s[j])) return false;
    v = atol(s.c_str());
    return true;
}
bool parse_range(std::string s, long& lo, long& hi) {  // pattern.rs:249-261
    size_t a = 0, b = s.size();
    while (a < b && (s[a] == '(' || s[a] == ')')) ++a;
    while (b > a && (s[b - 1] == '(' || s[b - 1] == ')')) --b;
    const auto parts = split(s.substr(a, b - a), "..");
    return parts.size() == 2 && parse_int(trim(parts[0]), lo) && parse_int(trim(parts[1]), hi);
}
bool parse_position(const std::string& p, int& rel, long& lo, long& hi) {  // pattern.rs:263-279
    const auto parts = split(p, "(");
    if (parts.size() != 2) return false;
    std::string name = parts[0];
    while (!name.empty() && name[0] == '@') name.erase(0, 1);
    if (name == "left") rel = BB_REL_LEFT;
    else if (name == "right") rel = BB_REL_RIGHT;
    else if (name == "prev_left") rel = BB_REL_PREV_LEFT;
    else return false;
    return parse_range(trim(p.substr(parts[0].size())), lo, hi);
}
bool parse_element(const std::string& s, PatternElement& el) {  // pattern.rs:287-356
    const size_t br = s.find('[');
    if (br == std::string::npos) return false;
    const std::string name = trim(s.substr(0, br));
    if (name == "Flank" || name == "flank") throw BarbellError(BB_E_INVALID, "Flank is not valid, use Fflank or Rflank");
    if (name == "Ftag") el.match_type = BarcodeType::Ftag;
    else if (name == "Rtag") el.match_type = BarcodeType::Rtag;
    else if (name == "Fflank") el.match_type = BarcodeType::Fflank;
    else if (name == "Rflank") el.match_type = BarcodeType::Rflank;
    else return false;
    std::string body = s.substr(br + 1);
    while (!body.empty() && body.back() == ']') body.pop_back();
    for (const auto& raw : split(body, ",")) {
        const std::string param = trim(raw);
        if (param == "fw") el.orientation = BB_FWD;
        else if (param == "rc") el.orientation = BB_RC;
        else if (!param.empty() && param[0] == '@') {
            int rel; long lo, hi;
            if (parse_position(param, rel, lo, hi)) { el.relative_to = rel; el.range_lo = lo; el.range_hi = hi; }
        } else if (!param.empty() && param[0] == '?') {
            long v;
            if (parse_int(param.substr(1), v) && v >= 0 && param[1] != '-') el.placeholder = (int)v;
        } else if (!param.empty() && (param[0] == '>' || param[0] == '<')) {
            if (auto c = Cut::from_pattern_string(param)) el.cuts.push_back(*c);
        } else if (param == "*") {
        } else {
            size_t a = 0, b = param.size();
            while (a < b && param[a] == '"') ++a;
            while (b > a && param[b - 1] == '"') --b;
            el.label = param.substr(a, b - a);
        }
    }
    return true;
}
}  // namespace

std::optional<Cut> Cut::from_pattern_string(const std::string& s) {
    if (s.size() < 2) throw BarbellError(BB_E_INVALID, "cut marker too short: '" + s + "'");  // the reference slices [..2] and panics
    if (s.compare(0, 2, ">>") != 0 && s.compare(0, 2, "<<") != 0) return std::nullopt;
    long gid = 0;
    if (s.size() > 2 && (!parse_int(s.substr(2), gid) || gid < 0 || s[2] == '-')) return std::nullopt;
    Cut c;
    c.group_id = (size_t)gid;
    c.after = s[0] == '>';
    return c;
}
std::string Cut::to_string() const { return std::string(after ? "After(" : "Before(") + std::to_string(group_id) + ")"; }

Pattern pattern_from_str(const std::string& s) {
    Pattern p;
    const auto parts = split(s, "__");
    for (const auto& part : parts) {
        PatternElement el;
        if (parse_element(trim(part), el)) p.elements.push_back(std::move(el));
    }
    if (p.elements.size() != parts.size())  // basic_verify, pattern.rs:281-285
        throw BarbellError(BB_E_INVALID, "Pattern parse error: could not convert all elements of '" + s + "'");
    return p;
}

std::vector<Pattern> patterns_from_files(const std::vector<std::string>& paths) {
    if (paths.empty()) throw BarbellError(BB_E_INVALID, "No filter pattern files provided");
    std::vector<Pattern> out;
    for (const auto& path : paths) {
        std::ifstream f(path);
        if (!f) throw BarbellError(BB_E_INVALID, "Failed to open pattern file: " + path);
        std::string line;
        while (std::getline(f, line)) {
            line = trim(line);
            if (!line.empty()) out.push_back(pattern_from_str(line));
        }
    }
    if (out.empty()) throw BarbellError(BB_E_INVALID, "No filter patterns found");
    return out;
}

std::vector<Pattern> kit_patterns(const std::string& kit_in, bool maximize) {
    std::string kit = kit_in;
    for (int pass = 0; pass < 2; ++pass) {
        for (const auto& k : KIT_FILTER)
            if (kit == k.kit) {
                std::vector<Pattern> out;
                const char* const* set = maximize ? k.maximize : k.safe;
                const int n = maximize ? k.n_maximize : k.n_safe;
                for (int i = 0; i < n; ++i) out.push_back(pattern_from_str(set[i]));
                return out;
            }
        std::replace(kit.begin(), kit.end(), '.', '-');
    }
    throw BarbellError(BB_E_INVALID, "Unsupported kit: " + kit_in);
}

// ---- Demuxer ------------------------------------------------------------------------------------
Demuxer::Demuxer(float alpha, bool verbose, double min_score_frac, double min_score_diff_frac, int device)
    : alpha_(alpha), verbose_(verbose), min_score_(min_score_frac), min_score_diff_(min_score_diff_frac), device_(device) {}
Demuxer::~Demuxer() {
    for (DevBuf* b : {&d_rows_, &d_ver_, &d_elems_, &d_text_, &d_slices_, &d_spans_, &d_status_, &d_tsv_}) b->release();
    {   // every holder of a landing buffer (the file writers) must be gone by now
        std::lock_guard<std::mutex> lk(text_pool_->mu);
        for (uint8_t* h : text_pool_->all) bb_host_free(ctx_, h);
        text_pool_->all.clear(); text_pool_->free_.clear();
    }
    if (ctx_) bb_destroy(ctx_);
}

Demuxer& Demuxer::add_query_group(BarcodeGroup g) {
    if (ctx_) throw BarbellError(BB_E_INVALID, "add_query_group after the first demux call");
    queries_.push_back(std::move(g));
    return *this;
}

void Demuxer::ensure_ctx() {
    if (ctx_) return;
    std::vector<bb_group_desc> descs(queries_.size());
    std::vector<std::vector<const uint8_t*>> ptrs(queries_.size());
    std::vector<std::vector<uint32_t>> lens(queries_.size());
    for (size_t i = 0; i < queries_.size(); ++i) {
        for (const auto& s : queries_[i].seqs) { ptrs[i].push_back((const uint8_t*)s.data()); lens[i].push_back((uint32_t)s.size()); }
        descs[i].seqs = ptrs[i].data();
        descs[i].seq_lens = lens[i].data();
        descs[i].n_seqs = (uint32_t)queries_[i].seqs.size();
        descs[i].type = queries_[i].barcode_type == BarcodeType::Rtag ? BB_RTAG : BB_FTAG;
        descs[i].flank_k = queries_[i].k_cutoff ? (int32_t)*queries_[i].k_cutoff : -1;
    }
    bb_params p{alpha_, min_score_, min_score_diff_, device_};
    const int rc = bb_create(descs.data(), (uint32_t)descs.size(), &p, &ctx_);
    if (rc != BB_OK) { ctx_ = nullptr; throw BarbellError(rc, std::string("bb_create: ") + bb_strerror(rc)); }
    // label strings for the TSV renderer (they do not cross bb_create)
    std::string blob;
    std::vector<uint32_t> off{0};
    for (const auto& l : slot_labels()) { blob += l; off.push_back((uint32_t)blob.size()); }
    blob.push_back('\0');
    const int r2 = bb_format_set_labels(ctx_, (const uint8_t*)blob.data(), off.data());
    if (r2 != BB_OK) throw BarbellError(r2, std::string("bb_format_set_labels: ") + bb_strerror(r2));
}

std::vector<std::string> Demuxer::slot_labels() const {
    std::vector<std::string> out;
    for (const auto& g : queries_) { out.insert(out.end(), g.labels.begin(), g.labels.end()); out.push_back("flank"); }
    return out;
}
std::vector<uint64_t> Demuxer::counts() {
    ensure_ctx();
    std::vector<uint64_t> c(bb_counts_len(ctx_));
    const int rc = bb_counts(ctx_, c.data());
    if (rc != BB_OK) throw BarbellError(rc, bb_strerror(rc));
    return c;
}

bb_group_info Demuxer::group_info(size_t g) {
    ensure_ctx();
    bb_group_info i;
    const int rc = bb_group_get_info(ctx_, (uint32_t)g, &i);
    if (rc != BB_OK) throw BarbellError(rc, bb_strerror(rc));
    return i;
}

std::vector<BarbellMatch> Demuxer::demux_batch(const std::vector<std::string>& read_ids, const std::vector<uint8_t>& bases,
                                               const std::vector<uint64_t>& offsets) {
    ensure_ctx();
    const uint32_t n = (uint32_t)read_ids.size();
    if (rows_.size() < (size_t)4 * n + 64) rows_.resize((size_t)4 * n + 64);
    uint64_t n_rows = 0;
    int rc = bb_annotate_batch(ctx_, bases.data(), offsets.data(), n, rows_.data(), rows_.size(), &n_rows);
    if (rc == BB_E_CAPACITY) {
        rows_.resize(n_rows);
        rc = bb_annotate_batch(ctx_, bases.data(), offsets.data(), n, rows_.data(), rows_.size(), &n_rows);
    }
    if (rc != BB_OK) throw BarbellError(rc, std::string("bb_annotate_batch: ") + bb_strerror(rc) + " " + bb_last_error(ctx_));
    n_rows_ = n_rows;
    return rows_to_matches(read_ids);
}

std::vector<BarbellMatch> Demuxer::rows_to_matches(const std::vector<std::string>& read_ids) const {
    std::vector<BarbellMatch> out;
    out.reserve(n_rows_);
    for (uint64_t i = 0; i < n_rows_; ++i) {
        const bb_row& r = rows_[i];
        const BarcodeGroup& g = queries_[r.group_idx];
        BarbellMatch m;
        m.read_id = read_ids[r.read_idx];
        m.read_len = r.read_len; m.rel_dist_to_end = r.rel_dist_to_end;
        m.read_start_bar = r.read_start_bar; m.read_end_bar = r.read_end_bar;
        m.read_start_flank = r.read_start_flank; m.read_end_flank = r.read_end_flank;
        m.bar_start = r.bar_start; m.bar_end = r.bar_end;
        m.match_type = (BarcodeType)r.match_type;
        m.flank_cost = r.flank_cost; m.barcode_cost = r.barcode_cost;
        m.label = r.barcode_idx < 0 ? "flank" : g.labels[(size_t)r.barcode_idx];
        m.strand_rc = r.strand == BB_RC;
        out.push_back(std::move(m));
    }
    return out;
}

void Demuxer::set_filter(const std::vector<Pattern>& patterns) {
    ensure_ctx();
    // histogram slot space: per group its labels then "flank" (bb_counts layout); equal strings share an id
    std::vector<std::string> slot;
    for (const auto& g : queries_) { slot.insert(slot.end(), g.labels.begin(), g.labels.end()); slot.push_back("flank"); }
    std::vector<uint32_t> ids(slot.size());
    label_strings_.clear();
    for (size_t i = 0; i < slot.size(); ++i) {  // dense ids in first-appearance order; equal strings share an id
        const auto it = std::find(label_strings_.begin(), label_strings_.end(), slot[i]);
        ids[i] = (uint32_t)(it - label_strings_.begin());
        if (it == label_strings_.end()) label_strings_.push_back(slot[i]);
    }
    std::vector<std::vector<bb_pattern_elem>> elems(patterns.size());
    std::vector<std::vector<uint8_t>> oks;
    size_t n_lab = 0;
    for (const auto& p : patterns) for (const auto& e : p.elements) n_lab += e.label ? 1 : 0;
    oks.reserve(n_lab);
    std::vector<bb_pattern> pats(patterns.size());
    for (size_t i = 0; i < patterns.size(); ++i) {
        for (const auto& e : patterns[i].elements) {
            if (e.cuts.size() > BB_MAX_CUTS) throw BarbellError(BB_E_INVALID, "more than 3 cut markers on one pattern element");
            bb_pattern_elem c{};
            c.match_type = (uint8_t)e.match_type; c.orientation = (int8_t)e.orientation; c.relative_to = (uint8_t)e.relative_to;
            c.n_cuts = (uint8_t)e.cuts.size(); c.placeholder = e.placeholder; c.range_lo = e.range_lo; c.range_hi = e.range_hi;
            if (e.label) {  // exact label or "~substring" (pattern.rs:108-121)
                std::vector<uint8_t> ok(slot.size());
                const bool sub = !e.label->empty() && (*e.label)[0] == '~';
                for (size_t q = 0; q < slot.size(); ++q)
                    ok[q] = sub ? slot[q].find(e.label->substr(1)) != std::string::npos : slot[q] == *e.label;
                oks.push_back(std::move(ok));
                c.label_ok = oks.back().data();
            }
            if (e.cuts.size() > BB_MAX_CUTS) throw BarbellError(BB_E_UNSUPPORTED, "more than 3 cut markers on one pattern element (kernel limit, include/barbell_amd_filter.h)");
            for (size_t q = 0; q < e.cuts.size(); ++q) {
                if (e.cuts[q].group_id > 0xFFFF) throw BarbellError(BB_E_UNSUPPORTED, "cut group id above 65535 (kernel limit, include/barbell_amd_filter.h)");
                c.cuts[q] = bb_cut{(uint8_t)(e.cuts[q].after ? BB_CUT_AFTER : BB_CUT_BEFORE), 0, (uint16_t)e.cuts[q].group_id};
            }
            elems[i].push_back(c);
        }
        pats[i] = bb_pattern{elems[i].data(), (uint32_t)elems[i].size()};
    }
    const int rc = bb_filter_set(ctx_, pats.data(), (uint32_t)pats.size(), ids.data());
    if (rc != BB_OK) throw BarbellError(rc, std::string("bb_filter_set: ") + bb_strerror(rc) + " " + bb_last_error(ctx_));
    has_filter_ = true;
}

std::vector<bb_row_verdict> Demuxer::filter_last_batch() {
    if (!has_filter_) throw BarbellError(BB_E_INVALID, "filter_last_batch without set_filter");
    std::vector<bb_row_verdict> v(n_rows_);
    if (n_rows_) {
        const int rc = bb_filter_rows(ctx_, rows_.data(), n_rows_, v.data());
        if (rc != BB_OK) throw BarbellError(rc, std::string("bb_filter_rows: ") + bb_strerror(rc) + " " + bb_last_error(ctx_));
    }
    return v;
}

// ---- trim (trim.rs) -------------------------------------------------------------------------------
TrimConfig TrimConfig::for_kit(std::optional<std::string> failed_out, bool gzip) {
    TrimConfig c;
    c.add_labels = true; c.add_orientation = false; c.add_flank = false; c.sort_labels = false; c.only_side = LabelSide::Left;
    c.failed_trimmed_writer = std::move(failed_out); c.write_full_header = true; c.skip_trim = false; c.flip = false; c.gzip = gzip;
    return c;
}

void FastqBatch::clear() {
    ids.clear(); bases.clear(); quals.clear(); hdr.clear(); id_len.clear(); desc_start.clear();
    offsets.assign(1, 0); hdr_offsets.assign(1, 0);
}

std::string Demuxer::part_str(uint32_t part) const {  // trim.rs:70-80
    std::string r = label_strings_.at(part >> 1);
    if (trim_cfg_.add_orientation) r += (part & 1) ? "_rc" : "_fw";
    return r;
}

std::string Demuxer::label_of_key(uint32_t key) const {  // trim.rs:88-104 on a bb_slice.label_key
    if (key == 0) return "none";
    std::string r = part_str((key >> 16) - 1);
    if (key & 0xFFFF) r += "__" + part_str((key & 0xFFFF) - 1);
    return r;
}

void Demuxer::set_trim(const TrimConfig& cfg) {
    if (!has_filter_) throw BarbellError(BB_E_INVALID, "set_trim needs set_filter first (cuts and label ids come from the filter)");
    if (cfg.sort_labels && cfg.only_side)  // trim.rs:330-334
        throw BarbellError(BB_E_INVALID, "Cannot enable only keeping left/right label and sorting; this is ambiguous");
    trim_cfg_ = cfg;
    const size_t n = label_strings_.size();
    std::vector<uint8_t> is_flank(n);
    for (size_t i = 0; i < n; ++i) is_flank[i] = label_strings_[i].find("flank") != std::string::npos;  // trim.rs:66
    std::vector<std::string> parts(2 * n);
    for (size_t i = 0; i < 2 * n; ++i) parts[i] = part_str((uint32_t)i);
    std::vector<std::string> sorted(parts);
    std::sort(sorted.begin(), sorted.end());
    sorted.erase(std::unique(sorted.begin(), sorted.end()), sorted.end());
    std::vector<uint32_t> rank(2 * n);
    for (size_t i = 0; i < 2 * n; ++i) rank[i] = (uint32_t)(std::lower_bound(sorted.begin(), sorted.end(), parts[i]) - sorted.begin());
    const bb_trim_config c = trim_config_pod();
    const int rc = bb_trim_set(ctx_, &c, is_flank.data(), rank.data(), (uint32_t)n);
    if (rc != BB_OK) throw BarbellError(rc, std::string("bb_trim_set: ") + bb_strerror(rc) + " " + bb_last_error(ctx_));
    has_trim_ = true;
}

TrimBatch Demuxer::trim_last_batch(const std::vector<bb_row_verdict>& verdicts, const FastqBatch& b) {
    if (!has_trim_) throw BarbellError(BB_E_INVALID, "trim_last_batch without set_trim");
    if (verdicts.size() != n_rows_) throw BarbellError(BB_E_INVALID, "verdicts do not belong to the last batch");
    const uint32_t n = (uint32_t)b.ids.size();
    TrimBatch t;
    t.status.assign(n, 0);
    t.text.resize(2 * b.bases.size() + b.hdr.size() + 16 * (size_t)n + 1024);
    t.slices.resize(2 * (size_t)n + 64);
    t.spans.resize(4096);
    const bb_headers h{b.hdr.data(), b.hdr_offsets.data(), b.id_len.data(), b.desc_start.data()};
    for (;;) {
        uint64_t tl = 0, ns = 0;
        uint32_t nsp = 0;
        const int rc = bb_trim_batch(ctx_, rows_.data(), verdicts.data(), n_rows_, b.bases.data(), b.quals.data(), b.offsets.data(), &h, n,
                                     t.text.data(), t.text.size(), &tl, t.slices.data(), t.slices.size(), &ns, t.spans.data(),
                                     (uint32_t)t.spans.size(), &nsp, t.status.data());
        if (rc == BB_E_CAPACITY) {
            if (tl > t.text.size()) t.text.resize(tl);
            if (ns > t.slices.size()) t.slices.resize(ns);
            if (nsp > t.spans.size()) t.spans.resize(nsp);
            continue;
        }
        if (rc != BB_OK) throw BarbellError(rc, std::string("bb_trim_batch: ") + bb_strerror(rc) + " " + bb_last_error(ctx_));
        t.text.resize(tl); t.slices.resize(ns); t.spans.resize(nsp);
        t.text_len = tl;
        return t;
    }
}

// ---- inspect (inspect.rs) -------------------------------------------------------------------------
std::vector<std::pair<uint32_t, std::string>> Demuxer::inspect_last_batch(const std::vector<bb_row_verdict>* verdicts, uint32_t bucket_size) {
    ensure_ctx();
    std::vector<bb_inspect_elem> el(n_rows_);
    if (n_rows_) {
        const int rc = bb_inspect_rows(ctx_, rows_.data(), verdicts ? verdicts->data() : nullptr, n_rows_, bucket_size, el.data());
        if (rc != BB_OK) throw BarbellError(rc, std::string("bb_inspect_rows: ") + bb_strerror(rc) + " " + bb_last_error(ctx_));
    }
    return elems_to_patterns(el);
}

std::vector<std::pair<uint32_t, std::string>> Demuxer::elems_to_patterns(const std::vector<bb_inspect_elem>& el) const {
    static const char* const TAG[] = {"", "@left", "@right", "@prev_left"};
    std::vector<std::pair<uint32_t, std::string>> out;
    for (uint64_t i = 0; i < n_rows_; ++i) {  // "{type}[{fw|rc}, *{cut}, {tag}({lo}..{hi})]" joined by "__" (inspect.rs:90-104)
        const bb_inspect_elem& e = el[i];
        char buf[128];
        snprintf(buf, sizeof buf, "%s[%s, *%s, %s(%u..%u)]", as_str((BarcodeType)e.match_type), e.strand ? "rc" : "fw",
                 e.has_cut ? (e.strand ? ", >>" : ", <<") : "", TAG[e.tag & 3], e.lo, e.hi);
        if (e.first) out.emplace_back(rows_[i].read_idx, buf);
        else { out.back().second += "__"; out.back().second += buf; }
    }
    return out;
}

std::vector<std::string> inspect_summary(const AnnotateStats& st, size_t top_n) {
    std::vector<std::string> lines{"Found " + std::to_string(st.patterns.size()) + " unique patterns"};
    for (size_t i = 0; i < st.patterns.size() && i < top_n; ++i) {
        lines.push_back("\tPattern " + std::to_string(i + 1) + ": " + std::to_string(st.patterns[i].second) + " occurrences");
        lines.push_back("\t\t" + st.patterns[i].first);
    }
    lines.push_back("Showed " + std::to_string(top_n) + " / " + std::to_string(st.patterns.size()) + " patterns");
    return lines;
}

// ---- device-resident path -------------------------------------------------------------------------
void DevBuf::ensure(bb_ctx* c, uint64_t bytes) {
    if (p && bytes <= cap) return;
    release();
    ctx = c;
    const uint64_t want = bytes + bytes / 4 + 256;
    const int rc = bb_dev_malloc(c, want, &p);
    if (rc != BB_OK) throw BarbellError(rc, std::string("bb_dev_malloc: ") + bb_strerror(rc));
    cap = want;
}
void DevBuf::release() {
    if (p) bb_dev_free(ctx, p);
    p = nullptr; cap = 0;
}

#define BB_THROW(rc, what) throw BarbellError((rc), std::string(what ": ") + bb_strerror(rc) + " " + bb_last_error(ctx_))

Demuxer::Ingested Demuxer::ingest(const uint8_t* text, uint64_t len, bool final_block, bool want_ids, bool two_line, bool packed) {
    ensure_ctx();
    ing_ = Ingested{};
    n_rows_ = 0;
    int rc = bb_fastq_ingest(ctx_, text, len, (final_block ? BB_FASTQ_FINAL : 0) | (two_line ? BB_FASTQ_TWO_LINE : 0) | (two_line && packed ? BB_FASTQ_PACKED : 0),
                             &ing_.info, &batch_);
    if (rc != BB_OK) BB_THROW(rc, "bb_fastq_ingest");
    if (!want_ids) return ing_;  // the TSV renderer reads the ids where they are
    const uint64_t n = ing_.info.n_records;
    std::vector<uint8_t> hdr(ing_.info.n_hdr);
    std::vector<uint64_t> hoff(n + 1);
    std::vector<uint32_t> idl(n);
    rc = bb_fastq_fetch(ctx_, nullptr, hdr.data(), hoff.data(), idl.data(), nullptr, nullptr, nullptr);
    if (rc != BB_OK) BB_THROW(rc, "bb_fastq_fetch");
    ing_.ids.reserve(n);
    for (uint64_t i = 0; i < n; ++i) ing_.ids.emplace_back((const char*)hdr.data() + hoff[i], idl[i]);
    return ing_;
}

uint64_t Demuxer::annotate_ingested() {
    const uint32_t n = (uint32_t)ing_.info.n_records;
    uint64_t cap = 4ull * n + 64, n_rows = 0;
    d_rows_.ensure(ctx_, cap * sizeof(bb_row));
    int rc = bb_annotate_batch_dev(ctx_, batch_.d_bases, batch_.d_offsets, n, (bb_row*)d_rows_.p, cap, &n_rows);
    if (rc == BB_E_CAPACITY) {
        cap = n_rows;
        d_rows_.ensure(ctx_, cap * sizeof(bb_row));
        rc = bb_annotate_batch_dev(ctx_, batch_.d_bases, batch_.d_offsets, n, (bb_row*)d_rows_.p, cap, &n_rows);
    }
    if (rc != BB_OK) BB_THROW(rc, "bb_annotate_batch_dev");
    n_rows_ = n_rows;
    if (rows_.size() < n_rows) rows_.resize(n_rows);
    if ((rc = bb_dev_download(ctx_, rows_.data(), d_rows_.p, n_rows * sizeof(bb_row))) != BB_OK) BB_THROW(rc, "bb_dev_download");
    return n_rows;
}

uint64_t Demuxer::format_ingested(int mode, std::vector<uint8_t>& out) {
    if (n_rows_ == 0) return 0;
    uint64_t cap = std::max<uint64_t>(1u << 16, 160 * n_rows_), tl = 0, nl = 0;
    for (;;) {
        d_tsv_.ensure(ctx_, cap);
        const int rc = bb_format_rows_dev(ctx_, (const bb_row*)d_rows_.p, mode == BB_FMT_ALL ? nullptr : (const bb_row_verdict*)d_ver_.p, n_rows_, mode,
                                          &batch_.d_headers, (uint8_t*)d_tsv_.p, cap, &tl, &nl);
        if (rc == BB_E_CAPACITY) { cap = tl; continue; }
        if (rc != BB_OK) BB_THROW(rc, "bb_format_rows_dev");
        break;
    }
    const size_t at = out.size();
    out.resize(at + tl);
    int rc;
    if (tl && (rc = bb_dev_download(ctx_, out.data() + at, d_tsv_.p, tl)) != BB_OK) BB_THROW(rc, "bb_dev_download");
    return nl;
}

std::vector<BarbellMatch> Demuxer::demux_ingested() {
    const uint32_t n = (uint32_t)ing_.info.n_records;
    uint64_t cap = 4ull * n + 64, n_rows = 0;
    d_rows_.ensure(ctx_, cap * sizeof(bb_row));
    int rc = bb_annotate_batch_dev(ctx_, batch_.d_bases, batch_.d_offsets, n, (bb_row*)d_rows_.p, cap, &n_rows);
    if (rc == BB_E_CAPACITY) {
        cap = n_rows;
        d_rows_.ensure(ctx_, cap * sizeof(bb_row));
        rc = bb_annotate_batch_dev(ctx_, batch_.d_bases, batch_.d_offsets, n, (bb_row*)d_rows_.p, cap, &n_rows);
    }
    if (rc != BB_OK) BB_THROW(rc, "bb_annotate_batch_dev");
    n_rows_ = n_rows;
    if (rows_.size() < n_rows) rows_.resize(n_rows);
    if ((rc = bb_dev_download(ctx_, rows_.data(), d_rows_.p, n_rows * sizeof(bb_row))) != BB_OK) BB_THROW(rc, "bb_dev_download");
    return rows_to_matches(ing_.ids);
}

std::vector<bb_row_verdict> Demuxer::filter_ingested() {
    if (!has_filter_) throw BarbellError(BB_E_INVALID, "filter_ingested without set_filter");
    std::vector<bb_row_verdict> v(n_rows_);
    d_ver_.ensure(ctx_, (n_rows_ + 1) * sizeof(bb_row_verdict));
    if (n_rows_) {
        int rc = bb_filter_rows_dev(ctx_, (const bb_row*)d_rows_.p, n_rows_, (bb_row_verdict*)d_ver_.p);
        if (rc != BB_OK) BB_THROW(rc, "bb_filter_rows_dev");
        if ((rc = bb_dev_download(ctx_, v.data(), d_ver_.p, n_rows_ * sizeof(bb_row_verdict))) != BB_OK) BB_THROW(rc, "bb_dev_download");
    }
    return v;
}

std::vector<std::pair<uint32_t, std::string>> Demuxer::inspect_ingested(bool with_verdicts, uint32_t bucket_size) {
    std::vector<bb_inspect_elem> el(n_rows_);
    if (n_rows_) {
        d_elems_.ensure(ctx_, n_rows_ * sizeof(bb_inspect_elem));
        int rc = bb_inspect_rows_dev(ctx_, (const bb_row*)d_rows_.p, with_verdicts ? (const bb_row_verdict*)d_ver_.p : nullptr, n_rows_, bucket_size,
                                     (bb_inspect_elem*)d_elems_.p);
        if (rc != BB_OK) BB_THROW(rc, "bb_inspect_rows_dev");
        if ((rc = bb_dev_download(ctx_, el.data(), d_elems_.p, n_rows_ * sizeof(bb_inspect_elem))) != BB_OK) BB_THROW(rc, "bb_dev_download");
    }
    return elems_to_patterns(el);
}

// The same, interned: a few hundred distinct patterns cover millions of reads, so a read's elements (their bytes are the key) are
// looked up first and the pattern text is only formatted the first time it is seen in the batch.
void Demuxer::inspect_ingested_interned(bool with_verdicts, uint32_t bucket_size, std::vector<std::string>& patterns,
                                        std::vector<std::pair<uint32_t, uint32_t>>& per_read) {
    static const char* const TAG[] = {"", "@left", "@right", "@prev_left"};
    patterns.clear(); per_read.clear();
    if (!n_rows_) return;
    std::vector<bb_inspect_elem> el(n_rows_);
    d_elems_.ensure(ctx_, n_rows_ * sizeof(bb_inspect_elem));
    int rc = bb_inspect_rows_dev(ctx_, (const bb_row*)d_rows_.p, with_verdicts ? (const bb_row_verdict*)d_ver_.p : nullptr, n_rows_, bucket_size,
                                 (bb_inspect_elem*)d_elems_.p);
    if (rc != BB_OK) BB_THROW(rc, "bb_inspect_rows_dev");
    if ((rc = bb_dev_download(ctx_, el.data(), d_elems_.p, n_rows_ * sizeof(bb_inspect_elem))) != BB_OK) BB_THROW(rc, "bb_dev_download");
    std::unordered_map<std::string, uint32_t> seen;
    std::string key;
    for (uint64_t i = 0; i < n_rows_;) {
        uint64_t j = i + 1;
        while (j < n_rows_ && !el[j].first) ++j;
        key.assign((const char*)&el[i], (size_t)(j - i) * sizeof(bb_inspect_elem));  // `first` is 1, 0, 0, .. in every key: harmless
        auto it = seen.find(key);
        if (it == seen.end()) {
            std::string text;
            for (uint64_t k = i; k < j; ++k) {
                const bb_inspect_elem& e = el[k];
                char buf[128];
                snprintf(buf, sizeof buf, "%s[%s, *%s, %s(%u..%u)]", as_str((BarcodeType)e.match_type), e.strand ? "rc" : "fw",
                         e.has_cut ? (e.strand ? ", >>" : ", <<") : "", TAG[e.tag & 3], e.lo, e.hi);
                if (k > i) text += "__";
                text += buf;
            }
            it = seen.emplace(key, (uint32_t)patterns.size()).first;
            patterns.push_back(std::move(text));
        }
        per_read.emplace_back(rows_[i].read_idx, it->second);
        i = j;
    }
}

TrimBatch Demuxer::trim_ingested() {
    if (!has_trim_) throw BarbellError(BB_E_INVALID, "trim_ingested without set_trim");
    const uint32_t n = (uint32_t)ing_.info.n_records;
    TrimBatch t;
    t.status.assign(n, 0);
    uint64_t text_cap = 2 * ing_.info.n_bases + 2 * ing_.info.n_hdr + 32ull * n + 1024, slices_cap = 2ull * n + 64;
    uint32_t spans_cap = 4096;
    d_status_.ensure(ctx_, (uint64_t)n + 16);
    for (;;) {
        d_text_.ensure(ctx_, text_cap);
        d_slices_.ensure(ctx_, slices_cap * sizeof(bb_slice));
        d_spans_.ensure(ctx_, (uint64_t)spans_cap * sizeof(bb_label_span));
        uint64_t tl = 0, ns = 0;
        uint32_t nsp = 0;
        const int rc = bb_trim_batch_dev(ctx_, (const bb_row*)d_rows_.p, (const bb_row_verdict*)d_ver_.p, n_rows_, batch_.d_bases, batch_.d_quals,
                                         batch_.d_offsets, &batch_.d_headers, n, (uint8_t*)d_text_.p, text_cap, &tl, (bb_slice*)d_slices_.p, slices_cap,
                                         &ns, (bb_label_span*)d_spans_.p, spans_cap, &nsp, (uint8_t*)d_status_.p);
        if (rc == BB_E_CAPACITY) {
            text_cap = std::max(text_cap, tl); slices_cap = std::max(slices_cap, ns); spans_cap = std::max(spans_cap, nsp);
            continue;
        }
        if (rc != BB_OK) BB_THROW(rc, "bb_trim_batch_dev");
        t.slices.resize(ns); t.spans.resize(nsp);
        int r2;
        uint8_t* hp = nullptr;
        uint64_t hcap = 0;
        {
            std::lock_guard<std::mutex> lk(text_pool_->mu);
            for (size_t i = 0; i < text_pool_->free_.size(); ++i)
                if (text_pool_->free_[i].second >= tl) { hp = text_pool_->free_[i].first; hcap = text_pool_->free_[i].second; text_pool_->free_.erase(text_pool_->free_.begin() + (long)i); break; }
        }
        if (!hp) {  // page-locked and not zero-filled: a 3 GB std::vector costs more than the copy itself
            void* q = nullptr;
            hcap = tl + tl / 4 + 4096;
            if ((r2 = bb_host_malloc(ctx_, hcap, &q)) != BB_OK) BB_THROW(r2, "bb_host_malloc");
            hp = (uint8_t*)q;
            std::lock_guard<std::mutex> lk(text_pool_->mu);
            text_pool_->all.push_back(hp);
        }
        {
            std::shared_ptr<TextPool> pool = text_pool_;
            t.text_hold = std::shared_ptr<void>((void*)hp, [pool, hp, hcap](void*) { std::lock_guard<std::mutex> lk(pool->mu); pool->free_.emplace_back(hp, hcap); });
        }
        t.text_ptr = hp; t.text_len = tl;
        if ((r2 = bb_dev_download(ctx_, hp, d_text_.p, tl)) != BB_OK) BB_THROW(r2, "bb_dev_download");
        if ((r2 = bb_dev_download(ctx_, t.slices.data(), d_slices_.p, ns * sizeof(bb_slice))) != BB_OK) BB_THROW(r2, "bb_dev_download");
        if ((r2 = bb_dev_download(ctx_, t.spans.data(), d_spans_.p, (uint64_t)nsp * sizeof(bb_label_span))) != BB_OK) BB_THROW(r2, "bb_dev_download");
        if ((r2 = bb_dev_download(ctx_, t.status.data(), d_status_.p, n)) != BB_OK) BB_THROW(r2, "bb_dev_download");
        return t;
    }
}

bb_trim_config Demuxer::trim_config_pod() const {
    const TrimConfig& cfg = trim_cfg_;
    bb_trim_config c{};
    c.add_labels = cfg.add_labels; c.add_orientation = cfg.add_orientation; c.add_flank = cfg.add_flank; c.sort_labels = cfg.sort_labels;
    c.only_side = !cfg.only_side ? BB_SIDE_NONE : (*cfg.only_side == LabelSide::Left ? BB_SIDE_LEFT : BB_SIDE_RIGHT);
    c.write_full_header = cfg.write_full_header; c.skip_trim = cfg.skip_trim; c.flip = cfg.flip;
    return c;
}

TrimPlan Demuxer::trim_plan_ingested() {
    if (!has_trim_) throw BarbellError(BB_E_INVALID, "trim_plan_ingested without set_trim");
    const uint32_t n = (uint32_t)ing_.info.n_records;
    TrimPlan t;
    t.status.assign(n, 0);
    uint64_t slices_cap = 2ull * n + 64;
    uint32_t spans_cap = 4096;
    d_status_.ensure(ctx_, (uint64_t)n + 16);
    for (;;) {
        d_slices_.ensure(ctx_, slices_cap * sizeof(bb_slice));
        d_spans_.ensure(ctx_, (uint64_t)spans_cap * sizeof(bb_label_span));
        uint64_t tl = 0, ns = 0;
        uint32_t nsp = 0;
        const int rc = bb_trim_plan_dev(ctx_, (const bb_row*)d_rows_.p, (const bb_row_verdict*)d_ver_.p, n_rows_, batch_.d_offsets, &batch_.d_headers, n,
                                        &tl, (bb_slice*)d_slices_.p, slices_cap, &ns, (bb_label_span*)d_spans_.p, spans_cap, &nsp, (uint8_t*)d_status_.p);
        if (rc == BB_E_CAPACITY) { slices_cap = std::max(slices_cap, ns); spans_cap = std::max(spans_cap, nsp); continue; }
        if (rc != BB_OK) BB_THROW(rc, "bb_trim_plan_dev");
        t.text_len = tl;
        t.slices.resize(ns); t.spans.resize(nsp);
        int r2;
        if (ns && (r2 = bb_dev_download(ctx_, t.slices.data(), d_slices_.p, ns * sizeof(bb_slice))) != BB_OK) BB_THROW(r2, "bb_dev_download");
        if (nsp && (r2 = bb_dev_download(ctx_, t.spans.data(), d_spans_.p, (uint64_t)nsp * sizeof(bb_label_span))) != BB_OK) BB_THROW(r2, "bb_dev_download");
        if (n && (r2 = bb_dev_download(ctx_, t.status.data(), d_status_.p, n)) != BB_OK) BB_THROW(r2, "bb_dev_download");
        if (ns) {  // what the writers need to find a record's lines in the block's text
            t.line_ends.resize(4ull * n); t.id_len.resize(n); t.desc_start.resize(n);
            if ((r2 = bb_fastq_fetch_lines(ctx_, t.line_ends.data())) != BB_OK) BB_THROW(r2, "bb_fastq_fetch_lines");
            if ((r2 = bb_fastq_fetch(ctx_, nullptr, nullptr, nullptr, t.id_len.data(), t.desc_start.data(), nullptr, nullptr)) != BB_OK) BB_THROW(r2, "bb_fastq_fetch");
        }
        return t;
    }
}

namespace {
struct CompTable {  // trim.rs:486-530: A<->T C<->G R<->Y K<->M B<->V D<->H in both cases, every other byte stays
    uint8_t t[256];
    CompTable() {
        for (int i = 0; i < 256; ++i) t[i] = (uint8_t)i;
        const char* pairs = "ATCGRYKMBVDH";
        for (int i = 0; pairs[i]; i += 2) {
            const uint8_t a = (uint8_t)pairs[i], b = (uint8_t)pairs[i + 1];
            t[a] = b; t[b] = a; t[a | 0x20] = (uint8_t)(b | 0x20); t[b | 0x20] = (uint8_t)(a | 0x20);
        }
    }
};
const CompTable kComp;
}  // namespace

size_t render_trim_record(uint8_t* dst, const uint8_t* text, const TrimPlan& plan, const bb_slice& s, const bb_trim_config& cfg) {
    const uint64_t* nl = plan.line_ends.data() + 4ull * s.read_idx;
    auto span = [&](int j, uint64_t& a, uint64_t& b) {  // line j of the record without its line end
        a = (s.read_idx || j) ? nl[j - 1] + 1 : 0;
        b = nl[j];
        if (b > a && text[b - 1] == '\r') --b;
    };
    uint64_t hs, he, ss, se, qs, qe;
    span(0, hs, he); span(1, ss, se); span(3, qs, qe);
    const uint32_t hl = he > hs ? (uint32_t)(he - hs - 1) : 0u, idl = plan.id_len[s.read_idx], ds = plan.desc_start[s.read_idx];
    uint8_t* w = dst;
    *w++ = '@';
    memcpy(w, text + hs + 1, idl); w += idl;
    if (s.suffix) {  // "_n" (trim.rs:271)
        char tmp[8];
        const int k = snprintf(tmp, sizeof(tmp), "_%u", (unsigned)s.suffix);
        memcpy(w, tmp, (size_t)k); w += k;
    }
    if (cfg.write_full_header && hl > ds) { *w++ = ' '; memcpy(w, text + hs + 1 + ds, hl - ds); w += hl - ds; }
    *w++ = '\n';
    const uint32_t seq_len = (uint32_t)(se - ss);
    const uint32_t s0 = cfg.skip_trim ? 0u : s.start, L = (cfg.skip_trim ? seq_len : s.end) - s0;
    const uint8_t* sp = text + ss + s0;
    const uint8_t* qp = text + qs + s0;
    if (!s.flip) {
        memcpy(w, sp, L); w += L;
        *w++ = '\n'; *w++ = '+'; *w++ = '\n';
        memcpy(w, qp, L); w += L;
    } else {
        for (uint32_t k = 0; k < L; ++k) w[k] = kComp.t[sp[L - 1u - k]];
        w += L;
        *w++ = '\n'; *w++ = '+'; *w++ = '\n';
        for (uint32_t k = 0; k < L; ++k) w[k] = qp[L - 1u - k];
        w += L;
    }
    *w++ = '\n';
    return (size_t)(w - dst);
}

// ---- annotate (annotator.rs) ----------------------------------------------------------------------
namespace {
// the automatic flank cutoff (edit_model.rs:2-11) is applied inside bb_create when k_cutoff is unset
// Raw FASTQ text in blocks of whole records, in page-locked memory, in stream order.
//
// Chunks of `chunk` bytes are read by a pool of reader threads (pread at fixed offsets for plain files, a copy out of
// the inflated image for gzip files — ParallelInflater below) into a ring of page-locked slots, each with HEAD bytes of
// headroom in front; every reader also counts its chunk's line ends.  A sequencer (next(), one caller) takes the chunks
// in order and turns them into blocks that hold complete 4-line records only: with the running number of complete lines
// it knows how many trailing lines of a chunk belong to a record that ends in the next chunk, finds that cut by walking
// back over those few lines, and copies the short tail into the next slot's headroom — exactly the `consumed` the GPU
// parser (bb_fastq_ingest) would have reported, without waiting for it, so block i+1 can go to another GPU while block i
// is still being parsed.  The last block of a file is handed over whole (the parser's final-block rules apply to it).
// What a reader reports about the lines of one chunk it compacted (two-line mode), so that the sequencer can check what the GPU parser
// checks in the 4-line form — a record's sequence and quality lines are equally long, the file ends on a record boundary — although the
// quality lines never leave the host.  Pairs that lie inside the chunk are compared by the reader; what crosses a chunk boundary is
// stitched from these fields, in order (BlockFeeder::stitch).  Line lengths exclude the line end ("\n" or "\r\n").
struct TwoLineSummary {
    size_t head_raw = 0; uint8_t head_last = 0;   // bytes before the chunk's first '\n' (the whole chunk if it has none), the last of them
    size_t tail_raw = 0; uint8_t tail_last = 0;   // bytes after its last '\n' (a line that ends in a later chunk, or at the end of the file)
    int64_t len1 = -1, len2 = -1;                 // lengths of the second and third line that END in the chunk
    int64_t pend = -1;                            // length of the last sequence line, not the chunk's first line, whose quality line does not end in the chunk
    bool pend_cleared = false;                    // some quality line other than the chunk's first two lines ends in the chunk: nothing older is pending after it
    int64_t last3[3] = {-1, -1, -1};              // lengths of the last three lines that end in the chunk (-2: that line is the chunk's first)
};

// thrown by the sequencer when a chunk cannot be staged in the packed form (two adjacent non-IUPAC characters in a read, a gzip chunk whose
// line layout the reader could not tell): annotate() starts over with the plain two-line form
struct PackFallback {};

struct BlockFeeder {
    struct Block { uint64_t index = 0; const uint8_t* data = nullptr; size_t len = 0; int slot = -1; std::shared_ptr<std::vector<uint8_t>> big; };
    struct Task { size_t file = 0; uint64_t off = 0; size_t len = 0; uint64_t seq = 0; bool last = false; std::shared_ptr<struct GzPiece> piece; };   // piece: gzip input, off within it
    struct Slot { uint8_t* p = nullptr; size_t cap = 0, got = 0, nl = 0; bool last = false; int state = 0; uint64_t seq = 0; int refs = 0; size_t file = 0;
                  // two-line mode: raw newlines of the chunk, the phase (line index mod 4) the reader took its first byte to be in
                  // (-1: none recognisable), where the raw bytes came from (to redo the chunk if the guess was wrong), malformed flag
                  size_t raw_nl = 0; int phase0 = 0; uint64_t off = 0; size_t raw_len = 0; bool bad = false; TwoLineSummary sum;
                  bool unpackable = false; };   // packed staging: this chunk cannot be packed (PackCtx::unpackable, or no look-back possible)
    size_t HEAD = 16u << 20;  // BARBELL_AMD_HEAD_BYTES overrides it (tests of the over-long-carry path)
    int device;                // the slots are page-locked for uploads to this device; a slot is allocated by the first reader that fills it
    // Ordinary (pageable, huge-page advised) memory by default: measured on the MI355X box the runtime uploads from it as fast as from
    // page-locked memory (10.0 M reads/s steady state either way, 2 contexts, 128 MiB blocks) and page-locking 2-6 GB cost 0.4-1.0 s of a
    // 1.6-2.9 s run, serialised inside the runtime against the contexts being created.  BARBELL_AMD_PINNED_SLOTS=1: hipHostMalloc.
    bool pageable = getenv("BARBELL_AMD_PINNED_SLOTS") == nullptr;
    bool keep_slots = false;   // the process is about to exit: the destructor leaves the slots to the OS (unpinning 6 GB costs ~0.5 s)
    std::vector<std::string> paths;
    std::vector<char> is_gz;
    std::vector<int> fds;
    std::vector<const uint8_t*> maps; // two-line mode, plain files: the file mapped (the readers compact out of the page cache)
    std::vector<uint64_t> sizes;      // plain: st_size; gzip: inflated size once known
    // the part of each file this process stages: [begins, ends) — the whole file, or (--shard R/W --shard-by bytes) the records that START in
    // the R-th of W equal byte ranges of a plain file: both ends are record starts, found by the same rule from either side (record_start)
    std::vector<uint64_t> begins, ends;
    uint32_t shard_rank = 0, shard_world = 1;   // byte-range sharding (1: off)
    std::vector<char> size_known;
    size_t chunk;
    std::vector<Slot> slots;
    std::vector<std::thread> readers;
    std::unique_ptr<struct ParallelInflater> inflater;
    std::mutex mu;
    std::condition_variable cv;
    // claim cursor
    size_t cur_file = 0; uint64_t cur_off = 0, next_seq = 0;
    std::shared_ptr<struct GzPiece> cur_piece;   // gzip input: the piece being chunked (claim)
    bool piece_fetching = false;                 // a reader is waiting for the inflater's next piece; the others wait for that reader
    bool stop = false, claims_done = false;
    std::string err;
    // sequencer state
    uint64_t want_seq = 0, n_blocks = 0;
    const uint8_t* carry_ptr = nullptr; size_t carry_len = 0, carry_lines = 0; int carry_slot = -1;
    std::vector<uint8_t> carry_buf;  // a carry that spans whole chunks (a record longer than a chunk) is kept here
    bool done = false;
    // Two-line mode (annotate without the trim step: annotator.rs:125-127 never looks at the quality line): the readers drop the
    // '+' and quality lines while they stage a chunk, so half the bytes cross PCIe and the GPU parses 2-line records
    // (BB_FASTQ_TWO_LINE).  Dropping lines is a pure per-byte filter on "index of the byte's line mod 4", so the compacted chunks
    // concatenate to the compacted stream; a reader only has to know the phase of its chunk's first byte.  It reads it off the
    // text ("@..." two lines above "+..."; a sequence line cannot start with '+', so the test is unambiguous for FASTQ) and the
    // sequencer, which knows the true phase from the running line count, checks every guess and redoes a chunk that was wrong.
    bool two_line = false;
    bool pack = false;                    // two-line mode with the sequence lines packed two bases per byte (PackCtx); needs the raw text in memory
    size_t lpr = 4;                       // lines per record in the staged text
    size_t seq_file = (size_t)-1; uint64_t seq_raw_lines = 0;   // sequencer: file in hand, its raw lines so far
    // stitch(): a line in progress across chunk ends, the sequence length waiting for its quality line, the last two lines' lengths
    size_t st_part = 0; uint8_t st_part_last = 0; int64_t st_pend = -1, st_last3[3] = {-1, -1, -1};
    void stitch(const Slot& sl, int ph0);

    BlockFeeder(int device_, const std::vector<std::string>& files, size_t chunk_bytes, unsigned n_slots, unsigned n_readers, unsigned n_inflate,
                bool two_line_mode = false, bool pack_mode = false, uint32_t byte_shard_rank = 0, uint32_t byte_shard_world = 1);
    ~BlockFeeder();
    static uint64_t record_start(int fd, uint64_t size, uint64_t pos, const std::string& path);
    // where in its line byte `off` of a file lies: the readers look back for the line's start (mapped file / inflated image)
    static size_t line_pos(const uint8_t* file_base, uint64_t off) {
        const void* q = off ? memrchr(file_base, '\n', (size_t)off) : nullptr;
        return q ? (size_t)(file_base + off - ((const uint8_t*)q + 1)) : (size_t)off;
    }
    void reader_loop();
    bool claim(Task& t);
    bool next(Block& b);
    void release(int slot);
    void unref(int slot);
    void fail(const std::string& e) { { std::lock_guard<std::mutex> lk(mu); if (err.empty() && !e.empty()) err = e; stop = true; } cv.notify_all(); }
};

// gzip files (and pipes): zlib inflates one stream on one core (~0.3 GB/s), far below what the GPU takes, so a pool of threads inflates
// several files at once, a few files ahead of the consumer, and hands them over in input order (the TSV keeps the reads' order).  A file
// comes as PIECES of at most `piece_bytes` of text, each cut after a whole record (the file starts with one, so the cut is after line
// 4 * floor(lines / 4) of what has been read): the feeder treats a piece like a small file of its own.  Round 5: before, a file was inflated
// whole — a 50 GB fastq.gz meant 200 GB of text in memory (and its buffer's last doubling as much again); now a file holds at most
// `pieces_ahead` pieces whatever its size.  n_threads comes from -t/--threads like the reference's worker count.
static size_t count_nl(const uint8_t* p, size_t n);
// libdeflate, where the system has it (dlopen: no build dependency; BARBELL_AMD_NO_LIBDEFLATE=1 switches it off): inflates a gzip member
// whose text fits a buffer 4-5 x as fast as zlib (320 MB of FASTQ text in 32 members: 0.41 against 1.84 s on one core).  It has no
// streaming form, so members too large to buffer — and systems without the library — take zlib as before.
struct LibDeflate {
    void* (*alloc)() = nullptr;
    void (*release)(void*) = nullptr;
    int (*gzip_ex)(void*, const void*, size_t, void*, size_t, size_t*, size_t*) = nullptr;   // 0 ok, 1 bad data, 3 insufficient space
    // the other direction (the per-label files of --gzip): a span of records -> one gzip member
    void* (*calloc_)(int) = nullptr;
    void (*cfree)(void*) = nullptr;
    size_t (*gzip_bound)(void*, size_t) = nullptr;
    size_t (*gzip_compress)(void*, const void*, size_t, void*, size_t) = nullptr;
    static const LibDeflate& get() {
        static const LibDeflate L = []() {
            LibDeflate l;
            if (getenv("BARBELL_AMD_NO_LIBDEFLATE")) return l;
            void* h = nullptr;
            for (const char* name : {"libdeflate.so.0", "libdeflate.so"}) if ((h = dlopen(name, RTLD_NOW | RTLD_LOCAL))) break;
            if (!h) return l;
            l.alloc = (void* (*)())dlsym(h, "libdeflate_alloc_decompressor");
            l.release = (void (*)(void*))dlsym(h, "libdeflate_free_decompressor");
            l.gzip_ex = (int (*)(void*, const void*, size_t, void*, size_t, size_t*, size_t*))dlsym(h, "libdeflate_gzip_decompress_ex");
            if (!l.alloc || !l.release || !l.gzip_ex) l.gzip_ex = nullptr;
            l.calloc_ = (void* (*)(int))dlsym(h, "libdeflate_alloc_compressor");
            l.cfree = (void (*)(void*))dlsym(h, "libdeflate_free_compressor");
            l.gzip_bound = (size_t (*)(void*, size_t))dlsym(h, "libdeflate_gzip_compress_bound");
            l.gzip_compress = (size_t (*)(void*, const void*, size_t, void*, size_t))dlsym(h, "libdeflate_gzip_compress");
            if (!l.calloc_ || !l.cfree || !l.gzip_bound || !l.gzip_compress) l.gzip_compress = nullptr;
            return l;
        }();
        return L;
    }
    struct Comp { void* c = nullptr; ~Comp() { if (c) LibDeflate::get().cfree(c); } };
    static void* compressor() {   // one per thread, level 6 (zlib's default, what gzopen("wb") writes with)
        static thread_local Comp t;
        if (!t.c && get().gzip_compress) t.c = get().calloc_(6);
        return t.c;
    }
    struct Dec { void* d = nullptr; ~Dec() { if (d) LibDeflate::get().release(d); } };
    static void* decompressor() {   // one per thread
        static thread_local Dec t;
        if (!t.d && get().gzip_ex) t.d = get().alloc();
        return t.d;
    }
};
// bytes without the zero-fill a std::vector pays on every growth (the inflaters write every byte they count; at GB/s the fill was a third of the time)
struct RawBuf {
    uint8_t* p = nullptr; size_t cap = 0;
    RawBuf() = default;
    RawBuf(const RawBuf&) = delete;
    RawBuf& operator=(const RawBuf&) = delete;
    ~RawBuf() { free(p); }
    void reserve(size_t c) {   // keeps the contents
        if (c <= cap) return;
        void* q = realloc(p, c);
        if (!q) throw BarbellError(BB_E_NOMEM, "out of memory (inflated text)");
        p = (uint8_t*)q; cap = c;
    }
    uint8_t* data() { return p; }
    const uint8_t* data() const { return p; }
    void swap(RawBuf& o) { std::swap(p, o.p); std::swap(cap, o.cap); }
};
// piece buffers go round: a fresh 256 MiB buffer costs 65 K page faults on first touch (a third of the coordinator's time per piece), a used one none
struct BufPool {
    std::mutex mu;
    std::vector<std::unique_ptr<RawBuf>> spare;
    void give(RawBuf& b) {
        if (b.cap < (16u << 20)) return;
        std::lock_guard<std::mutex> lk(mu);
        if (spare.size() >= 8) return;
        spare.emplace_back(new RawBuf());
        spare.back()->swap(b);
    }
    bool take(RawBuf& into, size_t want) {   // a spare buffer of at least `want` bytes, if there is one
        std::lock_guard<std::mutex> lk(mu);
        for (size_t i = 0; i < spare.size(); ++i)
            if (spare[i]->cap >= want) { into.swap(*spare[i]); spare.erase(spare.begin() + (long)i); return true; }
        return false;
    }
};
struct GzPiece {
    RawBuf data; size_t size = 0; uint64_t chunks_left = 0;
    std::shared_ptr<BufPool> pool;
    ~GzPiece() { if (pool) pool->give(data); }
};
struct ParallelInflater {
    std::vector<std::string> paths;
    std::vector<char> gz;    // files that are not gzip are skipped (the feeder reads them directly)
    struct FileState { std::deque<std::shared_ptr<GzPiece>> ready; int state = 0; /* 0 not started, 1 being inflated, 2 all pieces made */ size_t out = 0; /* pieces made, not yet consumed */ };
    std::vector<FileState> fs;
    std::vector<std::thread> pool;
    std::mutex mu;
    std::condition_variable cv;
    size_t next_file = 0, consumed_upto = 0, ahead;
    size_t piece_bytes = 256u << 20, pieces_ahead = 3;
    std::string err;
    bool cancelled = false;
    std::shared_ptr<BufPool> bufs = std::make_shared<BufPool>();
    unsigned n_threads_total = 1; size_t n_gz_files = 0;
    uint64_t range_bytes = 32u << 20;   // compressed bytes per range of the member-parallel inflate; BARBELL_AMD_GZ_RANGE (tests) fixes it and scales the limits with it
    bool range_cap = false;
    std::atomic<uint64_t> n_ranges_parallel{0};
    uint64_t n_pieces = 0, held = 0, max_held = 0;   // pieces made; bytes of inflated text made and not yet consumed, and the most there ever was (BARBELL_AMD_PROFILE)
    ParallelInflater(std::vector<std::string> p, std::vector<char> is_gz, unsigned n_threads)
        : paths(std::move(p)), gz(std::move(is_gz)), fs(paths.size()) {
        if (const char* e = getenv("BARBELL_AMD_GZ_PIECE")) piece_bytes = (size_t)std::max(64L, atol(e));   // tests: pieces of a few hundred bytes
        if (const char* e = getenv("BARBELL_AMD_GZ_RANGE")) { range_bytes = (uint64_t)std::max(64L, atol(e)); range_cap = true; }  // tests: ranges of a few hundred bytes; 0 threads' worth: BARBELL_AMD_GZ_SERIAL
        n_threads_total = getenv("BARBELL_AMD_GZ_SERIAL") ? 1u : std::max(1u, n_threads);
        for (char g : gz) n_gz_files += g ? 1 : 0;
        const unsigned nt = std::max(1u, std::min<unsigned>(n_threads, (unsigned)paths.size()));
        ahead = nt + 2;  // files being inflated or inflated and not yet consumed: bounds the memory
        for (unsigned i = 0; i < nt; ++i) pool.emplace_back([this]() { work(); });
    }
    // hands a piece over; waits while the file has pieces_ahead of them unconsumed.  false: cancelled
    bool publish(size_t i, std::shared_ptr<GzPiece> pc) {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&]() { return cancelled || !err.empty() || fs[i].out < pieces_ahead; });
        if (cancelled || !err.empty()) return false;
        ++n_pieces; held += pc->size; max_held = std::max(max_held, held);
        fs[i].ready.push_back(std::move(pc)); ++fs[i].out;
        lk.unlock();
        cv.notify_all();
        return true;
    }
    // Inflated text in, record-aligned pieces out: a piece is published when piece_bytes of text are there, cut after the last whole record
    // (the file starts with one, so the cut is after line 4 * floor(lines / 4) of what the piece holds); a record larger than a piece makes it grow.
    struct PieceSink {
        ParallelInflater& P; size_t file;
        RawBuf buf; size_t n = 0; bool any = false;
        bool no_fast = false;   // a member of this file did not fit libdeflate's buffers: zlib for the rest (inflate_member)
        PieceSink(ParallelInflater& p, size_t f) : P(p), file(f) {}
        size_t first_cap() const { return std::max<size_t>(64, std::min<size_t>(P.piece_bytes, 4u << 20)); }   // (small files do not pay for a piece-sized buffer)
        bool emit(size_t cut) {   // buf[0, cut) goes out as a piece, the rest starts the next
            auto pc = std::make_shared<GzPiece>();
            pc->pool = P.bufs;
            RawBuf next;
            const size_t rest = n - cut;
            if (!P.bufs->take(next, std::max<size_t>(P.piece_bytes, rest + 64))) next.reserve(std::max(first_cap(), rest + 64));
            if (rest) memcpy(next.data(), buf.data() + cut, rest);
            pc->size = cut; pc->data.swap(buf);
            buf.swap(next);
            any = true; n = rest;
            return P.publish(file, std::move(pc));
        }
        bool append(const uint8_t* p, size_t len) {
            while (len) {
                if (n == buf.cap) {
                    const size_t full = std::max<size_t>(P.piece_bytes, 64);
                    if (buf.cap < full) { buf.reserve(buf.cap ? std::min(full, buf.cap * 2) : first_cap()); continue; }
                    const size_t lines = count_nl(buf.data(), n);
                    if (lines < 4) { buf.reserve(buf.cap * 2); continue; }   // a record larger than the piece
                    const uint8_t* end = buf.data() + n;
                    for (size_t k = 0; k <= lines % 4; ++k) {   // back over the partial last line and the lines % 4 whole ones after the last record
                        end = (const uint8_t*)memrchr(buf.data(), '\n', (size_t)(end - buf.data()));   // not null: lines >= 4
                        if (k == lines % 4) ++end;                                                      // ... up to and including the record's last line end
                    }
                    if (!emit((size_t)(end - buf.data()))) return false;
                    continue;
                }
                const size_t take = std::min(len, buf.cap - n);
                memcpy(buf.data() + n, p, take); n += take; p += take; len -= take;
            }
            return true;
        }
        bool finish() {   // the end of the file: whatever is left (a last line without a line end, blank lines); an empty file gives one empty piece
            if (n == 0 && any) return true;
            auto pc = std::make_shared<GzPiece>();
            pc->pool = P.bufs;
            pc->size = n; pc->data.swap(buf);
            any = true; n = 0;
            return P.publish(file, std::move(pc));
        }
    };
    // ---- a regular gzip file of several MEMBERS (what `cat *.fastq.gz` and bgzip make; a sequencing run's files concatenated) on several cores.
    // A member's start cannot be read off the file, so the threads guess: the compressed bytes are cut into ranges, a range's thread looks for
    // the gzip magic from the range's start on and inflates what it finds, member after member, until one starts at or beyond the range's
    // end.  A false start fails within a few KB (or, with probability 2^-64, passes the member's CRC and length).  The coordinator takes the
    // ranges in order and accepts a range's text only if its first member starts exactly where the text accepted so far ended: every byte
    // of the file is then accounted for by members that inflated with their checksums right.  Anything else — a range whose chain does not
    // link up, a member too large to buffer (a single-member file), an error — and the rest of the file is inflated serially from the last
    // accepted position, in bounded pieces as before.
    struct GzRange {
        uint64_t a = 0, b = 0;             // members that START in [a, b)
        uint64_t first = UINT64_MAX, end = 0;   // compressed offsets: start of the first member found, end of the last one inflated
        RawBuf text; size_t text_n = 0;
        bool done = false, failed = false, garbage = false;   // garbage: what follows `end` is not a gzip member (trailing bytes: ignored, as gzread does)
    };
    static bool gz_magic(const uint8_t* p, uint64_t left) { return left >= 18 && p[0] == 0x1f && p[1] == 0x8b && p[2] == 8 && (p[3] & 0xE0) == 0; }
    // one member at map[pos ..]: its text appended to `out` (or handed to `sink` as it comes); returns the compressed bytes it took, 0 on error
    // or when it grows beyond the limits (max_in compressed bytes / max_out bytes of text in `out`)
    static uint64_t inflate_member(const uint8_t* map, uint64_t size, uint64_t pos, RawBuf* out, size_t* out_n, PieceSink* sink, uint64_t max_in, uint64_t max_out) {
        // libdeflate first: the member's text into a buffer of guessed size (doubled while it says the space does not suffice)
        if (void* dec = (sink && sink->no_fast) ? nullptr : LibDeflate::decompressor()) {
            const uint64_t left = size - pos;
            const size_t in_n = (size_t)std::min<uint64_t>(left, out ? std::min<uint64_t>(max_in, 1ull << 40) + 65536 : (512ull << 20));
            static const uint64_t serial_cap = getenv("BARBELL_AMD_LIBDEFLATE_MAX") ? (uint64_t)atoll(getenv("BARBELL_AMD_LIBDEFLATE_MAX")) : (1024ull << 20);   // (tests: a small one)
            const uint64_t cap_out = out ? max_out + (16ull << 20) : serial_cap;
            static thread_local RawBuf tmp;   // (serial path)
            const size_t have = out ? *out_n : 0;
            RawBuf& dst = out ? *out : tmp;
            // (serial path: a member that does not fit 512 MB of input or 1 GiB of text is zlib's, and so is the rest of its file: a huge single-member
            // file pays for one failed attempt, ~2 s, not for one per member)
            const bool whole = true;
            for (uint64_t space = std::min<uint64_t>(cap_out, std::max<uint64_t>(16ull << 20, 6ull * std::min<uint64_t>(in_n, 64ull << 20))); whole; space = std::min(cap_out, space * 2)) {
                dst.reserve(have + (size_t)space);
                size_t a_in = 0, a_out = 0;
                const int r = LibDeflate::get().gzip_ex(dec, map + pos, in_n, dst.data() + have, (size_t)space, &a_in, &a_out);
                if (r == 0) {
                    if (out) { *out_n = have + a_out; return a_in; }
                    return sink->append(tmp.data(), a_out) ? a_in : 0;
                }
                if (r != 3 || space >= cap_out) break;   // bad data (or cut off by the window), or larger than what may be buffered: zlib decides
            }
            if (sink) sink->no_fast = true;
        }
        z_stream zs;
        memset(&zs, 0, sizeof(zs));
        if (inflateInit2(&zs, 15 + 16) != Z_OK) return 0;
        uint64_t in_done = 0;
        std::vector<uint8_t> tmp(sink ? (4u << 20) : 0);
        const size_t out0 = out ? *out_n : 0;
        uint64_t result = 0;
        for (;;) {
            if (zs.avail_in == 0) {
                const uint64_t left = size - pos - in_done;
                if (left == 0) break;                       // the file ends inside the member
                zs.next_in = const_cast<Bytef*>(map + pos + in_done);
                zs.avail_in = (uInt)std::min<uint64_t>(left, 1u << 30);
                in_done += zs.avail_in;
            }
            if (out) {
                if (*out_n - out0 > max_out) break;
                const size_t have = *out_n;
                if (out->cap - have < (1u << 20)) out->reserve(std::max<size_t>(out->cap + out->cap / 2, have + (16u << 20)));
                const uInt room = (uInt)std::min<size_t>(out->cap - have, 1u << 30);
                zs.next_out = out->data() + have; zs.avail_out = room;
                const int r = inflate(&zs, Z_NO_FLUSH);
                *out_n = have + (room - zs.avail_out);
                if (r == Z_STREAM_END) { result = in_done - zs.avail_in; break; }
                if (r != Z_OK && r != Z_BUF_ERROR) break;
            } else {
                zs.next_out = tmp.data(); zs.avail_out = (uInt)tmp.size();
                const int r = inflate(&zs, Z_NO_FLUSH);
                if (!sink->append(tmp.data(), tmp.size() - zs.avail_out)) break;
                if (r == Z_STREAM_END) { result = in_done - zs.avail_in; break; }
                if (r != Z_OK && r != Z_BUF_ERROR) break;
            }
            if (in_done - zs.avail_in > max_in) break;
        }
        inflateEnd(&zs);
        if (!result && out) *out_n = out0;
        return result;
    }
    void range_work(const uint8_t* map, uint64_t size, GzRange& R, uint64_t range_bytes) {
        uint64_t p = R.a;
        // a member beyond these is left to the serial path (a single-member file: the ranges' threads would buffer all of it)
        const uint64_t max_in = range_cap ? 4 * range_bytes : (256ull << 20), max_out = range_cap ? 64 * range_bytes : (1536ull << 20);
        // the first member: the first candidate from the range's start on that inflates to its end with its checksum right
        while (p < R.b && p < size) {
            const uint8_t* q = (const uint8_t*)memchr(map + p, 0x1f, (size_t)std::min<uint64_t>(R.b, size) - p);
            if (!q) { p = R.b; break; }
            p = (uint64_t)(q - map);
            if (gz_magic(map + p, size - p)) {
                const uint64_t took = inflate_member(map, size, p, &R.text, &R.text_n, nullptr, max_in, max_out);
                if (took) { R.first = p; p += took; break; }
                if (R.a == 0 && p == 0) { R.failed = true; return; }   // the file's own first member does not inflate: leave it to the serial path and its error message
            }
            ++p;
        }
        if (R.first == UINT64_MAX) { R.end = R.a; return; }
        // the chain: member after member until one starts at or beyond the range's end
        while (p < R.b && p < size) {
            if (!gz_magic(map + p, size - p)) { R.garbage = true; break; }
            const uint64_t took = inflate_member(map, size, p, &R.text, &R.text_n, nullptr, max_in, max_out);
            if (!took) { R.failed = true; break; }
            p += took;
            if (R.text_n > max_out) { R.failed = true; break; }
        }
        R.end = p;
    }
    // the rest of the file from compressed offset pos, member after member on this thread, text straight into the sink; "" or an error message
    std::string inflate_serial(const uint8_t* map, uint64_t size, uint64_t pos, PieceSink& sink, const std::string& path) {
        while (pos < size) {
            if (!gz_magic(map + pos, size - pos)) {
                if (pos == 0) return "Error reading FASTQ file '" + path + "'";
                break;   // trailing bytes that are no gzip member: ignored (gzread does the same)
            }
            const uint64_t took = inflate_member(map, size, pos, nullptr, nullptr, &sink, UINT64_MAX, 0);
            if (!took) {
                std::lock_guard<std::mutex> lk(mu);
                return cancelled ? std::string() : "Error reading FASTQ file '" + path + "' (gzip data corrupt or truncated)";
            }
            pos += took;
        }
        return std::string();
    }
    std::string inflate_regular(size_t i, unsigned n_range_threads) {
        const int fd = open(paths[i].c_str(), O_RDONLY);
        struct stat st;
        if (fd < 0 || fstat(fd, &st) != 0) { if (fd >= 0) close(fd); return "Failed to open FASTQ input: " + paths[i]; }
        const uint64_t size = (uint64_t)st.st_size;
        PieceSink sink(*this, i);
        if (size == 0) { close(fd); sink.finish(); return std::string(); }
        void* m = mmap(nullptr, (size_t)size, PROT_READ, MAP_PRIVATE, fd, 0);
        if (m == MAP_FAILED) { close(fd); return "Failed to map FASTQ input: " + paths[i]; }
        (void)madvise(m, (size_t)size, MADV_SEQUENTIAL);
        const uint8_t* map = (const uint8_t*)m;
        uint64_t pos = 0;           // compressed bytes accounted for
        std::string e;
        const auto t_start = std::chrono::steady_clock::now();
        // ranges of 4 .. 32 MB of compressed bytes, two per thread for a mid-sized file; files under 16 MB are not worth the threads
        const uint64_t RB = range_cap ? range_bytes : std::min<uint64_t>(32u << 20, std::max<uint64_t>(4u << 20, size / (2ull * std::max(1u, n_range_threads))));
        if (n_range_threads >= 2 && size >= (range_cap ? 4 * RB : (16ull << 20))) {
            const size_t K = (size_t)((size + RB - 1) / RB);
            std::vector<std::unique_ptr<GzRange>> rg(K);
            std::mutex rmu; std::condition_variable rcv;
            size_t next_job = 0, taken_upto = 0; bool quit = false;
            const size_t window = n_range_threads + 2;   // ranges inflated or being inflated beyond the one the coordinator waits for: bounds the memory
            std::vector<std::thread> th;
            for (unsigned t = 0; t < n_range_threads; ++t)
                th.emplace_back([&]() {
                    for (;;) {
                        size_t k;
                        {
                            std::unique_lock<std::mutex> lk(rmu);
                            rcv.wait(lk, [&]() { return quit || (next_job < K && next_job < taken_upto + window); });
                            if (quit) return;
                            k = next_job++;
                            rg[k] = std::make_unique<GzRange>();
                            rg[k]->a = (uint64_t)k * RB; rg[k]->b = std::min<uint64_t>(size, (uint64_t)(k + 1) * RB);
                        }
                        range_work(map, size, *rg[k], RB);
                        { std::lock_guard<std::mutex> lk(rmu); rg[k]->done = true; }
                        rcv.notify_all();
                    }
                });
            bool ok = true, ended = false;
            for (size_t k = 0; k < K && ok && !ended; ++k) {
                std::unique_ptr<GzRange> R;
                {
                    std::unique_lock<std::mutex> lk(rmu);
                    rcv.wait(lk, [&]() { return rg[k] && rg[k]->done; });
                    R = std::move(rg[k]);
                    taken_upto = k + 1;
                }
                rcv.notify_all();
                if (R->failed) { ok = false; break; }
                if (R->first == UINT64_MAX) { ok = pos >= R->b; continue; }   // no member starts here: fine if the chain so far reaches past the range
                if (R->first != pos) { ok = false; break; }                         // does not link up with what has been accepted
                if (!sink.append(R->text.data(), R->text_n)) { ok = false; ended = true; break; }
                pos = R->end;
                ++n_ranges_parallel;
                if (R->garbage) ended = true;
            }
            { std::lock_guard<std::mutex> lk(rmu); quit = true; }
            rcv.notify_all();
            for (auto& t : th) t.join();
            if (ended && ok) pos = size;   // trailing bytes after the last member: ignored
        }
        { std::lock_guard<std::mutex> lk(mu); if (cancelled) { munmap(m, (size_t)size); close(fd); return std::string(); } }
        if (getenv("BARBELL_AMD_PROFILE"))
            fprintf(stderr, "profile: '%s': %llu of %llu compressed bytes inflated by ranges in %.3f s\n", paths[i].c_str(), (unsigned long long)pos, (unsigned long long)size,
                    std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count());
        if (pos < size) e = inflate_serial(map, size, pos, sink, paths[i]);
        munmap(m, (size_t)size); close(fd);
        if (e.empty()) sink.finish();
        if (getenv("BARBELL_AMD_PROFILE"))
            fprintf(stderr, "profile: '%s' inflated in %.3f s (its thread's time, waits for the consumer included)\n", paths[i].c_str(),
                    std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count());
        return e;
    }
    std::string inflate_stream(size_t i) {   // pipes, process substitutions, /dev/stdin: sequentially through gzread (plain text passes through)
        gzFile f = gzopen(paths[i].c_str(), "rb");
        if (!f) return "Failed to open FASTQ input: " + paths[i];
        gzbuffer(f, 1 << 20);
        PieceSink sink(*this, i);
        std::vector<uint8_t> tmp(4u << 20);
        std::string e;
        for (;;) {
            const int r = gzread(f, tmp.data(), (unsigned)tmp.size());
            if (r < 0) { e = "Error reading FASTQ file '" + paths[i] + "'"; break; }
            if (r == 0) break;
            if (!sink.append(tmp.data(), (size_t)r)) break;
        }
        gzclose(f);
        if (e.empty()) sink.finish();
        return e;
    }
    void work() {
        for (;;) {
            size_t i;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [this]() { return cancelled || next_file >= paths.size() || next_file < consumed_upto + ahead || !err.empty(); });
                if (cancelled || next_file >= paths.size() || !err.empty()) return;
                i = next_file++;
                if (!gz[i]) { fs[i].state = 2; continue; }
                fs[i].state = 1;
            }
            struct stat pst;
            const bool regular = stat(paths[i].c_str(), &pst) == 0 && S_ISREG(pst.st_mode);
            // threads for one file's ranges: all of them for a single gzip input, fewer where several files are being inflated side by side
            const unsigned per_file = std::max(1u, n_threads_total / (unsigned)std::max<size_t>(1, std::min<size_t>(n_gz_files, n_threads_total)));
            const std::string e = regular ? inflate_regular(i, per_file) : inflate_stream(i);
            {
                std::lock_guard<std::mutex> lk(mu);
                if (!e.empty() && err.empty()) err = e;
                fs[i].state = 2;
            }
            cv.notify_all();
        }
    }
    // the file's next piece, in order; blocks until it is there.  nullptr: the file has no more (an empty file gives one empty piece first)
    std::shared_ptr<GzPiece> next_piece(size_t i) {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&]() { return cancelled || !err.empty() || !fs[i].ready.empty() || fs[i].state == 2; });
        if (!err.empty()) throw BarbellError(BB_E_INVALID, err);
        if (cancelled) throw BarbellError(BB_E_INVALID, "cancelled");
        if (fs[i].ready.empty()) {   // every piece has been handed out: the next file may start (its pieces already made stay until consumed)
            consumed_upto = std::max(consumed_upto, i + 1);
            lk.unlock();
            cv.notify_all();
            return nullptr;
        }
        auto pc = fs[i].ready.front();
        fs[i].ready.pop_front();
        return pc;
    }
    // every chunk of a piece has been copied out: its memory goes with the last reference, the file may make another
    void piece_consumed(size_t i, size_t bytes) {
        { std::lock_guard<std::mutex> lk(mu); --fs[i].out; held -= bytes; }
        cv.notify_all();
    }
    // wakes everything that waits here (workers, and readers inside next_piece): called before the feeder joins its readers
    void cancel() {
        {
            std::lock_guard<std::mutex> lk(mu);
            next_file = paths.size();
            cancelled = true;
        }
        cv.notify_all();
    }
    ~ParallelInflater() {
        if (getenv("BARBELL_AMD_PROFILE"))
            fprintf(stderr, "profile: gzip / pipe input inflated in %llu piece(s) of at most %zu bytes; at most %llu bytes of inflated text held at once; %llu range(s) of members inflated side by side\n",
                    (unsigned long long)n_pieces, piece_bytes, (unsigned long long)max_held, (unsigned long long)n_ranges_parallel.load());
        if (getenv("BARBELL_AMD_PROFILE")) fprintf(stderr, "profile: gzip members inflated with %s\n", LibDeflate::get().gzip_ex ? "libdeflate (zlib for members too large to buffer)" : "zlib");
        cancel();
        for (auto& t : pool) if (t.joinable()) t.join();
    }
};

struct LabelWriters {  // the per-label writers of trim_matches (trim.rs:356-446)
    std::string folder;
    bool gz;
    // Writes run on K threads of their own (one write() stream moves ~3-6 GB/s of page cache, the GPU plans or renders records many
    // times faster).  A label's file is worked on by one thread at a time and its spans in submission order, so the records of a file
    // stay in batch order and no two threads share a handle; WHICH thread is decided when the work is there (labels with queued spans
    // wait in `ready`, a free thread takes the next one and drains it), so that no thread idles while another has twelve labels of
    // a block to itself.  A batch is done when the last of its spans is written; at most `max_outstanding` batches exist (wait()),
    // which bounds the text held in page-locked buffers.
    // A span is either bytes to write as they are (rendered on the GPU) or, with `cut`, the records slices[first, first + n_records) to be cut out
    // of the block's own text first (host_cut): the thread renders them into its buffer — n bytes, laid out by the plan's offsets — and writes that
    struct Cut { const uint8_t* text; TrimPlan plan; bb_trim_config cfg; std::shared_ptr<void> hold; };
    struct Span { std::string label; const uint8_t* p; size_t n; std::shared_ptr<void> keep; std::shared_ptr<const Cut> cut; uint64_t first = 0, off = 0; uint32_t n_records = 0; };
    struct Item { Span sp; std::shared_ptr<std::atomic<int>> left; };
    struct LabelQ {
        std::deque<Item> q;
        bool busy = false, listed = false;
        gzFile gzf = nullptr;
        FILE* plain = nullptr;
    };
    std::map<std::string, std::unique_ptr<LabelQ>> labels;
    std::deque<LabelQ*> ready;
    std::vector<std::thread> threads;
    std::mutex mu;
    std::condition_variable cv;
    bool stop = false;
    size_t outstanding = 0;  // batches submitted and not yet fully written
    std::string err;
    LabelWriters(std::string f, bool g, unsigned k = 0) : folder(std::move(f)), gz(g) {
        if (k == 0) { const char* e = getenv("BARBELL_AMD_WRITERS"); k = e ? (unsigned)std::max(1, atoi(e)) : 8u; }
        for (unsigned i = 0; i < k; ++i) threads.emplace_back([this]() { run(); });
    }
    void run() {
        std::vector<uint8_t> buf;
        for (;;) {
            LabelQ* L;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&]() { return stop || !ready.empty(); });
                if (ready.empty()) return;
                L = ready.front();
                ready.pop_front();
                L->listed = false; L->busy = true;
            }
            for (;;) {
                Item it;
                {
                    std::lock_guard<std::mutex> lk(mu);
                    if (L->q.empty()) { L->busy = false; break; }
                    it = std::move(L->q.front());
                    L->q.pop_front();
                }
                try {
                    if (it.sp.cut) {
                        const Cut& c = *it.sp.cut;
                        if (buf.size() < it.sp.n) buf.resize(it.sp.n + it.sp.n / 4);
                        size_t total = 0;
                        for (uint64_t k = it.sp.first; k < it.sp.first + it.sp.n_records; ++k) {
                            const bb_slice& sl = c.plan.slices[k];
                            const size_t got = render_trim_record(buf.data() + (sl.out_off - it.sp.off), c.text, c.plan, sl, c.cfg);
                            if (got != sl.rec_len) throw BarbellError(BB_E_INVALID, "internal: a record cut on the host differs in length from the GPU's plan");
                            total += got;
                        }
                        if (total != it.sp.n) throw BarbellError(BB_E_INVALID, "internal: records of a label do not fill its span");
                        write(*L, it.sp.label, buf.data(), it.sp.n);
                    } else write(*L, it.sp.label, it.sp.p, it.sp.n);
                } catch (const std::exception& e) {
                    std::lock_guard<std::mutex> lk(mu);
                    if (err.empty()) err = e.what();
                }
                it.sp.keep.reset();
                it.sp.cut.reset();
                if (it.left->fetch_sub(1) == 1) {
                    { std::lock_guard<std::mutex> lk(mu); --outstanding; }
                    cv.notify_all();
                }
            }
        }
    }
    // waits until at most `max_outstanding` batches are queued or being written, then rethrows a writer error if any
    void wait(size_t max_outstanding) {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&]() { return outstanding <= max_outstanding; });
        if (!err.empty()) throw BarbellError(BB_E_INVALID, err);
    }
    void submit(std::vector<Span> job) {
        if (job.empty()) return;
        auto left = std::make_shared<std::atomic<int>>((int)job.size());
        {
            std::lock_guard<std::mutex> lk(mu);
            ++outstanding;
            for (auto& sp : job) {
                auto& slot = labels[sp.label];
                if (!slot) slot = std::make_unique<LabelQ>();
                LabelQ* L = slot.get();
                L->q.push_back(Item{std::move(sp), left});
                if (!L->busy && !L->listed) { L->listed = true; ready.push_back(L); }   // a busy label's thread finds the new span itself
            }
        }
        cv.notify_all();
    }
    void write(LabelQ& L, const std::string& label, const uint8_t* p, size_t n) {
        const std::string path = folder + "/" + label + (gz ? ".trimmed.fastq.gz" : ".trimmed.fastq");
        if (gz && LibDeflate::compressor()) {
            // a span = one gzip member appended to the label's file (concatenated members are one gzip file): libdeflate compresses 2-3 x as fast
            // as zlib at the same level, and the writer threads are what `kit --gzip` waits for
            if (!L.plain) {
                L.plain = fopen(path.c_str(), "wb");
                if (!L.plain) throw BarbellError(BB_E_INVALID, "Failed to create output file '" + path + "'\nTry setting ulimit higher: \"ulimit -n 65000\"");
                setvbuf(L.plain, nullptr, _IONBF, 0);
            }
            static thread_local RawBuf zb;
            void* c = LibDeflate::compressor();
            size_t o = 0;
            do {   // (an empty span still leaves a member: the file is a gzip file from its first write on, as with gzopen)
                const size_t chunk = std::min<size_t>(n - o, 256u << 20);
                zb.reserve(LibDeflate::get().gzip_bound(c, chunk) + 64);
                const size_t z = LibDeflate::get().gzip_compress(c, p + o, chunk, zb.data(), zb.cap);
                if (!z || fwrite(zb.data(), 1, z, L.plain) != z) throw BarbellError(BB_E_INVALID, "Failed to write sequence to '" + path + "'");
                o += chunk;
            } while (o < n);
        } else if (gz) {
            if (!L.gzf) {
                L.gzf = gzopen(path.c_str(), "wb");
                if (!L.gzf) throw BarbellError(BB_E_INVALID, "Failed to create output file '" + path + "'\nTry setting ulimit higher: \"ulimit -n 65000\"");
            }
            for (size_t o = 0; o < n;) {
                const unsigned chunk = (unsigned)std::min<size_t>(n - o, 1u << 30);
                if (gzwrite(L.gzf, p + o, chunk) <= 0) throw BarbellError(BB_E_INVALID, "Failed to write sequence to '" + path + "'");
                o += chunk;
            }
        } else {
            if (!L.plain) {
                L.plain = fopen(path.c_str(), "wb");
                if (!L.plain) throw BarbellError(BB_E_INVALID, "Failed to create output file '" + path + "'\nTry setting ulimit higher: \"ulimit -n 65000\"");
                setvbuf(L.plain, nullptr, _IONBF, 0);  // spans are large and contiguous: straight to write()
            }
            if (n && fwrite(p, 1, n, L.plain) != n) throw BarbellError(BB_E_INVALID, "Failed to write sequence to '" + path + "'");
        }
    }
    ~LabelWriters() {
        {
            std::lock_guard<std::mutex> lk(mu);
            stop = true;
        }
        cv.notify_all();
        for (auto& t : threads) if (t.joinable()) t.join();  // they drain the ready list first
        for (auto& kv : labels) {
            if (kv.second->gzf) gzclose(kv.second->gzf);
            if (kv.second->plain) fclose(kv.second->plain);
        }
    }
};

// Packed staging (BB_FASTQ_PACKED, include/barbell_amd_fastq.h): the kernels only ever look at a read character's IUPAC base set, so the
// readers keep exactly that — two 4-bit codes per byte — and the sequence lines cross PCIe at half their size again (~2 KB per 4-kb read;
// annotator.rs:125-127 hands `demux` the bytes, nothing downstream of it reports them).  Pairs are aligned to the START of a line, so that
// the packed chunks still concatenate to the packed stream: a reader knows where in its line the chunk's first byte lies (it looks back in
// the mapped file / inflated image for the line's start), leaves a first byte at an odd position to the chunk before, and pairs a last
// unpaired base with the first byte of the chunk after (two bytes of look-ahead tell a base from a line end).
struct PackCtx {
    size_t line_pos0 = 0;      // index within its line of the chunk's first byte (used when that byte lies in a sequence line)
    size_t after = 0;          // bytes readable beyond the chunk's end (0: the chunk ends the file)
    bool prev_is_cr = false;   // the byte before the chunk is '\r' (a chunk that starts with the '\n' of a "\r\n": the '\r' is not a base of the line)
    bool prev_line_blank = false;   // the line before the one the chunk starts in is empty (blank lines after a file's last record that straddle a chunk start)
    // from the raw text around a chunk that starts at file_base + off
    void look_back(const uint8_t* file_base, uint64_t off) {    // line_pos0 is set
        prev_is_cr = off > 0 && file_base[off - 1] == '\r';
        prev_line_blank = false;
        const uint64_t ls = off - line_pos0;                       // start of the line the chunk begins in
        if (ls > 0 && file_base[ls - 1] == '\n') {
            uint64_t e = ls - 1;                                    // the '\n' that ends the line before
            if (e > 0 && file_base[e - 1] == '\r') --e;
            prev_line_blank = e == 0 || file_base[e - 1] == '\n';
        }
    }
    bool unpackable = false;   // out: two adjacent non-IUPAC characters would pack to '\n'; the run falls back to the plain two-line form
};
// keeps the bytes of the lines in phase 0 and 1 (header, sequence) of a chunk whose first byte lies in a line of phase ph0; in
// place when out == buf (the write position never passes the read position), or straight from a mapping of the file.  nl_kept / nl_all: newlines kept / seen; bad: a line that starts
// inside the chunk in phase 0 / 2 does not start with '@' / '+'.
// pk != nullptr: sequence lines packed (never in place: out and buf must not overlap; buf[n .. n + pk->after) must be readable).
static size_t compact_two_line(uint8_t* out, const uint8_t* buf, size_t n, int ph0, size_t& nl_kept, size_t& nl_all, bool& bad, TwoLineSummary& S,
                               PackCtx* pk = nullptr) {
    size_t d = 0, p = 0;
    // the header line of the record in hand was empty (blank lines after the last record stay blank lines); for a chunk that starts with a
    // sequence-phase line the reader has looked at the line before
    bool hdr_blank = pk && (ph0 & 3) == 1 && pk->prev_line_blank;
    int hdr_state = pk && (ph0 & 3) == 1 ? (pk->prev_line_blank ? 1 : 0) : -1;   // the same for the '+' line's check, in every mode: -1 = began before the chunk, not known
    int ph = ph0 & 3;
    bool line_start = false;  // the first line may be the tail of one that began in the previous chunk
    nl_kept = nl_all = 0; bad = false;
    S = TwoLineSummary();
    int64_t seq_len = -1;     // the sequence line of the record in hand, if it began in this chunk after the first line
    while (p < n) {
        if (line_start && ph == 0) hdr_state = (buf[p] == '\n' || (buf[p] == '\r' && p + 1 < n && buf[p + 1] == '\n')) ? 1 : 0;
        if (line_start && ((ph == 0 && buf[p] != '@' && buf[p] != '\n' && buf[p] != '\r') ||
                           (ph == 2 && buf[p] != '+' && !(hdr_state != 0 && (buf[p] == '\n' || buf[p] == '\r'))))) bad = true;   // a blank '+' line: only below a blank header (blank lines after the last record)
        const uint8_t* q = (const uint8_t*)memchr(buf + p, '\n', n - p);
        const size_t e = q ? (size_t)(q - buf) + 1 : n;
        if (q) {
            const size_t raw = (size_t)(q - buf) - p;   // without the '\n'
            if (nl_all == 0) { S.head_raw = raw; S.head_last = raw ? buf[p + raw - 1] : 0; S.last3[2] = -2; }
            else {
                const int64_t len = (int64_t)raw - (raw && buf[p + raw - 1] == '\r' ? 1 : 0);
                if (nl_all == 1) S.len1 = len;
                if (nl_all == 2) S.len2 = len;
                if (ph == 1) seq_len = len;
                if (ph == 3) {
                    if (seq_len >= 0) { if (seq_len != len) bad = true; }   // both lines of the pair inside the chunk
                    if (nl_all >= 3 || seq_len >= 0) S.pend_cleared = true;
                    seq_len = -1;
                }
                S.last3[0] = S.last3[1]; S.last3[1] = S.last3[2]; S.last3[2] = len;
            }
        } else { S.tail_raw = n - p; S.tail_last = buf[n - 1]; if (nl_all == 0) { S.head_raw = n - p; S.head_last = buf[n - 1]; } }
        if (ph == 1 && pk) {
            // the line's bases inside the chunk: [p, se); a '\r' belongs to the line end if a '\n' (or the end of the file) follows it
            size_t se = q ? (size_t)(q - buf) : n;
            if (se > p && buf[se - 1] == '\r' && (q || pk->after == 0 || buf[n] == '\n')) --se;
            const size_t i0 = nl_all == 0 ? pk->line_pos0 : 0;    // where in its line the segment starts
            size_t b = p;
            if ((i0 & 1u) && se > b) ++b;                          // an odd first base went into the last pair of the chunk before
            uint8_t next = 15;                                     // pairs with a last unpaired base: nothing, unless the line goes on in the next chunk
            if (!q && ((se - b) & 1u) && se == n && pk->after > 0) {
                const bool eol = buf[n] == '\n' || (buf[n] == '\r' && (pk->after == 1 || buf[n + 1] == '\n'));
                if (!eol) next = base_code_table()[buf[n]];
            }
            const size_t cr_split = (nl_all == 0 && i0 > 0 && q == buf + p && pk->prev_is_cr) ? 1 : 0;   // "\r" | "\n" split over two chunks
            if (i0 - cr_split == 0 && se == p && q && hdr_blank) {  // blank line after a blank header line: not a record, stays as it is ("\n" or "\r\n")
                if (cr_split) out[d++] = '\r';                      // (its '\r' ended the chunk before, which left it to this one)
                memcpy(out + d, buf + p, e - p); d += e - p; ++nl_kept;
            } else {
                d += pack_bases(out + d, buf + b, buf + se, next, pk->unpackable);
                if (q || pk->after == 0) {                          // the line ends here (its '\n', or the end of a file without one): parity terminator
                    const size_t cr_before = (nl_all == 0 && i0 > 0 && q == buf + p && pk->prev_is_cr) ? 1 : 0;   // "\r" | "\n" split over two chunks
                    out[d++] = ((i0 - cr_before + (se - p)) & 1u) ? 'O' : 'E';
                    if (q) { out[d++] = '\n'; ++nl_kept; }
                }
            }
        } else if (ph < 2) {
            if (ph == 0 && pk) {   // a header line without a character — counting what the chunk before holds of it (at most the '\r' of its "\r\n")
                const size_t raw = q ? (size_t)(q - buf) - p : 1;
                const size_t len = raw - ((raw && buf[p + raw - 1] == '\r') ? 1 : 0);
                const size_t before = nl_all == 0 ? pk->line_pos0 - ((pk->line_pos0 == 1 && raw == 0 && pk->prev_is_cr) ? 1 : 0) : 0;
                hdr_blank = q && len == 0 && before == 0;
            }
            if (out + d != buf + p) memmove(out + d, buf + p, e - p);
            d += e - p;
            if (q) ++nl_kept;
        }
        if (q) { ++nl_all; ph = (ph + 1) & 3; line_start = true; }
        p = e;
    }
    S.pend = seq_len;
    return d;
}
// The sequencer's half of the two-line mode's record checks: the chunks' summaries in stream order (their first byte in phase ph0 of its file).
void BlockFeeder::stitch(const Slot& sl, int ph0) {
    const TwoLineSummary& S = sl.sum;
    auto fail = [&](const char* what) {
        throw BarbellError(BB_E_FASTQ, "Input FASTQ parsing failed: '" + paths[sl.file] + "' " + what);
    };
    if (sl.raw_nl == 0) {   // no line ends here: the chunk continues the line in progress
        if (sl.raw_len) { st_part += S.head_raw; st_part_last = S.head_last; }
    } else {
        const size_t raw0 = st_part + S.head_raw;
        const uint8_t last0 = S.head_raw ? S.head_last : st_part_last;
        const int64_t L0 = (int64_t)raw0 - (raw0 && last0 == '\r' ? 1 : 0);
        if (ph0 == 1) st_pend = L0;
        if (ph0 == 3) { if (st_pend >= 0 && st_pend != L0) fail("holds a record whose quality line is not as long as its sequence"); st_pend = -1; }
        if (ph0 == 2 && sl.raw_nl >= 2) { if (st_pend >= 0 && st_pend != S.len1) fail("holds a record whose quality line is not as long as its sequence"); st_pend = -1; }
        if (ph0 == 1 && sl.raw_nl >= 3) { if (L0 != S.len2) fail("holds a record whose quality line is not as long as its sequence"); st_pend = -1; }
        if (S.pend_cleared) st_pend = -1;
        if (S.pend >= 0) st_pend = S.pend;
        // the last two lines that have ended, for the check at the end of the file
        const int64_t a = S.last3[0] == -2 ? L0 : S.last3[0], b = S.last3[1] == -2 ? L0 : S.last3[1], c = S.last3[2] == -2 ? L0 : S.last3[2];
        if (sl.raw_nl >= 3) { st_last3[0] = a; st_last3[1] = b; st_last3[2] = c; }
        else if (sl.raw_nl == 2) { st_last3[0] = st_last3[2]; st_last3[1] = b; st_last3[2] = c; }
        else { st_last3[0] = st_last3[1]; st_last3[1] = st_last3[2]; st_last3[2] = c; }
        st_part = S.tail_raw; st_part_last = S.tail_last;
    }
    if (sl.last) {  // the file's end: a last line without '\n' counts; blank lines may follow the last record (the GPU parser ignores them: whole blank records, then the surplus lines)
        uint64_t lines = seq_raw_lines + sl.raw_nl;
        int64_t tail_len = -1;
        if (st_part) { tail_len = (int64_t)st_part - (st_part_last == '\r' ? 1 : 0); ++lines; }
        const int r = (int)(lines & 3u);
        if (r == 0) {
            if (tail_len >= 0 && st_pend >= 0 && st_pend != tail_len) fail("holds a record whose quality line is not as long as its sequence");
        } else {
            // r surplus lines: they must all be blank
            const int64_t l1 = tail_len >= 0 ? tail_len : st_last3[2], l2 = tail_len >= 0 ? st_last3[2] : st_last3[1], l3 = tail_len >= 0 ? st_last3[1] : st_last3[0];
            const bool blank = l1 == 0 && (r < 2 || l2 == 0) && (r < 3 || l3 == 0);
            if (!blank) fail("ends inside a record (truncated file?)");
        }
        st_part = 0; st_part_last = 0; st_pend = -1; st_last3[0] = st_last3[1] = st_last3[2] = -1;
    }
}
// phase of a chunk's first byte, read off the text: the first line that starts with '@' and has a line starting with '+' two
// lines below is a header (phase 0); -1 if no such pair is found among the chunk's first lines
static int guess_phase(const uint8_t* buf, size_t n, bool at_file_start) {
    if (at_file_start) return 0;
    size_t st[16], ns = 0, p = 0;
    while (ns < 16 && p < n) {
        const void* q = memchr(buf + p, '\n', n - p);
        if (!q) break;
        p = (size_t)((const uint8_t*)q - buf) + 1;
        if (p < n) st[ns++] = p;
    }
    for (size_t j = 0; j + 2 < ns; ++j)
        if (buf[st[j]] == '@' && buf[st[j + 2]] == '+') return (int)((4 - ((j + 1) & 3)) & 3);  // line j+1 of the chunk is in phase 0
    return -1;
}
static size_t count_nl(const uint8_t* p, size_t n) {
    size_t c = 0;
    const uint8_t* e = p + n;
    while (p < e) {
        const void* q = memchr(p, '\n', (size_t)(e - p));
        if (!q) break;
        ++c;
        p = (const uint8_t*)q + 1;
    }
    return c;
}
static bool sniff_gzip(const std::string& path) {  // magic bytes, not the file name (the reference's reader sniffs too)
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) throw BarbellError(BB_E_INVALID, "Failed to open FASTQ input: " + path);
    unsigned char m[2] = {0, 0};
    const size_t n = fread(m, 1, 2, f);
    fclose(f);
    return n == 2 && m[0] == 0x1f && m[1] == 0x8b;
}

// First record start at or after byte `pos` of a plain FASTQ file (--shard-by bytes): the phase of the line `pos` lies in is read off the text
// as the readers do (guess_phase: a line that starts with '@' two lines above one that starts with '+'), then as many line ends are skipped
// as it takes to stand at the start of a header line.  Shard R ends where shard R + 1 begins: both call this with the same `pos`.
uint64_t BlockFeeder::record_start(int fd, uint64_t size, uint64_t pos, const std::string& path) {
    if (pos == 0) return 0;
    if (pos >= size) return size;
    const size_t want = (size_t)std::min<uint64_t>(size - (pos - 1), (32u << 20) + 1);   // from the byte before `pos` on
    std::vector<uint8_t> buf(want);
    size_t got = 0;
    while (got < want) {
        const ssize_t r = pread(fd, buf.data() + got, want - got, (off_t)(pos - 1 + got));
        if (r < 0) { if (errno == EINTR) continue; throw BarbellError(BB_E_INVALID, "Error reading FASTQ file '" + path + "'"); }
        if (r == 0) break;
        got += (size_t)r;
    }
    if (got < 2) return size;
    const bool at_line_start = buf[0] == '\n';
    const uint8_t* w = buf.data() + 1;
    const size_t n = got - 1;
    const int ph = guess_phase(w, n, false);
    if (ph < 0) {
        if (pos - 1 + got >= size) return size;   // fewer than three line ends from here to the end of the file: the last record started earlier
        throw BarbellError(BB_E_FASTQ, "--shard-by bytes: no record boundary found in '" + path + "' within 32 MiB of byte " + std::to_string(pos));
    }
    if (at_line_start && ph == 0) return pos;
    size_t skip = (size_t)((4 - ph) & 3);
    if (skip == 0) skip = 4;          // inside a header line: the next record
    size_t p = 0;
    for (size_t k = 0; k < skip; ++k) {
        const void* q = memchr(w + p, '\n', n - p);
        if (!q) return size;          // the file ends first
        p = (size_t)((const uint8_t*)q - w) + 1;
    }
    return pos + p;
}

BlockFeeder::BlockFeeder(int device_, const std::vector<std::string>& files, size_t chunk_bytes, unsigned n_slots, unsigned n_readers, unsigned n_inflate,
                         bool two_line_mode, bool pack_mode, uint32_t byte_shard_rank, uint32_t byte_shard_world)
    : device(device_), paths(files), chunk(chunk_bytes), two_line(two_line_mode), pack(two_line_mode && pack_mode), lpr(two_line_mode ? 2 : 4) {
    shard_rank = byte_shard_rank; shard_world = std::max(1u, byte_shard_world);
    begins.assign(paths.size(), 0); ends.assign(paths.size(), 0);
    if (const char* e = getenv("BARBELL_AMD_HEAD_BYTES")) HEAD = (size_t)std::max(16L, atol(e));
    is_gz.resize(paths.size()); fds.assign(paths.size(), -1); sizes.assign(paths.size(), 0); size_known.assign(paths.size(), 0);
    maps.assign(paths.size(), nullptr);
    std::vector<std::string> gz_paths;
    for (size_t i = 0; i < paths.size(); ++i) {
        // Pipes, process substitutions and /dev/stdin have no size and cannot be read at offsets (and a sniff would eat their
        // first bytes): they are read sequentially, whole, through zlib like a gzip file — gzread passes plain text through
        // and inflates gzip, whichever arrives (the reference's paraseq reader streams both as well, io.rs:29-33).
        struct stat pst;
        if (stat(paths[i].c_str(), &pst) != 0) throw BarbellError(BB_E_INVALID, "Failed to open FASTQ input: " + paths[i]);
        if (!S_ISREG(pst.st_mode)) { is_gz[i] = 1; }
        else is_gz[i] = sniff_gzip(paths[i]) ? 1 : 0;
        if (is_gz[i]) {
            if (shard_world > 1) throw BarbellError(BB_E_INVALID, "--shard-by bytes: '" + paths[i] + "' is gzip (or a pipe): only plain files can be cut by byte ranges; shard those by file");
            continue;
        }
        fds[i] = open(paths[i].c_str(), O_RDONLY);
        struct stat st;
        if (fds[i] < 0 || fstat(fds[i], &st) != 0) throw BarbellError(BB_E_INVALID, "Failed to open FASTQ input: " + paths[i]);
        sizes[i] = (uint64_t)st.st_size; size_known[i] = 1;
        ends[i] = sizes[i];
        if (shard_world > 1) {   // this process's byte range of the file, widened to record starts
            begins[i] = record_start(fds[i], sizes[i], sizes[i] / shard_world * shard_rank, paths[i]);
            ends[i] = shard_rank + 1 == shard_world ? sizes[i] : record_start(fds[i], sizes[i], sizes[i] / shard_world * (shard_rank + 1), paths[i]);
            if (ends[i] < begins[i]) ends[i] = begins[i];   // (not for FASTQ text: record_start is monotone there)
        }
        if (two_line && st.st_size > 0 && !getenv("BARBELL_AMD_NO_MMAP")) {  // the readers compact straight out of the page cache: one pass over the text, no copy of the dropped half
            void* m = mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, fds[i], 0);
            if (m != MAP_FAILED) {
                maps[i] = (const uint8_t*)m;
                (void)madvise(m, (size_t)st.st_size, MADV_SEQUENTIAL);
            }
        }
    }
    for (size_t i = 0; i < paths.size(); ++i)
        if (!is_gz[i] && !maps[i] && sizes[i] > 0) pack = false;   // a plain file that could not be mapped is read with pread: no look-back, no packing
    bool any_gz = false;
    for (char g : is_gz) any_gz = any_gz || g;
    if (any_gz) inflater = std::make_unique<ParallelInflater>(paths, is_gz, n_inflate);
    if (!any_gz) {  // every size is known: no slot needs to be larger than the largest file (small inputs do not page-lock gigabytes)
        uint64_t mx = 4096;
        for (uint64_t z : sizes) mx = std::max(mx, z);
        chunk = (size_t)std::min<uint64_t>(chunk, (mx + 4095) & ~(uint64_t)4095);
        HEAD = std::min(HEAD, (chunk + 15) & ~(size_t)15);
    }
    // Page-locking is the expensive part of starting up (6 GB took a second, and as long again to undo): a slot is allocated by the reader
    // that fills it first — in parallel, while the contexts are being created and the first blocks are already on the GPU —, and a short
    // input never touches most of them.
    slots.resize(std::max(3u, n_slots));
    for (size_t i = 0; i < slots.size(); ++i) { slots[i].p = nullptr; slots[i].cap = HEAD + chunk; slots[i].seq = i; }
    for (unsigned i = 0; i < std::max(1u, n_readers); ++i) readers.emplace_back([this]() { reader_loop(); });
}
BlockFeeder::~BlockFeeder() {
    { std::lock_guard<std::mutex> lk(mu); stop = true; }
    cv.notify_all();
    if (inflater) inflater->cancel();   // a reader may be waiting for a piece of inflated text
    for (auto& t : readers) if (t.joinable()) t.join();
    inflater.reset();
    if (!keep_slots)
        for (auto& sl : slots) if (sl.p) { if (pageable) free(sl.p); else bb_host_free_on(device, sl.p); }
    if (!keep_slots) {
        for (size_t i = 0; i < maps.size(); ++i) if (maps[i]) munmap((void*)maps[i], (size_t)sizes[i]);
    } else {
        // The process is about to exit and would take the mappings down by itself — on ONE thread: 128 GB of FASTQ are 32 M page-table
        // entries, 1.0 s of a 2.6 s run on 16 M reads.  Thirty-two threads drop them side by side in a few tens of milliseconds (the page cache
        // keeps the data).
        std::vector<std::thread> zap;
        const uintptr_t pg = (uintptr_t)sysconf(_SC_PAGESIZE);
        uint64_t total = 0;
        for (size_t i = 0; i < maps.size(); ++i) if (maps[i]) total += sizes[i];
        if (total >= (256u << 20)) {
            // ranges of at least 64 MiB, at most ~32 of them per run of threads, whatever the number of files
            const uint64_t part = std::max<uint64_t>(64u << 20, ((total / 32) + pg - 1) & ~(uint64_t)(pg - 1));
            std::vector<std::pair<const uint8_t*, uint64_t>> ranges;
            for (size_t i = 0; i < maps.size(); ++i) {
                if (!maps[i]) continue;
                for (uint64_t a0 = 0; a0 < sizes[i]; a0 += part) ranges.emplace_back(maps[i] + a0, std::min<uint64_t>(part, sizes[i] - a0));
            }
            std::atomic<size_t> next_range{0};
            const unsigned nt = (unsigned)std::min<size_t>(32, ranges.size());
            for (unsigned k = 0; k < nt; ++k)
                zap.emplace_back([&ranges, &next_range]() {
                    for (size_t r; (r = next_range.fetch_add(1)) < ranges.size();) (void)madvise((void*)ranges[r].first, (size_t)ranges[r].second, MADV_DONTNEED);
                });
            for (auto& t : zap) t.join();
            zap.clear();
        }
        for (auto& t : zap) t.join();
    }
    for (int fd : fds) if (fd >= 0) close(fd);
}
// next chunk of the stream; gzip files come as record-aligned pieces of inflated text (ParallelInflater), each chunked like a small file
bool BlockFeeder::claim(Task& t) {
    for (;;) {
        size_t f;
        {
            std::unique_lock<std::mutex> lk(mu);
            if (stop) return false;
            if (cur_file >= paths.size()) { if (!claims_done) { claims_done = true; cv.notify_all(); } return false; }
            f = cur_file;
            if (!is_gz[f]) {
                if (cur_off < begins[f]) cur_off = begins[f];
                if (ends[f] == begins[f]) {  // empty file (or an empty byte range of one): an empty last chunk keeps the sequence simple
                    t = Task{f, begins[f], 0, next_seq++, true, nullptr};
                    ++cur_file; cur_off = 0;
                    return true;
                }
                const size_t len = (size_t)std::min<uint64_t>(chunk, ends[f] - cur_off);
                t = Task{f, cur_off, len, next_seq++, cur_off + len == ends[f], nullptr};
                cur_off += len;
                if (t.last) { ++cur_file; cur_off = 0; }
                return true;
            }
            if (cur_piece) {
                const size_t len = (size_t)std::min<uint64_t>(chunk, cur_piece->size - cur_off);
                t = Task{f, cur_off, len, next_seq++, cur_off + len == cur_piece->size, cur_piece};
                cur_off += len;
                if (t.last) { cur_piece.reset(); cur_off = 0; }
                return true;
            }
            if (piece_fetching) { cv.wait(lk, [&]() { return stop || !piece_fetching; }); continue; }
            piece_fetching = true;
        }
        std::shared_ptr<GzPiece> pc;
        std::string e;
        try { pc = inflater->next_piece(f); } catch (const std::exception& ex) { e = ex.what(); }   // blocks until inflated
        {
            std::lock_guard<std::mutex> lk(mu);
            piece_fetching = false;
            if (e.empty()) {
                if (!pc) { ++cur_file; cur_off = 0; }
                else { cur_piece = pc; cur_off = 0; pc->chunks_left = std::max<uint64_t>(1, (pc->size + chunk - 1) / chunk); }
            }
        }
        cv.notify_all();
        if (!e.empty()) throw BarbellError(BB_E_INVALID, e);
    }
}
void BlockFeeder::reader_loop() {
    try {
        Task t;
        while (claim(t)) {
            Slot& sl = slots[t.seq % slots.size()];
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&]() { return stop || (sl.state == 0 && sl.seq == t.seq); });
                if (stop) return;
                sl.state = 1;
            }
            if (!sl.p) {  // first use of this slot (it is this reader's alone until it is marked full)
                void* q = nullptr;
                if (pageable) {
                    if (posix_memalign(&q, 2u << 20, sl.cap) != 0) throw BarbellError(BB_E_NOMEM, "out of memory (block buffer)");
                    (void)madvise(q, sl.cap, MADV_HUGEPAGE);
                } else if (bb_host_malloc_on(device, sl.cap, &q) != BB_OK) throw BarbellError(BB_E_NOMEM, "bb_host_malloc_on failed (page-locked block buffer)");
                sl.p = (uint8_t*)q;
            }
            uint8_t* dst = sl.p + HEAD;
            const uint8_t* src = nullptr;   // the chunk's raw bytes where they can be read in place (inflated image, mapped file)
            // the file the chunk lies in, for the look-ahead / look-back below: a plain file's range, or the piece of inflated text
            const uint64_t f_begin = t.piece ? 0 : begins[t.file], f_end = t.piece ? t.piece->size : ends[t.file], f_size = t.piece ? t.piece->size : sizes[t.file];
            if (t.piece) src = t.piece->data.data() + t.off;
            else if (maps[t.file]) {
                // a mapped file that has been truncated since it was opened would fault (SIGBUS) when its lost pages are touched: look at its
                // size again before every chunk and fail like the pread path does (a file cut while a chunk is being read is still a race)
                struct stat stn;
                if (fstat(fds[t.file], &stn) != 0 || (uint64_t)stn.st_size < t.off + t.len)
                    throw BarbellError(BB_E_INVALID, "FASTQ file '" + paths[t.file] + "' shrank while it was read");
                src = maps[t.file] + t.off;
            }
            static const bool map_populate = getenv("BARBELL_AMD_MAP_POPULATE") != nullptr;
            if (map_populate && src && maps[t.file] && t.len) {   // experiment: the chunk's pages mapped by one call instead of one fault per 64 KB
                const uintptr_t pg = (uintptr_t)sysconf(_SC_PAGESIZE);
                const uintptr_t a0 = (uintptr_t)src & ~(pg - 1), a1 = ((uintptr_t)src + t.len + pg - 1) & ~(pg - 1);
                (void)madvise((void*)a0, (size_t)(a1 - a0), 22 /* MADV_POPULATE_READ */);
            }
            size_t got_len = t.len, nl = 0, raw_nl = 0;
            int ph0 = 0;
            bool bad = false;
            TwoLineSummary sum;
            bool unpackable = false;
            if (two_line && src) {
                // the phase is read off the first lines from the chunk's start; they may lie beyond its end (a chunk shorter than three lines):
                // the mapped file / inflated image can be read ahead
                ph0 = guess_phase(src, t.len + (size_t)std::min<uint64_t>(f_size - (t.off + t.len), 1u << 20), t.off == f_begin);   // (a shard's range, and a piece, begin at a record start)
                if (ph0 < 0 && t.off > f_begin) {   // too few lines from here to the end of the file: read the phase off the text BEFORE the chunk and count on
                    const uint64_t back = std::min<uint64_t>(t.off - f_begin, 4u << 20);
                    const uint8_t* w = src - back;
                    const int pw = guess_phase(w, (size_t)(f_size - (t.off - back)), t.off - back == f_begin);
                    if (pw >= 0) ph0 = (int)((pw + count_nl(w, (size_t)back)) & 3u);
                }
                if (ph0 >= 0) {
                    PackCtx pk;
                    if (pack) { pk.line_pos0 = line_pos(src - t.off, t.off); pk.after = (size_t)(f_end - (t.off + t.len)); pk.look_back(src - t.off, t.off); }
                    got_len = compact_two_line(dst, src, t.len, ph0, nl, raw_nl, bad, sum, pack ? &pk : nullptr);
                    unpackable = pk.unpackable;
                } else { if (t.len) memcpy(dst, src, t.len); raw_nl = count_nl(dst, t.len); }   // left raw: the sequencer compacts it with the true phase
            } else {
                if (src) { if (t.len) memcpy(dst, src, t.len); }
                else {
                    size_t got = 0;
                    while (got < t.len) {
                        const ssize_t r = pread(fds[t.file], dst + got, t.len - got, (off_t)(t.off + got));
                        if (r < 0) { if (errno == EINTR) continue; throw BarbellError(BB_E_INVALID, "Error reading FASTQ file '" + paths[t.file] + "'"); }
                        if (r == 0) throw BarbellError(BB_E_INVALID, "FASTQ file '" + paths[t.file] + "' shrank while it was read");
                        got += (size_t)r;
                    }
                }
                if (two_line) {
                    ph0 = guess_phase(dst, t.len, t.off == f_begin);
                    if (ph0 >= 0) got_len = compact_two_line(dst, dst, t.len, ph0, nl, raw_nl, bad, sum);
                    else raw_nl = count_nl(dst, t.len);
                } else nl = count_nl(dst, t.len);
            }
            static const bool map_drop = getenv("BARBELL_AMD_MAP_DROP") != nullptr;
            if (src && maps[t.file] && t.len && map_drop) {
                // BARBELL_AMD_MAP_DROP=1: this chunk's pages of the mapping dropped as soon as it is staged (a process that must not hold page-table
                // entries for the whole input).  Measured on 8 M reads: steady state 12.5 -> 11.3 M reads/s (the shoot-downs disturb the upload
                // threads); by default the mapping is taken down at the end instead, by all readers at once (~BlockFeeder)
                const uintptr_t pg = (uintptr_t)sysconf(_SC_PAGESIZE);
                const uintptr_t a0 = ((uintptr_t)src + pg - 1) & ~(pg - 1), a1 = ((uintptr_t)src + t.len) & ~(pg - 1);
                if (a1 > a0) (void)madvise((void*)a0, (size_t)(a1 - a0), MADV_DONTNEED);
            }
            if (t.piece) {
                bool last_copy;
                { std::lock_guard<std::mutex> lk(mu); last_copy = --t.piece->chunks_left == 0; }
                if (last_copy) inflater->piece_consumed(t.file, t.piece->size);  // every chunk of the piece has been copied out
                t.piece.reset();
            }
            {
                std::lock_guard<std::mutex> lk(mu);
                sl.got = got_len; sl.nl = nl; sl.last = t.last; sl.file = t.file; sl.raw_nl = raw_nl; sl.phase0 = ph0; sl.off = t.off; sl.raw_len = t.len;
                sl.bad = bad; sl.sum = sum; sl.unpackable = unpackable; sl.state = 2;
            }
            cv.notify_all();
        }
    } catch (const std::exception& e) { fail(e.what()); }
}
void BlockFeeder::unref(int i) {
    bool freed = false;
    {
        std::lock_guard<std::mutex> lk(mu);
        Slot& sl = slots[(size_t)i];
        if (--sl.refs == 0) { sl.state = 0; sl.seq += slots.size(); freed = true; }
    }
    if (freed) cv.notify_all();
}
void BlockFeeder::release(int slot) { if (slot >= 0) unref(slot); }

bool BlockFeeder::next(Block& b) {
    for (;;) {
        if (done) return false;
        Slot* sl;
        {
            std::unique_lock<std::mutex> lk(mu);
            sl = &slots[want_seq % slots.size()];
            cv.wait(lk, [&]() { return !err.empty() || (sl->state == 2 && sl->seq == want_seq) || (claims_done && want_seq >= next_seq); });
            if (!err.empty()) throw BarbellError(BB_E_INVALID, err);
            if (!(sl->state == 2 && sl->seq == want_seq)) { done = true; return false; }  // every chunk has been sequenced
            sl->refs = 2;  // the worker that uploads the block + the sequencer (its tail is the next block's carry)
        }
        const int si = (int)(want_seq % slots.size());
        ++want_seq;
        uint8_t* body = sl->p + HEAD;
        if (two_line) {   // the reader's guess of the chunk's first phase against the running line count of the file
            if (sl->file != seq_file) { seq_file = sl->file; seq_raw_lines = 0; }
            const int truth = (int)(seq_raw_lines & 3u);
            if (sl->phase0 != truth && pack) {
                // packed staging: the chunk is redone from the mapped file under the true phase (look-back and look-ahead need the file); a
                // gzip image may be gone by now: the run then falls back to the plain two-line form
                if (is_gz[sl->file] || !maps[sl->file]) sl->unpackable = true;
                else {
                    PackCtx pk;
                    pk.line_pos0 = line_pos(maps[sl->file], sl->off); pk.after = (size_t)(ends[sl->file] - (sl->off + sl->raw_len));
                    pk.look_back(maps[sl->file], sl->off);
                    sl->got = compact_two_line(body, maps[sl->file] + sl->off, sl->raw_len, truth, sl->nl, sl->raw_nl, sl->bad, sl->sum, &pk);
                    sl->unpackable = pk.unpackable;
                }
            } else if (sl->phase0 != truth) {
                if (sl->phase0 >= 0) {  // compacted under a wrong phase: the raw bytes are needed again
                    if (is_gz[sl->file]) throw BarbellError(BB_E_FASTQ, "'" + paths[sl->file] + "': line layout not recognised while dropping quality lines; rerun with --no-compact");
                    if (maps[sl->file]) memcpy(body, maps[sl->file] + sl->off, sl->raw_len);
                    else {
                        size_t got = 0;
                        while (got < sl->raw_len) {
                            const ssize_t r = pread(fds[sl->file], body + got, sl->raw_len - got, (off_t)(sl->off + got));
                            if (r <= 0) { if (r < 0 && errno == EINTR) continue; throw BarbellError(BB_E_INVALID, "Error reading FASTQ file '" + paths[sl->file] + "'"); }
                            got += (size_t)r;
                        }
                    }
                }
                sl->got = compact_two_line(body, body, sl->raw_len, truth, sl->nl, sl->raw_nl, sl->bad, sl->sum);
            }
            if (sl->unpackable) {
                if (getenv("BARBELL_AMD_PROFILE")) fprintf(stderr, "profile: chunk at %llu of '%s' (%zu bytes, phase guessed %d, true %d) has no packed form\n",
                                                           (unsigned long long)sl->off, paths[sl->file].c_str(), sl->raw_len, sl->phase0, truth);
                throw PackFallback();
            }
            if (sl->bad) throw BarbellError(BB_E_FASTQ, "Input FASTQ parsing failed: '" + paths[sl->file] + "' holds a record that is not a 4-line FASTQ record");
            stitch(*sl, truth);
            seq_raw_lines += sl->raw_nl;
        }
        size_t cut = sl->got;  // bytes of this chunk that go into this block
        size_t lines_left = 0;
        if (!sl->last) {
            const size_t total = carry_lines + sl->nl;
            const size_t r = total % lpr;          // complete lines after the last complete record
            if (total < lpr || sl->nl <= r) {       // no record ends inside this chunk (a record longer than the block, or a tiny --block-bytes):
                // the whole chunk joins the carry, kept aside, and the next chunk continues the record
                std::vector<uint8_t> nb(carry_len + sl->got);
                if (carry_len) memcpy(nb.data(), carry_ptr, carry_len);
                if (sl->got) memcpy(nb.data() + carry_len, body, sl->got);
                carry_buf.swap(nb);
                if (carry_slot >= 0) unref(carry_slot);
                carry_slot = -1;
                carry_ptr = carry_buf.data(); carry_len = carry_buf.size(); carry_lines = total;
                unref(si); unref(si);  // neither a worker nor the sequencer keeps the slot
                continue;
            }
            // the cut is just after line end number (nl - r) of the chunk: walk back over the partial last line and r lines
            const uint8_t* e = body + sl->got;
            for (size_t k = 0; k <= r; ++k) {
                const void* q = memrchr(body, '\n', (size_t)(e - body));
                e = (const uint8_t*)q;  // not null: nl > r
            }
            cut = (size_t)(e - body) + 1;
            lines_left = r;
        }
        // assemble: carry (in the previous slot's tail, or aside) + chunk[0, cut)
        Block out;
        if (carry_len > HEAD) {  // a carry longer than the headroom (a huge record): assemble aside
            auto big = std::make_shared<std::vector<uint8_t>>(carry_len + cut);
            memcpy(big->data(), carry_ptr, carry_len);
            memcpy(big->data() + carry_len, body, cut);
            out.data = big->data(); out.len = big->size(); out.big = big; out.slot = -1;
            unref(si);  // the worker does not need the slot
        } else {
            if (carry_len) memcpy(body - carry_len, carry_ptr, carry_len);
            out.data = body - carry_len; out.len = carry_len + cut; out.slot = si;
        }
        if (carry_slot >= 0) unref(carry_slot);  // the previous slot's tail has been copied
        // the new carry
        carry_ptr = body + cut; carry_len = sl->got - cut; carry_lines = lines_left;
        if (carry_len) carry_slot = si;
        else { carry_slot = -1; unref(si); }
        if (sl->last) { carry_lines = 0; }
        if (out.len == 0) { if (out.slot >= 0) unref(out.slot); continue; }  // an empty file
        out.index = n_blocks++;
        b = out;
        return true;
    }
}
}  // namespace

// What one block of the stream turns into; produced by the worker that owns the block's context, committed to the
// output files by the main thread in block order.
namespace {
struct BlockResult {
    size_t n_reads = 0, found = 0, rows = 0, kept = 0, dropped = 0, trimmed = 0, split = 0, trim_failed = 0;
    std::vector<uint8_t> anno, kept_tsv, drop_tsv;              // TSV lines rendered on the GPU
    std::string ppr;                                            // the block's lines of pattern_per_read.tsv, rendered by the worker
    std::vector<std::pair<std::string, size_t>> patterns;       // (pattern, reads of the block that show it), first-appearance order
    std::string failed_ids;                                     // one id per line
    std::shared_ptr<void> text;                                 // holds the page-locked buffer of the rendered records
    const uint8_t* text_ptr = nullptr;
    struct Span { std::string label; size_t off, n; uint64_t first = 0; uint32_t n_records = 0; };
    std::vector<Span> spans;
    std::shared_ptr<const LabelWriters::Cut> cut;              // host_cut: the plan + the block's text, records cut by the writer threads
    double t_ingest = 0, t_gpu = 0, t_rest = 0, t_filter = 0, t_inspect = 0, t_trim = 0;
};
}  // namespace

// bb_rccl.cpp: sums the per-context histograms.  Contexts on distinct devices are all-reduced with RCCL over xGMI
// (ncclAllReduce, uint64 sum, in place on bb_counts_dev); contexts that share a device are first summed on the host.
std::vector<uint64_t> allreduce_counts(const std::vector<Demuxer*>& dms, std::string& how);
std::vector<uint64_t> allreduce_counts_shards(Demuxer* lead, const std::vector<uint64_t>& local, uint32_t rank, uint32_t world,
                                              const std::string& base, std::string& how);

// CPUs this process can keep busy: the affinity mask cut by the cgroup's CPU quota.  A container may see every CPU of its host and still be
// throttled to a few (round 5's MI355X box: 256 visible, cpu.max = "1600000 100000" = 16): more runnable threads than that only buy
// throttling.  BARBELL_AMD_CPUS overrides.
unsigned effective_cpus() {
    if (const char* e = getenv("BARBELL_AMD_CPUS")) { const long v = atol(e); if (v > 0) return (unsigned)v; }
    unsigned n = std::max(1u, std::thread::hardware_concurrency());
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof(set), &set) == 0) n = std::max(1, CPU_COUNT(&set));
    if (FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r")) {            // cgroup v2: "<quota|max> <period>"
        char q[32] = ""; long per = 0;
        if (fscanf(f, "%31s %ld", q, &per) == 2 && strcmp(q, "max") != 0 && per > 0) n = std::min<unsigned>(n, (unsigned)std::max(1L, (atol(q) + per / 2) / per));
        fclose(f);
    } else if (FILE* g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) {   // cgroup v1
        long q = -1, per = 100000;
        if (fscanf(g, "%ld", &q) != 1) q = -1;
        fclose(g);
        if (FILE* h = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) { if (fscanf(h, "%ld", &per) != 1) per = 100000; fclose(h); }
        if (q > 0 && per > 0) n = std::min<unsigned>(n, (unsigned)std::max(1L, (q + per / 2) / per));
    }
    return std::max(1u, n);
}

// `barbell-amd stage` (tests, no GPU): the text the reader threads and the sequencer stage for upload — the blocks of whole records, one after
// the other — written to a file.  Returns the form that was staged: 4 (4-line text), 2 (two-line), 1 (two-line, sequence lines packed); a
// packed run that meets input without a packed form falls back to the two-line form like annotate() does.
int stage_blocks(const std::vector<std::string>& read_files, size_t block_bytes, unsigned n_threads, bool two_line, bool pack, const std::string& out_path,
                 size_t& n_blocks, uint32_t byte_shard_rank, uint32_t byte_shard_world) {
    for (int attempt = 0;; ++attempt) {
        FILE* f = fopen(out_path.c_str(), "wb");
        if (!f) throw BarbellError(BB_E_INVALID, "Failed to create '" + out_path + "'");
        n_blocks = 0;
        try {
            BlockFeeder feeder(-1, read_files, std::max<size_t>(block_bytes, 16), 8, std::max(1u, n_threads), std::max(1u, n_threads), two_line, pack,
                               byte_shard_rank, byte_shard_world);
            const bool packed = feeder.pack;
            BlockFeeder::Block b;
            while (feeder.next(b)) {
                if (b.len && fwrite(b.data, 1, b.len, f) != b.len) { fclose(f); throw BarbellError(BB_E_INVALID, "write failed"); }
                ++n_blocks;
                feeder.release(b.slot);
            }
            fclose(f);
            return packed ? 1 : (two_line ? 2 : 4);
        } catch (const PackFallback&) {
            fclose(f);
            if (attempt) throw BarbellError(BB_E_INVALID, "staging failed twice");
            pack = false;
        } catch (...) { fclose(f); throw; }
    }
}

static AnnotateStats annotate_once(const std::vector<std::string>& read_files, const std::string& out_file,
                                   std::vector<BarcodeGroup> query_groups, const AnnotateConfig& config);
AnnotateStats annotate(const std::vector<std::string>& read_files, const std::string& out_file,
                       std::vector<BarcodeGroup> query_groups, const AnnotateConfig& config) {
    try {
        return annotate_once(read_files, out_file, query_groups, config);
    } catch (const PackFallback&) {
        // a read with two adjacent characters that are not IUPAC letters (or a gzip chunk whose line layout could not be told) has no packed
        // form: the same run again with the sequence lines as text — every output file is created anew
        if (config.verbose || getenv("BARBELL_AMD_PROFILE")) fputs("note: input not representable in the packed upload form; staging the sequence lines as text\n", stderr);
        AnnotateConfig plain = config;
        plain.pack_upload = false;
        return annotate_once(read_files, out_file, std::move(query_groups), plain);
    }
}
static AnnotateStats annotate_once(const std::vector<std::string>& read_files, const std::string& out_file,
                                   std::vector<BarcodeGroup> query_groups, const AnnotateConfig& config) {
    if (read_files.empty()) throw BarbellError(BB_E_INVALID, "No FASTQ input files provided");  // io.rs:20-26
    const bool filtering = !config.filter_patterns.empty();
    const bool trimming = config.trim.has_value();
    if (trimming && !filtering) throw BarbellError(BB_E_INVALID, "the trim step needs filter patterns (cuts come from the filter)");
    // contexts: block i of the stream -> context i mod G
    std::vector<int> devs = config.devices;
    if (devs.empty()) devs.assign(std::max(1u, config.streams_per_device), config.device);
    const size_t G = devs.size();
    const bool prof0 = getenv("BARBELL_AMD_PROFILE") != nullptr;
    auto now0 = []() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t_enter = now0();
    // The readers start on the input right away (the feeder needs a device, not a context): by the time the contexts exist — created side
    // by side, one thread each — the first blocks are staged.  Nothing is written before every context is up: geometry / device errors
    // surface before any output file exists.
    const bool two_line = config.compact_upload && !trimming;  // nothing downstream of annotate / filter / inspect reads qualities
    const bool host_cut = trimming && config.host_cut && !getenv("BARBELL_AMD_GPU_RENDER");
    const size_t block = config.batch_reads ? std::max<size_t>(config.batch_reads * 4096, 4096) : std::max<size_t>(config.block_bytes, 4096);
    // two-line mode: a slot is about half full and a chunk costs its reader a pass over the text, so twice the slots and readers
    // host_cut: a slot also waits for the writer threads (at most 4 blocks there), and the last holder may be one of them
    const unsigned n_threads = config.n_threads ? config.n_threads : std::min(32u, std::max(4u, effective_cpus()));
    auto feeder_p = std::make_shared<BlockFeeder>(devs[0], read_files, block, (unsigned)((two_line ? 2 : 1) * (3 * G + 2) + (host_cut ? 6 : 0)),
                                                  std::min<unsigned>(std::min<unsigned>(std::max(1u, n_threads), 32u), std::max(4u, effective_cpus())), n_threads, two_line,
                                                  config.pack_upload && !getenv("BARBELL_AMD_NO_PACK"),
                                                  config.shard_by_bytes ? config.shard_rank : 0u, config.shard_by_bytes ? config.shard_world : 1u);
    const bool packed = feeder_p->pack;   // two bases per byte in the sequence lines (needs the raw text in memory: mapped or inflated)
    feeder_p->keep_slots = config.process_exits_after;
    const double t_feeder_up = now0();
    std::vector<std::unique_ptr<Demuxer>> dms(G);
    {
        std::vector<std::thread> makers;
        std::vector<std::exception_ptr> errs(G);
        for (size_t w = 0; w < G; ++w)
            makers.emplace_back([&, w]() {
                try {
                    dms[w] = std::make_unique<Demuxer>(config.alpha, config.verbose, config.min_score, config.min_score_diff, devs[w]);
                    for (const auto& g : query_groups) dms[w]->add_query_group(g);
                    dms[w]->ctx();
                    if (filtering) dms[w]->set_filter(config.filter_patterns);
                    if (trimming) dms[w]->set_trim(*config.trim);
                } catch (...) { errs[w] = std::current_exception(); }
            });
        for (auto& t : makers) t.join();
        for (auto& e : errs)
            if (e) { feeder_p->fail("cancelled"); std::rethrow_exception(e); }
    }
    const double t_ctx_done = now0();
    FILE* out = fopen(out_file.c_str(), "w");
    if (!out) throw BarbellError(BB_E_INVALID, "Failed to create annotation output file '" + out_file + "'");
    FILE* kept_f = nullptr;
    FILE* drop_f = nullptr;
    if (filtering) {
        if (!config.filtered_file.empty() && !(kept_f = fopen(config.filtered_file.c_str(), "w"))) {
            fclose(out);
            throw BarbellError(BB_E_INVALID, "Failed to create filtered output file '" + config.filtered_file + "'");
        }
        if (!config.dropped_file.empty() && !(drop_f = fopen(config.dropped_file.c_str(), "w"))) {
            fclose(out);
            if (kept_f) fclose(kept_f);
            throw BarbellError(BB_E_INVALID, "Failed to create dropped output file '" + config.dropped_file + "'");
        }
    }
    AnnotateStats st;
    std::unique_ptr<LabelWriters> writers;
    FILE* failed_f = nullptr;
    if (trimming) {
        if (mkdir(config.trim_folder.c_str(), 0777) != 0 && errno != EEXIST) {
            fclose(out);
            throw BarbellError(BB_E_INVALID, "Failed to create output folder '" + config.trim_folder + "'");
        }
        writers = std::make_unique<LabelWriters>(config.trim_folder, config.trim->gzip, getenv("BARBELL_AMD_WRITERS") ? 0u : (host_cut ? 16u : 8u));
        if (config.trim->failed_trimmed_writer) failed_f = fopen(config.trim->failed_trimmed_writer->c_str(), "w");
    }
    FILE* ppr_f = nullptr;
    if (config.inspect && !config.read_pattern_out.empty()) ppr_f = fopen(config.read_pattern_out.c_str(), "w");
    std::map<std::string, size_t> pattern_count;
    std::vector<std::string> pattern_order;  // first-appearance order, for a deterministic tie order in the summary
    bool header = false, kept_header = false, drop_header = false;
    auto close_all = [&]() {
        fclose(out);
        if (kept_f) fclose(kept_f);
        if (drop_f) fclose(drop_f);
        if (failed_f) fclose(failed_f);
        if (ppr_f) fclose(ppr_f);
        writers.reset();
    };
    const bool prof = getenv("BARBELL_AMD_PROFILE") != nullptr;
    auto now = []() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const bool want_ids = ppr_f != nullptr || failed_f != nullptr;
    const bool feed_only = getenv("BARBELL_AMD_FEED_ONLY") != nullptr;   // measurement aid (tools/e2e_rate.py): what one host can feed, whatever the GPUs do
    std::atomic<uint64_t> fed_bytes{0};

    // ---- one block on its context: parsed, annotated, rendered, filtered, inspected and trimmed in HBM -------------
    auto process = [&](Demuxer& dm, const BlockFeeder::Block& blk, const std::shared_ptr<BlockFeeder>& feeder) -> BlockResult {
        BlockResult R;
        double t0 = now();
        if (feed_only) {  // BARBELL_AMD_FEED_ONLY=1: the host side alone — files -> reader threads -> blocks of whole records —, nothing uploaded
            fed_bytes += blk.len;
            feeder->release(blk.slot);
            return R;
        }
        const auto ing = dm.ingest(blk.data, blk.len, true, want_ids, two_line, packed);  // blocks hold whole records only
        std::shared_ptr<void> text_hold;   // host_cut: the slot stays until the writer threads have cut the block's records out of it
        if (host_cut) {
            const int slot = blk.slot;
            auto big = blk.big;
            text_hold = std::shared_ptr<void>((void*)blk.data, [feeder, slot, big](void*) { feeder->release(slot); });
        } else feeder->release(blk.slot);                               // the text is in HBM: the slot can be refilled
        R.t_ingest = now() - t0; t0 = now();
        const auto& ids = ing.ids;
        R.n_reads = (size_t)ing.info.n_records;
        if (R.n_reads == 0) return R;
        const uint64_t n_rows = dm.annotate_ingested();
        R.rows = (size_t)n_rows;
        const bb_row* rows = dm.rows();
        for (uint64_t i = 0; i < n_rows; ++i) R.found += i == 0 || rows[i].read_idx != rows[i - 1].read_idx;
        dm.format_ingested(BB_FMT_ALL, R.anno);
        R.t_gpu = now() - t0; t0 = now();
        std::vector<bb_row_verdict> verdicts;
        double t1 = now();
        if (filtering) {
            verdicts = dm.filter_ingested();
            for (uint64_t i = 0; i < n_rows; ++i)
                if (i == 0 || rows[i].read_idx != rows[i - 1].read_idx) ++(verdicts[i].pass ? R.kept : R.dropped);
            if (kept_f) dm.format_ingested(BB_FMT_KEPT, R.kept_tsv);
            if (drop_f) dm.format_ingested(BB_FMT_DROPPED, R.drop_tsv);
        }
        R.t_filter = now() - t1; t1 = now();
        if (config.inspect) {  // inspect.rs:128-184 on the annotation rows (no cuts yet)
            // counted per block here, merged by the commit stage (it used to look every read's string up in one map)
            std::vector<std::string> distinct;
            std::vector<std::pair<uint32_t, uint32_t>> per_read;   // (read, index into distinct), read order; distinct is in first-appearance order
            dm.inspect_ingested_interned(false, config.bucket_size, distinct, per_read);
            for (auto& d : distinct) R.patterns.emplace_back(d, 0);
            if (ppr_f) R.ppr.reserve(per_read.size() * 64);
            for (const auto& rp : per_read) {
                if (ppr_f) { R.ppr += ids[rp.first]; R.ppr += '\t'; R.ppr += distinct[rp.second]; R.ppr += '\n'; }
                ++R.patterns[rp.second].second;
            }
        }
        R.t_inspect = now() - t1; t1 = now();
        if (trimming && host_cut) {  // trim.rs:385-460: the GPU decided slices, labels and the layout of every label's records; the writers copy
            auto cut = std::make_shared<LabelWriters::Cut>();
            cut->text = blk.data; cut->plan = dm.trim_plan_ingested(); cut->cfg = dm.trim_config_pod(); cut->hold = text_hold;
            const TrimPlan& t = cut->plan;
            for (const auto& sp : t.spans) R.spans.push_back({dm.label_of_key(sp.label_key), (size_t)sp.off, (size_t)sp.len, sp.first, sp.n_records});
            std::vector<uint32_t> per_read(R.n_reads, 0);
            for (const auto& sl : t.slices) ++per_read[sl.read_idx];
            for (size_t i = 0; i < R.n_reads; ++i) {
                if (t.status[i] == BB_TRIM_TRIMMED) ++R.trimmed;
                if (per_read[i] > 1) ++R.split;
                if (t.status[i] == BB_TRIM_FAILED) { ++R.trim_failed; if (failed_f) { R.failed_ids += ids[i]; R.failed_ids += '\n'; } }
            }
            R.cut = std::move(cut);
        } else if (trimming) {  // the GPU cut and rendered the records, one write per label
            const TrimBatch t = dm.trim_ingested();
            R.text = t.text_hold; R.text_ptr = t.data();  // the page-locked landing buffer goes back to the demuxer's pool when the writers are done
            for (const auto& sp : t.spans) R.spans.push_back({dm.label_of_key(sp.label_key), (size_t)sp.off, (size_t)sp.len});
            std::vector<uint32_t> per_read(R.n_reads, 0);
            for (const auto& sl : t.slices) ++per_read[sl.read_idx];
            for (size_t i = 0; i < R.n_reads; ++i) {
                if (t.status[i] == BB_TRIM_TRIMMED) ++R.trimmed;
                if (per_read[i] > 1) ++R.split;
                if (t.status[i] == BB_TRIM_FAILED) { ++R.trim_failed; if (failed_f) { R.failed_ids += ids[i]; R.failed_ids += '\n'; } }
            }
        }
        R.t_trim = now() - t1;
        R.t_rest = now() - t0;
        return R;
    };
    double t_starved = 0;  // workers waiting for a block from the reader pool (under `mu`)
    double t_commit = 0, t_ingest = 0, t_gpu = 0, t_rest = 0, t_filter = 0, t_inspect = 0, t_trim = 0, t_wwait = 0;
    auto commit = [&](BlockResult& R) {
        const double t0 = now();
        st.total += R.n_reads; st.found += R.found; st.rows += R.rows; st.kept += R.kept; st.dropped += R.dropped;
        st.trimmed += R.trimmed; st.trimmed_split += R.split; st.trim_failed += R.trim_failed;
        auto put = [](FILE* f, bool& hdr, const std::vector<uint8_t>& text) {
            if (!f || text.empty()) return;
            if (!hdr) { fputs(TSV_HEADER, f); fputc('\n', f); hdr = true; }  // csv writer: header with the first record
            if (fwrite(text.data(), 1, text.size(), f) != text.size()) throw BarbellError(BB_E_INVALID, "Failed to write annotation rows");
        };
        put(out, header, R.anno);
        put(kept_f, kept_header, R.kept_tsv);
        put(drop_f, drop_header, R.drop_tsv);
        if (ppr_f && !R.ppr.empty()) fwrite(R.ppr.data(), 1, R.ppr.size(), ppr_f);
        for (auto& pc : R.patterns) {
            auto it = pattern_count.find(pc.first);
            if (it == pattern_count.end()) { pattern_count.emplace(pc.first, pc.second); pattern_order.push_back(pc.first); }
            else it->second += pc.second;
        }
        if (failed_f && !R.failed_ids.empty()) fwrite(R.failed_ids.data(), 1, R.failed_ids.size(), failed_f);
        if (writers && !R.spans.empty()) {
            const double tw = now();
            writers->wait(3);  // bounds the rendered text waiting for the writer threads
            t_wwait += now() - tw;
            std::vector<LabelWriters::Span> job;
            for (const auto& sp : R.spans) {
                if (R.cut) job.push_back({sp.label, nullptr, sp.n, nullptr, R.cut, sp.first, sp.off, sp.n_records});
                else job.push_back({sp.label, R.text_ptr + sp.off, sp.n, R.text, nullptr, 0, 0, 0});
            }
            R.cut.reset();
            writers->submit(std::move(job));
        }
        t_ingest += R.t_ingest; t_gpu += R.t_gpu; t_rest += R.t_rest; t_filter += R.t_filter; t_inspect += R.t_inspect; t_trim += R.t_trim;
        t_commit += now() - t0;
    };

    // ---- the pipeline: readers -> sequencer (dispatcher thread) -> G workers -> ordered commit (this thread) -------
    std::mutex mu;
    std::condition_variable cv;
    std::vector<std::deque<BlockFeeder::Block>> inq(G);
    std::map<uint64_t, BlockResult> results;
    bool feed_done = false, abort = false;
    uint64_t n_blocks = 0, next_commit = 0;
    std::exception_ptr first_err;
    auto set_err = [&](std::exception_ptr e) { { std::lock_guard<std::mutex> lk(mu); if (!first_err) first_err = e; abort = true; } cv.notify_all(); };
    double t_start = 0, t_end = 0;
    try {
        BlockFeeder& feeder = *feeder_p;
        t_start = now();
        if (prof0) fprintf(stderr, "profile: start-up: feeder (files opened, reader threads started) %.3f s, %zu context(s) side by side %.3f s\n", t_feeder_up - t_enter, G, t_ctx_done - t_feeder_up);
        std::thread dispatcher([&]() {
            try {
                BlockFeeder::Block b;
                while (feeder.next(b)) {
                    std::unique_lock<std::mutex> lk(mu);
                    auto& q = inq[b.index % G];
                    cv.wait(lk, [&]() { return abort || q.size() < 2; });
                    if (abort) return;
                    q.push_back(b);
                    ++n_blocks;
                    cv.notify_all();
                }
            } catch (...) { set_err(std::current_exception()); }
            { std::lock_guard<std::mutex> lk(mu); feed_done = true; }
            cv.notify_all();
        });
        std::vector<std::thread> workers;
        for (size_t w = 0; w < G; ++w)
            workers.emplace_back([&, w]() {
                try {
                    for (;;) {
                        BlockFeeder::Block b;
                        {
                            const double tw0 = now();
                            std::unique_lock<std::mutex> lk(mu);
                            cv.wait(lk, [&]() { return abort || !inq[w].empty() || feed_done; });
                            t_starved += now() - tw0;
                            if (abort) return;
                            if (inq[w].empty()) return;  // feed_done
                            b = inq[w].front();
                            // do not run far ahead of the committer (bounds the rendered text held in `results`); the block the
                            // committer waits for is always inside the window, so this cannot deadlock
                            cv.wait(lk, [&]() { return abort || b.index < next_commit + 2 * G + 2; });
                            if (abort) return;
                            inq[w].pop_front();
                        }
                        cv.notify_all();
                        BlockResult R = process(*dms[w], b, feeder_p);
                        {
                            std::lock_guard<std::mutex> lk(mu);
                            results.emplace(b.index, std::move(R));
                        }
                        cv.notify_all();
                    }
                } catch (...) { set_err(std::current_exception()); }
            });
        try {
            for (;;) {
                BlockResult R;
                {
                    std::unique_lock<std::mutex> lk(mu);
                    cv.wait(lk, [&]() { return abort || results.count(next_commit) || (feed_done && next_commit >= n_blocks); });
                    if (abort) break;
                    if (!results.count(next_commit)) break;  // all committed
                    R = std::move(results[next_commit]);
                    results.erase(next_commit);
                }
                commit(R);
                { std::lock_guard<std::mutex> lk(mu); ++next_commit; }
                cv.notify_all();
            }
        } catch (...) { set_err(std::current_exception()); }
        { std::lock_guard<std::mutex> lk(mu); if (first_err) abort = true; }
        cv.notify_all();
        feeder.fail(first_err ? "cancelled" : "");  // unblocks readers if we are bailing out (no-op message at the normal end)
        dispatcher.join();
        for (auto& t : workers) t.join();
        t_end = now();
        if (first_err) std::rethrow_exception(first_err);
        if (writers) writers->wait(0);
        feeder_p.reset();
        if (prof0) fprintf(stderr, "profile: feeder torn down %.3f s after the last block; %.3f s since annotate() was entered\n", now() - t_end, now() - t_enter);  // all records on disk (or the writer's error rethrown) before the files are closed
    } catch (...) {
        close_all();
        throw;
    }
    close_all();
    st.seconds_pipeline = t_end - t_start;
    // per-barcode histogram over all contexts (the one collective of the path)
    {
        std::vector<Demuxer*> ptrs;
        for (auto& d : dms) ptrs.push_back(d.get());
        std::vector<uint64_t> total = allreduce_counts(ptrs, st.counts_reduce);
        bool counts_mine = true;   // does this process write counts_file?
        if (!config.rccl_id.empty()) {
            std::string how;
            total = allreduce_counts_shards(ptrs[0], total, config.shard_rank, config.shard_world, config.rccl_id, how);
            st.counts_reduce += " + " + how;
            counts_mine = config.shard_rank == 0;
        } else if (config.shard_world > 1) {
            fprintf(stderr, "warning: --shard %u/%u without --rccl-id: the counts are this process's own, not the run's (give every shard the same --rccl-id PATH "
                            "for one all-reduced histogram)\n", config.shard_rank, config.shard_world);
        }
        const std::vector<std::string> labels = dms[0]->slot_labels();
        for (size_t i = 0; i < total.size(); ++i) st.counts.emplace_back(labels[i], total[i]);
        if (!config.counts_file.empty() && counts_mine) {
            FILE* cf = fopen(config.counts_file.c_str(), "w");
            if (!cf) throw BarbellError(BB_E_INVALID, "Failed to create counts file '" + config.counts_file + "'");
            size_t gi = 0, left = dms[0]->queries().empty() ? 0 : dms[0]->queries()[0].labels.size() + 1;
            for (size_t i = 0; i < total.size(); ++i) {
                fprintf(cf, "%zu\t%s\t%llu\n", gi, labels[i].c_str(), (unsigned long long)total[i]);
                if (--left == 0 && gi + 1 < dms[0]->queries().size()) { ++gi; left = dms[0]->queries()[gi].labels.size() + 1; }
            }
            fclose(cf);
        }
    }
    if (feed_only) fprintf(stderr, "feed-only: %llu bytes of staged text in %.3f s (%.2f GB/s into the block buffers; no GPU work)\n", (unsigned long long)fed_bytes.load(),
                           st.seconds_pipeline, st.seconds_pipeline > 0 ? (double)fed_bytes.load() / st.seconds_pipeline / 1e9 : 0.0);
    if (prof) fprintf(stderr, "profile: pipeline %.3f s for %zu reads (%.2f M reads/s) on %zu context(s); summed over blocks: upload+parse %.3f s, annotate+render %.3f s, "
                      "filter/inspect/trim %.3f s (%.3f / %.3f / %.3f); commit (file writes) %.3f s, of which waiting for the label writers %.3f s; workers waiting for input %.3f s\n",
                      st.seconds_pipeline, st.total, st.seconds_pipeline > 0 ? st.total / st.seconds_pipeline / 1e6 : 0.0, G, t_ingest, t_gpu, t_rest,
                      t_filter, t_inspect, t_trim, t_commit, t_wwait, t_starved);
    for (const auto& p : pattern_order) st.patterns.emplace_back(p, pattern_count[p]);
    std::stable_sort(st.patterns.begin(), st.patterns.end(), [](const auto& a, const auto& c) { return a.second > c.second; });
    return st;
}

AnnotateStats annotate_with_groups(const std::vector<std::string>& read_files, const std::string& out_file,
                                   std::vector<BarcodeGroup> query_groups, const AnnotateConfig& config) {
    for (auto& g : query_groups)
        if (config.max_flank_errors) g.set_flank_threshold(*config.max_flank_errors);  // else: automatic cutoff inside bb_create
    return annotate(read_files, out_file, std::move(query_groups), config);
}
AnnotateStats annotate_with_kit(const std::vector<std::string>& read_files, const std::string& out_file, const std::string& kit,
                                const AnnotateConfig& config) {
    return annotate_with_groups(read_files, out_file, BarcodeGroup::new_from_kit(kit, config.use_extended), config);
}
AnnotateStats annotate_with_files(const std::vector<std::string>& read_files, const std::vector<std::string>& query_files,
                                  const std::vector<BarcodeType>& query_types, const std::string& out_file,
                                  const AnnotateConfig& config) {
    if (query_files.size() != query_types.size())
        throw BarbellError(BB_E_INVALID, "Expected the same number of query files and barcode types, got " +
                                             std::to_string(query_files.size()) + " query file(s) and " +
                                             std::to_string(query_types.size()) + " barcode type(s)");
    std::vector<BarcodeGroup> groups;
    for (size_t i = 0; i < query_files.size(); ++i) groups.push_back(BarcodeGroup::new_from_fasta(query_files[i], query_types[i]));
    return annotate_with_groups(read_files, out_file, std::move(groups), config);
}

AnnotateStats demux_using_kit(const std::vector<std::string>& fastq_files, const KitConfig& k) {
    if (mkdir(k.output_folder.c_str(), 0777) != 0 && errno != EEXIST)
        throw BarbellError(BB_E_INVALID, "Failed to create output folder '" + k.output_folder + "'");
    AnnotateConfig c;
    c.max_flank_errors = k.max_flank_errors; c.alpha = k.alpha; c.n_threads = (unsigned)k.threads; c.verbose = k.verbose;
    c.min_score = k.min_score; c.min_score_diff = k.min_score_diff; c.use_extended = k.use_extended;
    c.batch_reads = k.batch_reads; c.device = k.device; c.devices = k.devices; c.streams_per_device = k.streams_per_device; c.counts_file = k.counts_file;
    c.shard_rank = k.shard_rank; c.shard_world = k.shard_world; c.rccl_id = k.rccl_id; c.shard_by_bytes = k.shard_by_bytes;
    c.filter_patterns = kit_patterns(k.kit_name, k.maximize);
    c.filtered_file = k.output_folder + "/filtered.tsv";
    c.trim = TrimConfig::for_kit(k.failed_out, k.gzip);
    c.trim_folder = k.output_folder;
    c.inspect = true;
    c.host_cut = k.host_cut;
    c.process_exits_after = k.process_exits_after;
    c.read_pattern_out = k.output_folder + "/pattern_per_read.tsv";
    return annotate_with_kit(fastq_files, k.output_folder + "/annotation.tsv", k.kit_name, c);
}

}  // namespace barbell
