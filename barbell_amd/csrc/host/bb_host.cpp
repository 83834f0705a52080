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

}  // namespace barbell
