// barbell-amd — command-line driver of the MI355X annotate path.  Mirrors the flags and defaults of
// the reference's `barbell annotate` (bin/main.rs:64-112).  The filter step (`barbell filter`,
// bin/main.rs:114-135) is available fused into annotate: --filter-file / --kit-filter [--maximize] with
// --filtered / --dropped outputs, and so are trim (`barbell trim`, bin/main.rs:136-186: --trim-output DIR plus
// its label flags) and inspect (--inspect / --read-pattern-out).  `barbell-amd kit` is `barbell kit`
// (bin/main.rs:208-262, use_kit.rs:11-109) in one pass over the reads.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <time.h>
#include <unistd.h>
#include <vector>

#include "bb_host.hpp"

using namespace barbell;

// --shard R/W: this process takes the input files whose index is R modulo W (one process per GPU with --device R;
// reads shard trivially, SURVEY §8e — each process writes its own rows; with --rccl-id PATH (the same fresh path for every shard) the
// per-barcode histograms are all-reduced over the W processes and shard 0 writes --counts)
// --shard-by bytes: every process keeps the whole file list and takes the records that start in its R-th of W equal byte ranges of each file
// (one big FASTQ over W GPUs; plain files only — a gzip stream has no entry points)
static bool apply_shard(const std::string& spec, std::vector<std::string>& files, uint32_t& rank, uint32_t& world, bool by_bytes = false) {
    if (spec.empty()) return true;
    const size_t slash = spec.find('/');
    if (slash == std::string::npos) return false;
    const long r = atol(spec.substr(0, slash).c_str()), w = atol(spec.substr(slash + 1).c_str());
    if (w < 1 || r < 0 || r >= w) return false;
    rank = (uint32_t)r; world = (uint32_t)w;
    if (by_bytes) return true;
    std::vector<std::string> mine;
    for (size_t i = 0; i < files.size(); ++i)
        if ((long)(i % (size_t)w) == r) mine.push_back(files[i]);
    files.swap(mine);
    return true;
}

// sizes: a byte count with an optional binary suffix (4096, 64Ki, 256Mi, 1Gi; K / M / G mean the same)
static size_t parse_size(const char* v, const char* flag) {
    char* end = nullptr;
    const unsigned long long x = strtoull(v, &end, 10);
    unsigned shift = 0;
    if (end == v) { fprintf(stderr, "error: %s takes a byte count (e.g. 4096, 256Mi)\n", flag); exit(2); }
    if (*end == 'K' || *end == 'k') shift = 10; else if (*end == 'M' || *end == 'm') shift = 20; else if (*end == 'G' || *end == 'g') shift = 30;
    if (shift) ++end;
    if (shift && (*end == 'i' || *end == 'I')) ++end;
    if (shift && (*end == 'B' || *end == 'b')) ++end;
    if (*end) { fprintf(stderr, "error: %s takes a byte count (e.g. 4096, 256Mi), got '%s'\n", flag, v); exit(2); }
    return (size_t)(x << shift);
}

static bool parse_shard_by(const std::string& v, bool& by_bytes) {
    if (v == "bytes") { by_bytes = true; return true; }
    if (v == "files") { by_bytes = false; return true; }
    fputs("error: --shard-by takes files or bytes\n", stderr);
    return false;
}

// --devices 0,1,2 (a device may repeat: 0,0 = two contexts on GPU 0)
static std::vector<int> parse_devices(const std::string& spec) {
    std::vector<int> out;
    size_t pos = 0;
    while (pos <= spec.size()) {
        const size_t c = spec.find(',', pos);
        const std::string tok = spec.substr(pos, c == std::string::npos ? std::string::npos : c - pos);
        if (tok.empty() || tok.find_first_not_of("0123456789") != std::string::npos) { fprintf(stderr, "error: --devices takes a comma-separated list of device ordinals\n"); exit(2); }
        out.push_back(atoi(tok.c_str()));
        if (c == std::string::npos) break;
        pos = c + 1;
    }
    return out;
}

// --policy: the text form of include/barbell_amd_policy.h (what the un-vendored crates are assumed to do where Barbell's own
// code does not pin it); checked here, handed to every context through BARBELL_AMD_POLICY (bb_create reads it)
static bool set_policy(const char* text) {
    bb_policy p;
    bb_policy_default(&p);
    if (bb_policy_parse(text, &p) != 0) {
        fprintf(stderr, "error: --policy '%s': expected e.g. lm=left,rc=fwd,trace=MSID,ovh=ceil,tie=last,lodhi=3:0.5:2211\n", text);
        return false;
    }
    setenv("BARBELL_AMD_POLICY", text, 1);
    return true;
}

static void usage() {
    fputs(
        "Usage: barbell-amd annotate -i <FASTQ>... [-o output.tsv] (--kit <KIT> | -q <FASTA>... [-b Ftag|Rtag ...])\n"
        "                            [--flank-max-errors INT] [--min-score F=0.2] [--min-score-diff F=0.1]\n"
        "                            [--alpha F=0.4] [--use-extended] [-t THREADS=auto (reader / inflater threads: the CPUs the process may use, at most 32; the reference's default is 10)] [--verbose]\n"
        "                            [--block-bytes N=256Mi | --batch-reads N (= N*4096 bytes)] [--device D=0]\n"
        "                            [--shard R/W [--shard-by files|bytes (files: the inputs with index R mod W; bytes: the records that start in the R-th of W byte ranges of each plain file)]\n"
        "                             [--rccl-id PATH (one fresh path for all W processes: their histograms are all-reduced, RCCL ncclCommInitRank; shard 0 writes --counts)]]\n"
        "                            [--devices D0,D1,.. (one FASTQ stream over several contexts, block i -> context i mod G; RCCL all-reduce of the counts)]\n"
        "                            [--streams S=2 (contexts per device when --devices is not given)] [--counts FILE]\n"
        "                            [--policy lm=..,rc=..,trace=..,ovh=..,tie=..,lodhi=.. (include/barbell_amd_policy.h)]\n"
        "                            [--no-compact (upload the quality lines too; without --trim-output they are dropped on the host)]\n"
        "                            [--no-pack (upload the sequence lines as text; by default two bases per byte: the kernels only look at IUPAC base sets)]\n"
        "                            [(-f <PATTERN_FILE>... | --kit-filter [--maximize]) [--filtered FILE] [--dropped FILE]]\n"
        "                            [--trim-output DIR [--no-label] [--no-orientation] [--no-flanks] [--sort-labels]\n"
        "                             [--only-side left|right] [--failed-out FILE] [--skip-trim] [--flip] [--gzip]\n"
        "                             [--gpu-render (records rendered in HBM and downloaded; default: the GPU plans, the file writers cut them out of the staged text)]]\n"
        "                            [--inspect [-n TOP=10] [--read-pattern-out FILE] [-s BUCKET=250]]\n"
        "       barbell-amd kit -k <KIT> -i <FASTQ>... -o <OUT_DIR> [--maximize] [--min-score F] [--min-score-diff F]\n"
        "                       [--flank-max-errors INT] [--failed-out FILE] [--use-extended] [--alpha F] [--gzip] [-t N=auto]\n"
        "                       [--device D=0] [--shard R/W [--shard-by files|bytes] [--rccl-id PATH]] [--gpu-render]\n"
        "       barbell-amd filter -i <annotation.tsv> -o <filtered.tsv> -f <PATTERN_FILE>... [--dropped FILE] [--device D=0]\n"
        "       barbell-amd inspect -i <annotation.tsv> [-n TOP=10] [-o pattern_per_read.tsv] [-s BUCKET=250] [--device D=0]\n"
        "       barbell-amd trim -i <filtered.tsv> -r <FASTQ>... -o <OUT_DIR> [--no-label] [--no-orientation] [--no-flanks] [--sort-labels]\n"
        "                        [--only-side left|right] [--failed-out FILE] [--skip-trim] [--flip] [--gzip] [--device D=0]\n"
        "                        (the reference's stand-alone steps on files written earlier; annotate / kit do all of it in one pass)\n"
        "       barbell-amd kits          list the supported kit names\n"
        "       barbell-amd pattern <STR>...   parse filter pattern strings and print their elements\n",
        stderr);
}

// Every output file is closed by now: leave without the HIP runtime's exit handlers and the un-page-locking of the block buffers (together
// 0.5-0.9 s of a 2.6 s run on 4 M reads); the OS takes everything back with the process.
static double g_t_main = 0.0;
static double mono() { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec; }
static int done_ok() {
    if (getenv("BARBELL_AMD_PROFILE")) fprintf(stderr, "profile: %.3f s between main() and exit\n", mono() - g_t_main);
    fflush(stdout);
    fflush(stderr);
    _exit(0);
}

int main(int argc, char** argv) {
    g_t_main = mono();
    if (argc < 2) { usage(); return 2; }
    const std::string cmd = argv[1];
    if (cmd == "kits") {
        for (const auto& k : supported_kits()) puts(k.c_str());
        return 0;
    }
    if (cmd == "pattern") {  // parse check: one canonical line per element (type|orientation|label|?N|rel|lo|hi|cuts)
        try {
            for (int i = 2; i < argc; ++i) {
                const Pattern p = pattern_from_str(argv[i]);
                for (const auto& e : p.elements) {
                    std::string cuts;
                    for (const auto& c : e.cuts) cuts += (cuts.empty() ? "" : ",") + c.to_string();
                    printf("%s|%d|%s|%d|%d|%ld|%ld|%s\n", as_str(e.match_type), e.orientation, e.label ? e.label->c_str() : "*",
                           e.placeholder, e.relative_to, e.range_lo, e.range_hi, cuts.c_str());
                }
                puts("--");
            }
        } catch (const BarbellError& e) { fprintf(stderr, "error: %s\n", e.what()); return 1; }
        return 0;
    }
    if (cmd == "filter" || cmd == "inspect" || cmd == "trim") {   // the stand-alone steps on files (bb_steps.cpp); flags as bin/main.rs:114-206
        std::string input, output, dropped, failed_out, read_pattern_out;
        std::vector<std::string> files, reads;
        std::vector<std::string>* multi = nullptr;
        TrimConfig tcfg;
        size_t top_n = 10;
        uint32_t bucket = 250;
        int device = 0;
        for (int i = 2; i < argc; ++i) {
            const std::string a = argv[i];
            auto need = [&](const char* what) -> const char* { if (i + 1 >= argc) { fprintf(stderr, "error: %s needs a value\n", what); exit(2); } multi = nullptr; return argv[++i]; };
            if (a == "-i" || a == "--input") input = need("--input");
            else if (a == "-o" || a == "--output" || (cmd == "inspect" && a == "--read-pattern-out")) output = need("--output");
            else if ((a == "-f" || a == "--file") && cmd == "filter") multi = &files;
            else if ((a == "-r" || a == "--reads") && cmd == "trim") multi = &reads;
            else if (a == "--dropped" && cmd == "filter") dropped = need("--dropped");
            else if ((a == "-n" || a == "--top-n") && cmd == "inspect") top_n = (size_t)atol(need("--top-n"));
            else if ((a == "-s" || a == "--bucket-size") && cmd == "inspect") bucket = (uint32_t)atol(need("--bucket-size"));
            else if (a == "--device") device = atoi(need("--device"));
            else if (a == "--verbose") { tcfg.verbose = true; multi = nullptr; }
            else if (cmd == "trim" && a == "--no-label") { tcfg.add_labels = false; multi = nullptr; }
            else if (cmd == "trim" && a == "--no-orientation") { tcfg.add_orientation = false; multi = nullptr; }
            else if (cmd == "trim" && a == "--no-flanks") { tcfg.add_flank = false; multi = nullptr; }
            else if (cmd == "trim" && a == "--sort-labels") { tcfg.sort_labels = true; multi = nullptr; }
            else if (cmd == "trim" && a == "--skip-trim") { tcfg.skip_trim = true; multi = nullptr; }
            else if (cmd == "trim" && a == "--flip") { tcfg.flip = true; multi = nullptr; }
            else if (cmd == "trim" && a == "--gzip") { tcfg.gzip = true; multi = nullptr; }
            else if (cmd == "trim" && a == "--failed-out") tcfg.failed_trimmed_writer = std::string(need("--failed-out"));
            else if (cmd == "trim" && a == "--only-side") {
                const std::string v = need("--only-side");
                if (v == "left") tcfg.only_side = LabelSide::Left;
                else if (v == "right") tcfg.only_side = LabelSide::Right;
                else { fprintf(stderr, "error: --only-side takes left or right\n"); return 2; }
            }
            else if (!a.empty() && a[0] != '-' && multi) multi->push_back(a);
            else { fprintf(stderr, "error: unexpected argument '%s'\n", a.c_str()); usage(); return 2; }
        }
        if (input.empty()) { fprintf(stderr, "error: %s needs --input\n", cmd.c_str()); return 2; }
        try {
            if (cmd == "filter") {
                if (output.empty() || files.empty()) { fputs("error: filter needs --output and --file\n", stderr); return 2; }
                puts("Starting filtering...");
                const StepStats st = filter_file(input, output, dropped.empty() ? std::nullopt : std::optional<std::string>(dropped), patterns_from_files(files), device);
                printf("Filtering complete! %zu reads, %zu kept, %zu dropped\n", st.total, st.kept, st.dropped);
                if (tcfg.verbose) write_progress_log("filter", parent_dir(output), {{"Total:", st.total}, {"Kept:", st.kept}, {"Dropped:", st.dropped}});   // filter.rs:18-25
            } else if (cmd == "inspect") {
                puts("Inspecting...");
                AnnotateStats pats;
                inspect_file(input, output.empty() ? std::nullopt : std::optional<std::string>(output), bucket, pats, device);
                for (const auto& l : inspect_summary(pats, top_n)) puts(l.c_str());
            } else {
                if (output.empty() || reads.empty()) { fputs("error: trim needs --output and --reads\n", stderr); return 2; }
                if (tcfg.sort_labels && tcfg.only_side) { fputs("error: --only-side conflicts with --sort-labels (bin/main.rs:160-161)\n", stderr); return 2; }
                puts("Starting trimming...");
                const StepStats st = trim_file(input, reads, output, tcfg, device);
                printf("Trimming complete! %zu reads, %zu trimmed (%zu split), %zu failed\n", st.total, st.kept, st.split, st.dropped);
                if (tcfg.verbose) write_progress_log("trim", output, {{"Total:", st.total}, {"Kept:", st.kept}, {"Kept split:", st.split}, {"Failed:", st.dropped}});   // trim.rs:341-345
            }
        } catch (const std::exception& e) { fprintf(stderr, "error: %s\n", e.what()); return 1; }
        return done_ok();
    }
    if (cmd == "rendezvous") {   // test aid, no GPU: one rank of the --shard R/W --rccl-id meeting, summing the given counts through the files
        std::string base, shard, bus = "cpu", counts;
        double delay = 0;
        bool say_hello = true;
        for (int i = 2; i < argc; ++i) {
            const std::string a = argv[i];
            if (a == "--rccl-id" && i + 1 < argc) base = argv[++i];
            else if (a == "--shard" && i + 1 < argc) shard = argv[++i];
            else if (a == "--bus" && i + 1 < argc) bus = argv[++i];
            else if (a == "--counts" && i + 1 < argc) counts = argv[++i];
            else if (a == "--start-delay" && i + 1 < argc) delay = atof(argv[++i]);   // seconds before this rank "starts" (says hello)
            else if (a == "--no-hello") say_hello = false;
            else { fprintf(stderr, "error: unexpected argument '%s'\n", a.c_str()); return 2; }
        }
        std::vector<std::string> none;
        uint32_t rank = 0, world = 1;
        if (base.empty() || shard.empty() || !apply_shard(shard, none, rank, world, true)) { fputs("usage: barbell-amd rendezvous --rccl-id PATH --shard R/W --counts a,b,c [--bus NAME] [--start-delay S]\n", stderr); return 2; }
        std::vector<uint64_t> local;
        for (size_t pos = 0; pos < counts.size();) { const size_t c = counts.find(',', pos); local.push_back(strtoull(counts.substr(pos, c - pos).c_str(), nullptr, 10)); if (c == std::string::npos) break; pos = c + 1; }
        try {
            if (delay > 0) usleep((useconds_t)(delay * 1e6));
            if (say_hello) shard_rendezvous_reset(base, rank);
            bool shared = false;
            const std::vector<uint64_t> total = rendezvous_sum_counts(base, rank, world, bus, local, &shared);
            printf("total");
            for (uint64_t v : total) printf(" %llu", (unsigned long long)v);
            printf("\nshared_device %d\n", shared ? 1 : 0);
        } catch (const BarbellError& e) { fprintf(stderr, "error: %s\n", e.what()); return 1; }
        return 0;
    }
    if (cmd == "stage") {   // test aid, no GPU: the staged upload text of the annotate path -> a file
        std::vector<std::string> in;
        std::string out;
        size_t block = 128u << 20;
        unsigned threads = 4;
        bool two_line = true, pack = true, multi_in = false, by_bytes = false;
        std::string shard;
        uint32_t srank = 0, sworld = 1;
        for (int i = 2; i < argc; ++i) {
            const std::string a = argv[i];
            if (a == "-i") multi_in = true;
            else if (a == "-o" && i + 1 < argc) { out = argv[++i]; multi_in = false; }
            else if (a == "--block-bytes" && i + 1 < argc) { block = parse_size(argv[++i], "--block-bytes"); multi_in = false; }
            else if (a == "-t" && i + 1 < argc) { threads = (unsigned)atoi(argv[++i]); multi_in = false; }
            else if (a == "--no-pack") { pack = false; multi_in = false; }
            else if (a == "--no-compact") { two_line = false; pack = false; multi_in = false; }
            else if (a == "--shard" && i + 1 < argc) { shard = argv[++i]; multi_in = false; }
            else if (a == "--shard-by" && i + 1 < argc) { if (!parse_shard_by(argv[++i], by_bytes)) return 2; multi_in = false; }
            else if (multi_in && !a.empty() && a[0] != '-') in.push_back(a);
            else { fprintf(stderr, "error: unexpected argument '%s'\n", a.c_str()); return 2; }
        }
        if (!apply_shard(shard, in, srank, sworld, by_bytes)) { fputs("error: --shard takes R/W with 0 <= R < W\n", stderr); return 2; }
        if (!by_bytes) { srank = 0; sworld = 1; }
        if (in.empty() || out.empty()) { fputs("usage: barbell-amd stage -i FASTQ... -o FILE [--block-bytes N] [-t N] [--no-pack] [--no-compact] [--shard R/W [--shard-by files|bytes]]\n", stderr); return 2; }
        try {
            size_t nb = 0;
            const int form = stage_blocks(in, block, threads, two_line, pack, out, nb, srank, sworld);
            printf("form %d blocks %zu\n", form, nb);
        } catch (const BarbellError& e) { fprintf(stderr, "error: %s\n", e.what()); return 1; }
        return 0;
    }
    if (cmd == "-h" || cmd == "--help") { usage(); return 0; }
    if (cmd == "kit") {
        KitConfig k;
        std::vector<std::string> input;
        std::string shard;
        bool multi_in = false;
        k.process_exits_after = true;
        for (int i = 2; i < argc; ++i) {
            const std::string a = argv[i];
            auto need = [&](const char* what) -> const char* { if (i + 1 >= argc) { fprintf(stderr, "error: %s needs a value\n", what); exit(2); } multi_in = false; return argv[++i]; };
            if (a == "-i" || a == "--input") multi_in = true;
            else if (a == "-k" || a == "--kit") k.kit_name = need("--kit");
            else if (a == "-o" || a == "--output") k.output_folder = need("--output");
            else if (a == "-t" || a == "--threads") k.threads = (size_t)atol(need("--threads"));
            else if (a == "--min-score") k.min_score = atof(need("--min-score"));
            else if (a == "--min-score-diff") k.min_score_diff = atof(need("--min-score-diff"));
            else if (a == "--flank-max-errors") k.max_flank_errors = (size_t)atol(need("--flank-max-errors"));
            else if (a == "--failed-out") k.failed_out = std::string(need("--failed-out"));
            else if (a == "--alpha") k.alpha = (float)atof(need("--alpha"));
            else if (a == "--batch-reads") k.batch_reads = (size_t)atol(need("--batch-reads"));
            else if (a == "--device") k.device = atoi(need("--device"));
            else if (a == "--devices") { k.devices = parse_devices(need("--devices")); }
            else if (a == "--streams") k.streams_per_device = (unsigned)atoi(need("--streams"));
            else if (a == "--counts") k.counts_file = need("--counts");
            else if (a == "--policy") { if (!set_policy(need("--policy"))) return 2; }
            else if (a == "--shard") shard = need("--shard");
            else if (a == "--rccl-id") k.rccl_id = need("--rccl-id");
            else if (a == "--shard-by") { if (!parse_shard_by(need("--shard-by"), k.shard_by_bytes)) return 2; }
            else if (a == "--maximize") { k.maximize = true; multi_in = false; }
            else if (a == "--verbose") { k.verbose = true; multi_in = false; }
            else if (a == "--use-extended") { k.use_extended = true; multi_in = false; }
            else if (a == "--gzip") { k.gzip = true; multi_in = false; }
            else if (a == "--gpu-render") { k.host_cut = false; multi_in = false; }
            else if (!a.empty() && a[0] != '-' && multi_in) input.push_back(a);
            else { fprintf(stderr, "error: unexpected argument '%s'\n", a.c_str()); usage(); return 2; }
        }
        if (k.kit_name.empty() || k.output_folder.empty()) { fputs("error: kit needs --kit and --output\n", stderr); return 2; }
        if (input.empty()) { fputs("error: No FASTQ input files provided\n", stderr); return 2; }
        if (!apply_shard(shard, input, k.shard_rank, k.shard_world, k.shard_by_bytes)) { fputs("error: --shard takes R/W with 0 <= R < W\n", stderr); return 2; }
        if (input.empty() && k.rccl_id.empty()) { puts("Nothing to do for this shard"); return 0; }
        if (input.empty()) { fputs("error: --rccl-id: this shard has no input file, the other shards would wait for it (fewer files than shards)\n", stderr); return 2; }
        if (!k.rccl_id.empty()) shard_rendezvous_reset(k.rccl_id, k.shard_rank);
        try {
            printf("Kit name: %s\nKit type: %s\n", k.kit_name.c_str(), k.maximize ? "Maximize" : "Safe");
            const AnnotateStats st = demux_using_kit(input, k);
            puts("Top 10 most common patterns");
            for (const auto& l : inspect_summary(st, 10)) puts(l.c_str());
            printf("Annotated %zu of %zu reads; filter kept %zu, dropped %zu; trimmed %zu (%zu split, %zu failed)\nDone!\n", st.found, st.total,
                   st.kept, st.dropped, st.trimmed, st.trimmed_split, st.trim_failed);
            if (k.verbose) {   // use_kit.rs:38,73,97 hands --verbose to its three steps: each leaves its log in the output folder
                write_progress_log("annotate", k.output_folder, {{"Total:", st.total}, {"Kept:", st.found}, {"Dropped:", st.total - st.found}});
                write_progress_log("filter", k.output_folder, {{"Total:", st.kept + st.dropped}, {"Kept:", st.kept}, {"Dropped:", st.dropped}});
                write_progress_log("trim", k.output_folder, {{"Total:", st.total}, {"Kept:", st.trimmed}, {"Kept split:", st.trimmed_split}, {"Failed:", st.trim_failed}});
            }
            fprintf(stderr, "Done: %zu records (%.3f s in the pipeline, %.2f M reads/s; histogram summed by %s)\n", st.total, st.seconds_pipeline,
                    st.seconds_pipeline > 0 ? st.total / st.seconds_pipeline / 1e6 : 0.0, st.counts_reduce.c_str());
        } catch (const BarbellError& e) { fprintf(stderr, "error: %s\n", e.what()); return 1; }
        return done_ok();
    }
    if (cmd != "annotate") { usage(); return 2; }
    std::vector<std::string> input, queries, btypes, pattern_files;
    std::string shard;
    bool kit_filter = false, maximize = false;
    TrimConfig tcfg;
    size_t top_n = 10;
    std::string output = "output.tsv", kit;
    AnnotateConfig cfg;
    cfg.process_exits_after = true;   // this process ends when the run does: the page-locked block buffers go back with it
    std::vector<std::string>* multi = nullptr;
    for (int i = 2; i < argc; ++i) {
        const std::string a = argv[i];
        auto need = [&](const char* what) -> const char* { if (i + 1 >= argc) { fprintf(stderr, "error: %s needs a value\n", what); exit(2); } return argv[++i]; };
        if (a == "-i" || a == "--input") { multi = &input; }
        else if (a == "-q" || a == "--queries") { multi = &queries; }
        else if (a == "-b" || a == "--barcode-types") { multi = &btypes; }
        else if (a == "-o" || a == "--output") { output = need("--output"); multi = nullptr; }
        else if (a == "-t" || a == "--threads") { cfg.n_threads = (unsigned)atoi(need("--threads")); multi = nullptr; }
        else if (a == "--kit") { kit = need("--kit"); multi = nullptr; }
        else if (a == "--flank-max-errors") { cfg.max_flank_errors = (size_t)atol(need("--flank-max-errors")); multi = nullptr; }
        else if (a == "--min-score") { cfg.min_score = atof(need("--min-score")); multi = nullptr; }
        else if (a == "--min-score-diff") { cfg.min_score_diff = atof(need("--min-score-diff")); multi = nullptr; }
        else if (a == "--alpha") { cfg.alpha = (float)atof(need("--alpha")); multi = nullptr; }
        else if (a == "--batch-reads") { cfg.batch_reads = (size_t)atol(need("--batch-reads")); multi = nullptr; }
        else if (a == "--block-bytes") { cfg.block_bytes = parse_size(need("--block-bytes"), "--block-bytes"); multi = nullptr; }
        else if (a == "--device") { cfg.device = atoi(need("--device")); multi = nullptr; }
        else if (a == "--devices") { cfg.devices = parse_devices(need("--devices")); multi = nullptr; }
        else if (a == "--streams") { cfg.streams_per_device = (unsigned)atoi(need("--streams")); multi = nullptr; }
        else if (a == "--counts") { cfg.counts_file = need("--counts"); multi = nullptr; }
        else if (a == "--no-compact") { cfg.compact_upload = false; multi = nullptr; }
        else if (a == "--no-pack") { cfg.pack_upload = false; multi = nullptr; }
        else if (a == "--gpu-render") { cfg.host_cut = false; multi = nullptr; }
        else if (a == "--policy") { if (!set_policy(need("--policy"))) return 2; multi = nullptr; }
        else if (a == "--shard") { shard = need("--shard"); multi = nullptr; }
        else if (a == "--rccl-id") { cfg.rccl_id = need("--rccl-id"); multi = nullptr; }
        else if (a == "--shard-by") { if (!parse_shard_by(need("--shard-by"), cfg.shard_by_bytes)) return 2; multi = nullptr; }
        else if (a == "-f" || a == "--filter-file") { multi = &pattern_files; }
        else if (a == "--filtered") { cfg.filtered_file = need("--filtered"); multi = nullptr; }
        else if (a == "--dropped") { cfg.dropped_file = need("--dropped"); multi = nullptr; }
        else if (a == "--kit-filter") { kit_filter = true; multi = nullptr; }
        else if (a == "--trim-output") { cfg.trim_folder = need("--trim-output"); multi = nullptr; }
        else if (a == "--no-label") { tcfg.add_labels = false; multi = nullptr; }
        else if (a == "--no-orientation") { tcfg.add_orientation = false; multi = nullptr; }
        else if (a == "--no-flanks") { tcfg.add_flank = false; multi = nullptr; }
        else if (a == "--sort-labels") { tcfg.sort_labels = true; multi = nullptr; }
        else if (a == "--only-side") {
            const std::string v = need("--only-side"); multi = nullptr;
            if (v == "left") tcfg.only_side = LabelSide::Left;
            else if (v == "right") tcfg.only_side = LabelSide::Right;
            else { fprintf(stderr, "error: --only-side takes left or right\n"); return 2; }
        }
        else if (a == "--failed-out") { tcfg.failed_trimmed_writer = std::string(need("--failed-out")); multi = nullptr; }
        else if (a == "--skip-trim") { tcfg.skip_trim = true; multi = nullptr; }
        else if (a == "--flip") { tcfg.flip = true; multi = nullptr; }
        else if (a == "--gzip") { tcfg.gzip = true; multi = nullptr; }
        else if (a == "--inspect") { cfg.inspect = true; multi = nullptr; }
        else if (a == "--read-pattern-out") { cfg.read_pattern_out = need("--read-pattern-out"); cfg.inspect = true; multi = nullptr; }
        else if (a == "-n" || a == "--top-n") { top_n = (size_t)atol(need("--top-n")); multi = nullptr; }
        else if (a == "-s" || a == "--bucket-size") { cfg.bucket_size = (uint32_t)atol(need("--bucket-size")); multi = nullptr; }
        else if (a == "--maximize") { maximize = true; multi = nullptr; }
        else if (a == "--use-extended") { cfg.use_extended = true; multi = nullptr; }
        else if (a == "--verbose") { cfg.verbose = true; multi = nullptr; }
        else if (a == "-h" || a == "--help") { usage(); return 0; }
        else if (!a.empty() && a[0] != '-' && multi) { multi->push_back(a); }
        else { fprintf(stderr, "error: unexpected argument '%s'\n", a.c_str()); usage(); return 2; }
    }
    if (input.empty()) { fputs("error: No FASTQ input files provided\n", stderr); return 2; }
    if (!apply_shard(shard, input, cfg.shard_rank, cfg.shard_world, cfg.shard_by_bytes)) { fputs("error: --shard takes R/W with 0 <= R < W\n", stderr); return 2; }
    if (input.empty() && cfg.rccl_id.empty()) { fputs("Nothing to do for this shard\n", stderr); return 0; }
    if (input.empty()) { fputs("error: --rccl-id: this shard has no input file, the other shards would wait for it (fewer files than shards)\n", stderr); return 2; }
    if (!cfg.rccl_id.empty()) shard_rendezvous_reset(cfg.rccl_id, cfg.shard_rank);
    if (kit.empty() == queries.empty()) { fputs("error: give either --kit or --queries (they conflict, bin/main.rs:85-87)\n", stderr); return 2; }
    if (kit_filter && kit.empty()) { fputs("error: --kit-filter needs --kit\n", stderr); return 2; }
    if (!cfg.trim_folder.empty()) {
        if (tcfg.sort_labels && tcfg.only_side) { fputs("error: --only-side conflicts with --sort-labels (bin/main.rs:160-161)\n", stderr); return 2; }
        if (!kit_filter && pattern_files.empty()) { fputs("error: --trim-output needs filter patterns (-f or --kit-filter)\n", stderr); return 2; }
        cfg.trim = tcfg;
    }
    if (cfg.bucket_size == 0) { fputs("error: --bucket-size must be positive\n", stderr); return 2; }
    if (kit_filter && !pattern_files.empty()) { fputs("error: give either --filter-file or --kit-filter\n", stderr); return 2; }
    try {
        if (!pattern_files.empty()) cfg.filter_patterns = patterns_from_files(pattern_files);
        if (kit_filter) cfg.filter_patterns = kit_patterns(kit, maximize);
        AnnotateStats st;
        if (!kit.empty()) {
            st = annotate_with_kit(input, output, kit, cfg);
        } else {
            if (btypes.empty()) btypes.push_back("Ftag");  // bin/main.rs:81-82 default
            std::vector<BarcodeType> types;
            for (const auto& b : btypes) {
                if (b == "Ftag") types.push_back(BarcodeType::Ftag);
                else if (b == "Rtag") types.push_back(BarcodeType::Rtag);
                else { fprintf(stderr, "error: unknown barcode type '%s' (Ftag or Rtag)\n", b.c_str()); return 2; }
            }
            st = annotate_with_files(input, queries, types, output, cfg);
        }
        fprintf(stderr, "Done: %zu records, %zu with annotations, %zu rows -> %s (%.3f s in the pipeline, %.2f M reads/s; histogram summed by %s)\n", st.total, st.found,
                st.rows, output.c_str(), st.seconds_pipeline, st.seconds_pipeline > 0 ? st.total / st.seconds_pipeline / 1e6 : 0.0, st.counts_reduce.c_str());
        if (cfg.verbose) write_progress_log("annotate", parent_dir(output), {{"Total:", st.total}, {"Kept:", st.found}, {"Dropped:", st.total - st.found}});   // annotator.rs:259-266, :110-112
        if (!cfg.filter_patterns.empty()) fprintf(stderr, "Filter: %zu kept, %zu dropped\n", st.kept, st.dropped);
        if (cfg.trim) fprintf(stderr, "Trim: %zu trimmed, %zu split, %zu failed\n", st.trimmed, st.trimmed_split, st.trim_failed);
        if (cfg.inspect) for (const auto& l : inspect_summary(st, top_n)) puts(l.c_str());
    } catch (const BarbellError& e) {
        fprintf(stderr, "error: %s\n", e.what());  // the reference prints the anyhow error and exits non-zero (bin/main.rs:301-304)
        return 1;
    }
    return done_ok();
}
