// bb_annotate.cpp — annotate / annotate_with_* / demux_using_kit of the C++ host (annotator.rs:155-285, use_kit.rs:11-109; round 6: split from
// bb_host.cpp, no behaviour change): one FASTQ stream over the contexts of a run, blocks committed in order, the fused filter / trim / inspect steps.
#include "bb_host.hpp"
#include "bb_feed.hpp"
#include "bb_inflate.hpp"
#include "bb_writers.hpp"

#include <zlib.h>

#include <algorithm>
#include <array>
#include <atomic>
#include <cctype>
#include <cerrno>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstring>
#include <deque>
#include <fstream>
#include <map>
#include <memory>
#include <mutex>
#include <thread>
#include <unordered_map>

#include <dlfcn.h>
#include <fcntl.h>
#include <sched.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

namespace barbell {


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
