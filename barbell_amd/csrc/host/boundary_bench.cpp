// bb-boundary-bench — the C ABI driven the way integration/annotator.patch drives it: T worker threads (paraseq's `process_parallel(.., n_threads,
// ..)`, annotator.rs:278-280), each with ITS OWN context (one GpuDemuxer per worker: annotator.rs:88-101 keeps one Demuxer per thread), each
// calling bb_annotate_batch on PAGEABLE host buffers (a Rust Vec<u8> / Vec<u64>) with B reads per call and reading the rows back into its own
// Vec.  Reports reads/s for every (B, T), the per-call fixed cost (intercept of time over B at T = 1) and the host synchronisations per call.
// `--check` annotates the same reads once per B (and T) and prints an order-independent hash of all rows with batch-global read indices: the
// rows of 1 k-read calls from ten threads must be the rows of one big call.
//
// Measurement tool (bench.py's `boundary_step`, tools/boundary_rate.py); no part of the product path.
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "bb_host.hpp"
#include "../../../include/barbell_amd_synth.h"

using namespace barbell;
using Clock = std::chrono::steady_clock;

namespace {
struct Descs {
    std::vector<bb_group_desc> d;
    std::vector<std::vector<const uint8_t*>> ptrs;
    std::vector<std::vector<uint32_t>> lens;
    explicit Descs(const std::vector<BarcodeGroup>& gs) : d(gs.size()), ptrs(gs.size()), lens(gs.size()) {
        for (size_t i = 0; i < gs.size(); ++i) {
            for (const auto& s : gs[i].seqs) { ptrs[i].push_back((const uint8_t*)s.data()); lens[i].push_back((uint32_t)s.size()); }
            d[i].seqs = ptrs[i].data(); d[i].seq_lens = lens[i].data(); d[i].n_seqs = (uint32_t)gs[i].seqs.size();
            d[i].type = gs[i].barcode_type == BarcodeType::Rtag ? BB_RTAG : BB_FTAG;
            d[i].flank_k = gs[i].k_cutoff ? (int32_t)*gs[i].k_cutoff : -1;
        }
    }
};
std::vector<uint64_t> parse_list(const char* v) {
    std::vector<uint64_t> out;
    for (const char* p = v; *p;) { char* e; out.push_back(strtoull(p, &e, 10)); p = *e ? e + 1 : e; }
    return out;
}
inline uint64_t mix(uint64_t x) { x += 0x9E3779B97F4A7C15ull; x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull; x = (x ^ (x >> 27)) * 0x94D049BB133111EBull; return x ^ (x >> 31); }
uint64_t row_hash(bb_row r, uint64_t first_read) {   // by value: the pad bytes are not part of a row
    memset(r._pad, 0, sizeof r._pad);
    uint64_t w[6];
    memcpy(w, &r, 48);
    uint64_t h = mix(first_read + r.read_idx);
    w[0] = (w[0] & 0xFFFFFFFF00000000ull);   // the batch-local read index is replaced by the global one above
    for (uint64_t x : w) h = mix(h ^ x);
    return h;
}
void die(const char* what, int rc, bb_ctx* c) { fprintf(stderr, "bb-boundary-bench: %s: %s %s\n", what, bb_strerror(rc), c ? bb_last_error(c) : ""); exit(1); }
}  // namespace

int main(int argc, char** argv) {
    std::string kit = "SQK-NBD114-96";
    long flank_k = 3;
    uint64_t n_reads = 1u << 20, L = 4000, lmin = 0;
    std::vector<uint64_t> batches{1024, 8192, 65536, 524288}, threads{1, 10};
    double seconds = 2.0;
    bool pinned = false, check = false, extended = false, phases = false;
    int packed = 0;   // 1: the worker packs its batch two bases per byte (bb_pack_bases per read, as the binding does in process_record) and calls bb_annotate_batch_packed; 2: packed once, outside the clock
    int device = 0;
    for (int i = 1; i < argc; ++i) {
        const std::string a = argv[i];
        auto val = [&]() -> const char* { if (i + 1 >= argc) { fprintf(stderr, "%s needs a value\n", a.c_str()); exit(2); } return argv[++i]; };
        if (a == "--kit") kit = val();
        else if (a == "--flank-max-errors") flank_k = atol(val());
        else if (a == "--use-extended") extended = true;
        else if (a == "--reads") n_reads = strtoull(val(), nullptr, 10);
        else if (a == "--read-len") L = strtoull(val(), nullptr, 10);
        else if (a == "--read-len-min") lmin = strtoull(val(), nullptr, 10);   // reads of differing lengths: uniform in [min, read-len]
        else if (a == "--batch") batches = parse_list(val());
        else if (a == "--threads") threads = parse_list(val());
        else if (a == "--seconds") seconds = atof(val());
        else if (a == "--pinned") pinned = true;     // page-locked buffers (bb_host_malloc) instead of a Vec's pages: what the binding COULD do
        else if (a == "--check") check = true;
        else if (a == "--phases") phases = true;
        else if (a == "--packed") packed = 1;
        else if (a == "--prepacked") packed = 2;
        else if (a == "--device") device = atoi(val());
        else { fprintf(stderr, "usage: bb-boundary-bench [--kit K] [--flank-max-errors k] [--use-extended] [--reads N] [--read-len L] [--read-len-min l] [--batch B,B,..] "
                               "[--threads T,T,..] [--seconds s] [--pinned] [--check] [--phases] [--packed | --prepacked] [--device d]\n"); return 2; }
    }
    if (!lmin) lmin = L;
    std::vector<BarcodeGroup> groups;
    try { groups = BarcodeGroup::new_from_kit(kit, extended); } catch (const std::exception& e) { fprintf(stderr, "%s\n", e.what()); return 2; }
    if (flank_k >= 0) for (auto& g : groups) g.set_flank_threshold((size_t)flank_k);
    Descs descs(groups);
    bb_params prm{0.4f, 0.2, 0.1, device};
    uint64_t max_t = 1;
    for (uint64_t t : threads) max_t = std::max(max_t, t);
    std::vector<bb_ctx*> ctx(max_t, nullptr);
    for (auto& c : ctx) { const int rc = bb_create(descs.d.data(), (uint32_t)descs.d.size(), &prm, &c); if (rc != BB_OK) die("bb_create", rc, nullptr); }

    // the reads: synthesised on the device (counter-based), then brought to ordinary host memory
    std::vector<uint64_t> offsets(n_reads + 1);
    int rc = bb_synth_offsets(0xBA7BE11 ^ 2, (uint32_t)lmin, (uint32_t)L, 0, (uint32_t)n_reads, offsets.data());
    if (rc != BB_OK) die("bb_synth_offsets", rc, nullptr);
    const uint64_t n_bytes = offsets[n_reads];
    uint8_t* bases = nullptr;
    std::vector<uint8_t> bases_vec;
    if (pinned) { if ((rc = bb_host_malloc(ctx[0], n_bytes + 16, (void**)&bases)) != BB_OK) die("bb_host_malloc", rc, ctx[0]); }
    else { bases_vec.resize(n_bytes + 16); bases = bases_vec.data(); }
    {
        void *d_b = nullptr, *d_o = nullptr;
        if ((rc = bb_dev_malloc(ctx[0], n_bytes + 16, &d_b)) || (rc = bb_dev_malloc(ctx[0], (n_reads + 1) * 8, &d_o))) die("bb_dev_malloc", rc, ctx[0]);
        if ((rc = bb_dev_upload(ctx[0], d_o, offsets.data(), (n_reads + 1) * 8))) die("upload", rc, ctx[0]);
        if ((rc = bb_synth_reads_dev(ctx[0], 0xBA7BE11 ^ 2, (uint32_t)lmin, (uint32_t)L, 0, (uint32_t)n_reads, (const uint64_t*)d_o, (uint8_t*)d_b))) die("bb_synth_reads_dev", rc, ctx[0]);
        if ((rc = bb_dev_download(ctx[0], bases, d_b, n_bytes))) die("download", rc, ctx[0]);
        bb_dev_free(ctx[0], d_b); bb_dev_free(ctx[0], d_o);
    }

    // --prepacked: every read two bases per byte once, each from a byte of its own
    std::vector<uint8_t> all_packed;
    std::vector<uint64_t> all_poff;
    if (packed == 2) {
        all_poff.resize(n_reads + 1);
        all_poff[0] = 0;
        for (uint64_t i = 0; i < n_reads; ++i) all_poff[i + 1] = all_poff[i] + (offsets[i + 1] - offsets[i] + 1) / 2;
        all_packed.resize(all_poff[n_reads] + 16);
        for (uint64_t i = 0; i < n_reads; ++i) bb_pack_bases(bases + offsets[i], offsets[i + 1] - offsets[i], all_packed.data() + all_poff[i]);
    }

    printf("{\"kit\": \"%s\", \"form\": \"%s\", \"flank_max_errors\": %ld, \"reads\": %llu, \"read_len\": [%llu, %llu], \"host_buffers\": \"%s\", \"runs\": [", kit.c_str(),
           packed == 1 ? "two bases per byte, packed by the worker inside the clock" : packed == 2 ? "two bases per byte, packed beforehand" : "one byte per base", flank_k,
           (unsigned long long)n_reads, (unsigned long long)lmin, (unsigned long long)L, pinned ? "page-locked" : "pageable");
    bool first_run = true;
    for (uint64_t B : batches) {
        if (B > n_reads) B = n_reads;
        const uint64_t n_batches = n_reads / B;   // whole batches only
        for (uint64_t T : threads) {
            std::atomic<uint64_t> next{0}, calls{0}, rows_total{0}, hash{0}, syncs{0};
            std::atomic<bool> stop{false};
            std::atomic<int> err{0};
            // warm-up: every context sees one batch of this size (buffers grown, scan kinds decided)
            auto worker = [&](uint64_t t, bool timed) {
                bb_ctx* c = ctx[t];
                std::vector<bb_row> rows(4 * B + 64);
                std::vector<uint64_t> rel(B + 1), prel(B + 1);
                std::vector<uint8_t> pbuf;
                for (bool once = false; !stop.load(std::memory_order_relaxed) && !(once && !timed); once = true) {
                    const uint64_t i = timed ? next.fetch_add(1) : t;   // warm-up: one batch per context
                    if (check && timed && i >= n_batches) break;
                    const uint64_t b = i % n_batches, f = b * B;
                    for (uint64_t j = 0; j <= B; ++j) rel[j] = offsets[f + j] - offsets[f];   // a batch's own offsets start at 0, as the binding's Vec does
                    uint64_t got = 0;
                    const uint8_t* pk = nullptr;
                    const uint64_t* po = nullptr;
                    if (packed == 1) {
                        pbuf.resize((offsets[f + B] - offsets[f]) / 2 + B + 16);
                        prel[0] = 0;
                        for (uint64_t j = 0; j < B; ++j) prel[j + 1] = prel[j] + bb_pack_bases(bases + offsets[f + j], offsets[f + j + 1] - offsets[f + j], pbuf.data() + prel[j]);
                        pk = pbuf.data(); po = prel.data();
                    } else if (packed == 2) { pk = all_packed.data(); po = all_poff.data() + f; }
                    auto call = [&]() {
                        return packed ? bb_annotate_batch_packed(c, pk, po, rel.data(), (uint32_t)B, rows.data(), rows.size(), &got)
                                      : bb_annotate_batch(c, bases + offsets[f], rel.data(), (uint32_t)B, rows.data(), rows.size(), &got);
                    };
                    int r = call();
                    if (r == BB_E_CAPACITY) { rows.resize(got); r = call(); }
                    if (r != BB_OK) { fprintf(stderr, "bb_annotate_batch: %s %s\n", bb_strerror(r), bb_last_error(c)); err = r; stop = true; break; }
                    if (timed) {
                        calls.fetch_add(1); rows_total.fetch_add(got);
                        syncs.fetch_add((uint64_t)bb_last_host_syncs(c));
                        if (check) { uint64_t h = 0; for (uint64_t k = 0; k < got; ++k) h += row_hash(rows[k], f); hash.fetch_add(h); }
                    }
                }
            };
            {
                std::vector<std::thread> th;
                for (uint64_t t = 0; t < T; ++t) th.emplace_back(worker, t, false);
                for (auto& x : th) x.join();
            }
            if (err) return 1;
            next = 0; stop = false;
            for (uint64_t t = 0; t < T; ++t) bb_host_phases(ctx[t], phases ? 1 : 0, nullptr, nullptr);
            const auto t0 = Clock::now();
            std::vector<std::thread> th;
            for (uint64_t t = 0; t < T; ++t) th.emplace_back(worker, t, true);
            if (!check) {
                while (std::chrono::duration<double>(Clock::now() - t0).count() < seconds && !stop) std::this_thread::sleep_for(std::chrono::milliseconds(2));
                stop = true;
            }
            for (auto& x : th) x.join();
            const double el = std::chrono::duration<double>(Clock::now() - t0).count();
            if (err) return 1;
            const uint64_t nc = calls.load();
            printf("%s{\"batch\": %llu, \"threads\": %llu, \"calls\": %llu, \"seconds\": %.4f, \"reads_per_s\": %.1f, \"ms_per_call\": %.4f, \"rows\": %llu, "
                   "\"host_syncs_per_call\": %.2f",
                   first_run ? "" : ", ", (unsigned long long)B, (unsigned long long)T, (unsigned long long)nc, el, (double)nc * (double)B / el,
                   nc ? el * 1e3 * (double)T / (double)nc : 0.0, (unsigned long long)rows_total.load(), nc ? (double)syncs.load() / (double)nc : 0.0);
            if (phases) {   // mean over the run's calls, all contexts
                double sum[BB_N_HOST_PHASES] = {};
                uint64_t n_calls = 0;
                for (uint64_t t = 0; t < T; ++t) { double ms[BB_N_HOST_PHASES]; uint64_t k = 0; bb_host_phases(ctx[t], 0, ms, &k); n_calls += k; for (int i = 0; i < BB_N_HOST_PHASES; ++i) sum[i] += ms[i]; }
                static const char* const names[BB_N_HOST_PHASES] = {"upload", "pipeline", "rows_back", "lengths", "scans_to_hit_count", "trace_to_row_count", "emit_and_drain"};
                printf(", \"phase_ms_per_call\": {");
                for (int i = 0; i < BB_N_HOST_PHASES; ++i) printf("%s\"%s\": %.4f", i ? ", " : "", names[i], n_calls ? sum[i] / (double)n_calls : 0.0);
                printf("}");
            }
            if (check) printf(", \"rows_hash\": \"%016llx\", \"reads_checked\": %llu", (unsigned long long)hash.load(), (unsigned long long)(n_batches * B));
            printf("}");
            fflush(stdout);
            first_run = false;
        }
    }
    printf("]}\n");
    for (auto& c : ctx) bb_destroy(c);
    return 0;
}
