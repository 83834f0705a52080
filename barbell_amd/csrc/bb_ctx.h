// bb_ctx.h — the context behind the C ABI's opaque bb_ctx, shared by the translation units that launch kernels on it
// (barbell_amd.hip: the batch pipeline; bb_tu_scan.hip, bb_tu_trace.hip, bb_tu_bar.hip, bb_tu_class.hip: one stage each).
#pragma once
#include <hip/hip_runtime.h>

#include <string>
#include <vector>

#include "../../include/barbell_amd.h"
#include "../../include/barbell_amd_filter.h"
#include "../../include/barbell_amd_inspect.h"
#include "../../include/barbell_amd_synth.h"
#include "bb_common.h"
#include "bb_ctx_view.h"
#include "bb_synth.h"

enum { K_SCAN = 0, K_PREFIX, K_TRACE, K_LISTS, K_BARCODE, K_COLLAPSE, K_EMIT, K_COUNT };
static const char* const kKernelNames[K_COUNT] = {"k_flank_scan", "k_scan_*", "k_flank_trace", "k_hit_lists",
                                           "k_barcode",    "k_collapse", "k_emit"};

struct HostGroup {
    std::vector<std::string> seqs;
    std::string flank;
    std::vector<std::string> pat[2];
    bb_group_info info;
    uint8_t type;
};

// The batch's counters, one allocation with the filter's flag words behind them: ONE memset per batch (eight until round 6 — a small batch's
// host thread spent more time enqueueing them than the device spent on its kernels), and ONE copy back into the page-locked twin when the
// batch ends (hit count, row count, flag counts, list counts came back in four round trips).
struct bb_ctl {
    uint32_t hitcount[4];
    uint32_t total_rows, pad_[3];
    uint32_t vqueue[2 * BB_MAX_GROUPS];            // k_flank_verify's item counters, one pair (strands) per group
    unsigned long long nflag[BB_MAX_GROUPS];       // flagged 16-byte pieces per group (k_flank_filter)
    uint32_t listcnt[4 * BB_MAX_GROUPS];           // hit lists per (group, window class, strand) (k_hit_lists)
    uint32_t fbcnt[4 * BB_MAX_GROUPS];             // ... of the hits the fast barcode kernels' bounds left undecided
};
#define BB_CTL_BYTES 2048u
static_assert(sizeof(bb_ctl) <= BB_CTL_BYTES, "the flag words begin BB_CTL_BYTES behind the control block");

struct bb_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    // k_barcode_lane's waves all take the same time, so a launch ends with a round of waves that fills a fraction of the GPU (13.3 rounds
    // for the forward hits of a 2 M-read batch: 5 % of the kernel).  The rc hits' launch goes to a second stream and fills that tail.
    hipStream_t side = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    bool use_side = false;       // between fork and join of a barcode pass
    // Per (group, strand): k_barcode_lane's walk-free bound decides as often as the traced one while the shared rows match (they are the
    // flank the hit was found with) — on text where they do not, more hits go on to the exact kernel.  Each batch's undecided fraction
    // is read back with the row count; above BARBELL_AMD_LANE_FB_FRAC (0.2) the pair takes k_barcode_pfx for the next 32 batches.
    uint8_t lane_off[BB_MAX_GROUPS][2]{};   // batches left on k_barcode_pfx
    // ... which pays only if that kernel's bounds (the shared rows walked) decide MUCH more: per hit the lane kernel costs ~3.4 ns, k_barcode_pfx ~6.3 ns,
    // the exact pass ~10.3 ns (profiles/r05_fallback_bench.txt) — 3.4 + 10.3 u_lane against 6.3 + 10.3 u_pfx.  The first backed-off batch is a probe:
    // unless it leaves lane_pfx_gain (0.28) of the hits fewer undecided, the pair returns to the lane kernel and stays for 64 batches (round 6: reads
    // full of near-copies of the flank leave both kernels 39 % undecided, and the back-off cost 19.8 -> 27.5 ms per 1 M reads).
    float lane_und_frac[BB_MAX_GROUPS][2]{};
    uint8_t lane_noback[BB_MAX_GROUPS][2]{};
    double lane_pfx_gain = 0.28;            // BARBELL_AMD_LANE_PFX_GAIN
    uint8_t lane_used[BB_MAX_GROUPS][2]{};  // this batch: the pair ran k_barcode_lane
    double lane_fb_frac = 0.2;
    uint64_t last_listed[BB_MAX_GROUPS][2]{}, last_undecided[BB_MAX_GROUPS][2]{};  // of the last batch (bb_last_barcode_stats)
    uint32_t pfx_fast_launches = 0;  // per batch: fast k_barcode_pfx launches (their records need k_rows; k_barcode_lane decides in its final trip)
    bool lazy_prefix = false;    // this batch: every split (group, strand) takes k_barcode_lane, prefix records only for the hits that go on to the exact kernel
    bb_params params{};
    bb_policy policy{};          // include/barbell_amd_policy.h: the switchable assumptions about sassy / cigar-lodhi-rs
    bool generic_barcode = false;  // the policy asks for what only the any-policy barcode kernel (k_barcode) computes: Lodhi p != 3 or lambda != 0.5
    int prio_class = 0;          // class of the policy's traceback order (bb_prio.h): which unit's fast barcode kernels run
    std::vector<HostGroup> groups;
    std::vector<bb_group_dev> gdev;
    bb_group_dev* d_groups = nullptr;
    uint8_t* d_tables = nullptr;
    uint32_t counts_len = 0;
    unsigned long long* d_counts = nullptr;
    // work buffers
    uint64_t cap_m = 0;  // entries of cnt/base (n*G*2+1)
    uint32_t cap_reads = 0, cap_hits = 0;
    uint32_t *d_cnt = nullptr, *d_base = nullptr, *d_sums = nullptr, *d_nrows = nullptr, *d_rowoff = nullptr;
    bb_ctl* d_ctl = nullptr;     // d_hitcount, d_vqueue, d_nflag, d_listcnt, d_fbcnt and d_flags point into this allocation (ensure_ctl)
    bb_ctl* h_ctl = nullptr;     // page-locked: where the batch's numbers come back to
    // A batch of up to defer_max reads runs DEFERRED: nothing between the upload and the rows waits for the device — the hit count stays on the
    // device (launches are sized by the hit buffers' capacity, BB_HITS_ON_DEVICE), every filtered group is verified and its flag count decides for
    // the NEXT batches, and the batch's numbers come back in one copy at the end.  A batch of up to small_pfx_max reads also takes one lane per
    // (hit, barcode) in the barcode stage.  BARBELL_AMD_DEFER_MAX / BARBELL_AMD_SMALL_PFX_MAX (0 = never).
    uint32_t defer_max = 1u << 16, small_pfx_max = 1u << 12;
    bool batch_no_lane = false;  // the batch in hand: bb_takes_lane says no
    // host-pointer form of a deferred batch: the first rows are copied to the caller's buffer before the batch's only wait
    bb_row* spec_dst = nullptr; uint64_t spec_cap = 0, spec_done = 0;
    uint32_t *d_hitcount = nullptr, *d_lists = nullptr, *d_listcnt = nullptr;
    uint32_t *d_fb_lists = nullptr, *d_fbcnt = nullptr;  // hits the fast barcode kernel's bounds left undecided, per (group, strand)
    uint32_t* d_vqueue = nullptr;  // k_flank_verify's item counters (one per strand)
    unsigned long long* d_nflag = nullptr;  // flagged 16-byte pieces of the batch in hand, per group (k_flank_filter)
    double adapt_frac = 0.13;    // BARBELL_AMD_ADAPT_FRAC: flagged fraction of a batch's pieces above which the full scan takes over
    uint64_t last_flagged[BB_MAX_GROUPS]{}, last_pieces[BB_MAX_GROUPS]{};
    uint8_t last_scan_kind[BB_MAX_GROUPS]{};  // 0 full scan, 1 filter + verification, 2 filter, then the full scan (too many flags), 3 full scan while backed off
    // TWIN filter windows (round 6): the right-hand pattern of a dual-end kit is (nearly) the reverse complement of the left-hand one — its filter
    // window on the forward strand is then, row for row, the left-hand window's rc block and vice versa, and its flags are the other group's with the
    // strands swapped.  filt_twin[g] = the group whose filter pass says it all for g (upload_tables: g's window is laid where it mirrors that
    // group's), -1 otherwise; last_twin[g] = the group whose pass g's verification read in the batch in hand (both filtered there), -1: its own.
    int8_t filt_twin[BB_MAX_GROUPS], last_twin[BB_MAX_GROUPS];
    bool filt_twin_swap[BB_MAX_GROUPS]{};   // the twin's window is g's reverse-complemented (strands swapped) / the very same rows (two groups of a kit that share most of their flank)
    bool use_twins = true;   // BARBELL_AMD_FILTER_TWINS=0: every filtered group runs its own pass (tests; the windows stay where they are)
    uint8_t scan_off[BB_MAX_GROUPS]{};        // batches for which the group goes straight to the full scan (set to 16 by a batch of kind 2: its filter pass was wasted)
    uint32_t* d_flags = nullptr; uint64_t cap_flags = 0;  // filtered scan: one bit per 32 text bytes and strand (k_flank_filter)
    // reads of differing lengths (bb_len.h): the batch's segments by falling length, or null for a batch of (nearly) equal reads
    struct bb_lenstat* d_lenstat = nullptr; struct bb_lencur* d_lencur = nullptr; struct bb_lencur* h_lencur = nullptr;   // h_: page-locked host copy
    uint2* d_vtab = nullptr; uint64_t cap_vtab = 0;
    uint32_t* d_vcut = nullptr; uint64_t cap_vcut = 0;         // per vtab entry: its cell among the cut reads' segments
    uint32_t* d_cutread = nullptr; uint64_t cap_cutread = 0;   // per cell: the read
    uint4* d_cutlist = nullptr; uint64_t cap_cutlist = 0;      // per cut read: read, first cell, segments
    uint32_t* d_vcnt = nullptr; uint64_t cap_vcnt = 0;         // hit counts per (cell, group, strand) of the segmented full scan
    uint32_t n_cut_reads = 0, n_cut_segs = 0;
    const uint2* vtab = nullptr;   // of the batch in hand (d_vtab or null)
    uint32_t n_virtual = 0;        // entries of vtab
    uint32_t seg_lines = 32, split_above = 64;   // BARBELL_AMD_SEG_LINES (a multiple of 4; split_above = twice that; 0 = never cut, never sort)
    bool seg_from_env = false;                   // ... given explicitly: no per-batch choice below
    uint32_t batch_seg_lines = 32, batch_split_above = 64;   // of the batch in hand (what the scans' launches pass)
    // A SMALL host batch whose groups all take the filter pass (round 6): a 4 kb read is one lane's 210 us — most of a lone caller's call.  The host
    // form cuts the reads into 512-byte segments itself (it holds the offsets): the segment table rides behind the offsets in the same upload (no
    // kernel, no extra submission), the filter's lanes take 4 lines + 1 of lead-in each.  BARBELL_AMD_SMALL_SEG_MAX reads (0 = never).
    uint32_t small_seg_max = 8192;
    uint64_t* h_offs = nullptr; uint64_t cap_h_offs = 0;    // page-locked: offsets + segment table of such a batch
    const uint2* host_vtab = nullptr; uint32_t host_n_virtual = 0; bool host_vtab_valid = false;
    uint32_t last_min_lines = 0, last_max_lines = 0, last_segments = 0;   // of the last batch (bb_last_length_stats)
    int scan_filter = -1;        // BARBELL_AMD_SCAN_FILTER: 0 never, 1 wherever it is valid (tests), unset: where the prefix says enough
    bool fast_path = true;       // BARBELL_AMD_NO_FAST=1: score every barcode of every hit exactly (the fallback kernel only)
    double fast_margin = 1e-9;   // BARBELL_AMD_FAST_MARGIN: slack of the bound test in k_rows (tests: a huge value sends every hit to the fallback)
    bb_hit_raw* d_raw = nullptr;
    bb_hit* d_hits = nullptr;
    uint32_t* d_hitmeta = nullptr;  // bb_hit_meta per ordered flank match (k_flank_trace -> k_hit_lists, k_barcode_lane)
    bb_hit_pfx* d_pfx = nullptr;  // shared-prefix records of the hits (groups with pfx > 0)
    bb_rowtmp* d_rows = nullptr;
    // staging for the host-pointer variant
    uint8_t* d_in_bases = nullptr; uint64_t cap_in_bases = 0;
    uint64_t* d_in_offsets = nullptr; uint64_t cap_in_offsets = 0;
    bb_row* d_out_rows = nullptr; uint64_t cap_out_rows = 0;
    uint8_t* d_in_packed = nullptr; uint64_t cap_in_packed = 0;      // bb_annotate_batch_packed: the batch two bases per byte, and where its reads begin
    uint64_t* d_in_poffs = nullptr; uint64_t cap_in_poffs = 0;
    // host-pointer form of a batch of up to host_len_max reads: the batch's length statistics (bb_len.h) are taken on the host from the offsets it has
    // in hand instead of in a kernel and a round trip (BARBELL_AMD_HOST_LEN_MAX; 0 = never)
    uint32_t host_len_max = 1u << 17;
    struct bb_lenstat* host_len = nullptr; bool host_len_valid = false;
    // bb_host_phases: wall time of the phases of the host-pointer form, summed over calls (tools/boundary_rate.py)
    bool phases = false; double ph[BB_N_HOST_PHASES]{}; uint64_t ph_calls = 0;
    // filter step
    bb_pat_dev* d_fpats = nullptr;
    bb_pat_elem_dev* d_felems = nullptr;
    uint8_t* d_flabel_ok = nullptr;
    uint32_t* d_flabel_ids = nullptr;
    uint32_t n_fpats = 0;
    bb_row* d_frows = nullptr; uint64_t cap_frows = 0;
    bb_row_verdict* d_fout = nullptr; uint64_t cap_fout = 0;
    bb_inspect_elem* d_iout = nullptr; uint64_t cap_iout = 0;
    // trim step, owned by bb_trim.hip
    bb_trim_state* trim = nullptr;
    // FASTQ ingest, owned by bb_fastq.hip
    bb_fastq_state* fastq = nullptr;
    // TSV renderer, owned by bb_format.hip
    bb_format_state* format = nullptr;
    // synth
    uint8_t* d_synth_table = nullptr;
    bb_synth_params synth{};
    // timing
    bool timing = false;
    bool use_lists = false;  // per-(group, strand class) hit lists in use for the current batch
    int n_cus = 256;
    uint32_t reg_blocks_mult = 1;  // BARBELL_AMD_REG_BLOCKS: persistent blocks per resident slot (tuning knob)
    uint32_t reg_threads = 512;  // BARBELL_AMD_REG_THREADS: block size of k_barcode_reg (tuning knob)
    uint32_t pfx_threads = 0;    // BARBELL_AMD_PFX_THREADS: block size of k_barcode_pfx (0 = as many lanes as fit a CU)
    int lane_kernel = 1;         // BARBELL_AMD_LANE: 1 = the fast barcode stage with one lane per hit (k_barcode_lane) for groups whose flank budget is small
                                 // (its bound assumes the shared rows match, which they do when the flank was found with few edits: at k = 20 eight times
                                 // as many hits go on to the exact kernel), 0 = one lane per (hit, barcode) everywhere (k_barcode_pfx), 2 = one lane per hit everywhere
    bool lane_nm = true;         // BARBELL_AMD_LANE_NM=0: groups with flank budgets above BB_LANE_MAX_FLANK_K take k_barcode_pfx (round 3) instead of k_barcode_lane<.., NM = true>
    bool force_generic = false;  // BARBELL_AMD_GENERIC=1: use the generic (any-geometry) kernels, for tests
    hipEvent_t ev[K_COUNT + 1]{};
    float ms[K_COUNT]{};
    // timing on: every launch of the barcode stage's main kernels between two events on ITS stream, named as rocprofv3 names it, so that the
    // roofline's kernel is a kernel and not a stage (bb_last_dominant_kernel)
    struct LaunchEv { hipEvent_t a = nullptr, b = nullptr; char name[72] = ""; };
    LaunchEv lev[8 * BB_MAX_GROUPS];
    uint32_t n_lev = 0;
    float dom_ms = 0.f; char dom_name[72] = "";
    std::string last_error;
    // blocking waits of the host on the device inside the batch call in hand (stream synchronisations and blocking copies): what a call costs
    // whatever its size (bb_last_host_syncs; tools/boundary_rate.py)
    uint32_t n_syncs = 0, last_syncs = 0;
    int call_depth = 0;
};

// counts the host's blocking waits of one outermost batch call (the host-pointer form calls the device form)
struct bb_call_scope {
    bb_ctx* c;
    explicit bb_call_scope(bb_ctx* ctx) : c(ctx) { if (c->call_depth++ == 0) c->n_syncs = 0; }
    ~bb_call_scope() { if (--c->call_depth == 0) c->last_syncs = c->n_syncs; }
};

#define HIPCHK(ctx, call)                                                                        \
    do {                                                                                         \
        hipError_t e_ = (call);                                                                  \
        if (e_ != hipSuccess) {                                                                  \
            (ctx)->last_error = std::string(#call) + ": " + hipGetErrorString(e_);               \
            return BB_E_HIP;                                                                     \
        }                                                                                        \
    } while (0)

#define BB_SYNC(ctx, stream) do { ++(ctx)->n_syncs; HIPCHK(ctx, hipStreamSynchronize(stream)); } while (0)

#define BB_LDS_MAX (144 * 1024)  // dynamic LDS a block may ask for (160 KB per CU on gfx950, some of it static)

template <typename T>
int grow(bb_ctx* c, T*& p, uint64_t& cap, uint64_t need) {
    if (need <= cap && p) return BB_OK;
    if (p) HIPCHK(c, hipFree(p));
    p = nullptr;
    uint64_t ncap = need + need / 4 + 64;
    HIPCHK(c, hipMalloc((void**)&p, ncap * sizeof(T)));
    cap = ncap;
    return BB_OK;
}

