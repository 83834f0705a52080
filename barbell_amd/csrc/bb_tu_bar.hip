// bb_tu_bar.hip — the barcode stage's dispatch (searcher.rs:267-426) and the kernels that take the traceback order at run time: the prefix
// kernels, k_barcode (any geometry), k_barcode_reg, the exact and 64-column variants of k_barcode_pfx.  The fast 48-column kernels are
// instantiated per class of traceback orders in bb_tu_class.hip (bb_launch.h).
#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>

#include "bb_launch.h"
#include "bb_k_bar_generic.h"
#include "bb_k_bar_prefix.h"
#include "bb_k_bar_pfx.h"

// ---- the per-class units: weak references, so that a development build may link fewer classes (make CLASSES="0 5") ----
#define BB_DECL_CLASS(K)                                                                                                                      \
    extern bool bb_class_lane_##K(int, bool, uint32_t, size_t, hipStream_t, const bb_lane_args&) __attribute__((weak));                        \
    extern bool bb_class_pfx_##K(bool, bool, bool, uint32_t, uint32_t, size_t, hipStream_t, const bb_pfx_args&) __attribute__((weak));
#define BB_FOR_CLASSES(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15) X(16) X(17)
BB_FOR_CLASSES(BB_DECL_CLASS)
static_assert(BB_PRIO_CLASSES == 18, "BB_FOR_CLASSES lists the classes of bb_prio.h");
const bb_class_unit& bb_class_unit_of(int cls) {
#define BB_UNIT_ENTRY(K) {bb_class_lane_##K, bb_class_pfx_##K},
    static const bb_class_unit units[BB_PRIO_CLASSES] = {BB_FOR_CLASSES(BB_UNIT_ENTRY)};
    static const bb_class_unit none = {nullptr, nullptr};
    return cls >= 0 && cls < BB_PRIO_CLASSES ? units[cls] : none;
}

extern "C" uint32_t bb_build_trace_classes(void) {
    uint32_t m = 0;
    for (int i = 0; i < BB_PRIO_CLASSES; ++i) m |= bb_class_unit_of(i).lane ? 1u << i : 0u;
    return m;
}

void bb_launch_timed_begin(bb_ctx* c, hipStream_t st, const char* fmt, ...) {
    if (!c->timing || c->n_lev >= sizeof(c->lev) / sizeof(c->lev[0])) return;
    bb_ctx::LaunchEv& e = c->lev[c->n_lev];
    if (!e.a && (hipEventCreate(&e.a) != hipSuccess || hipEventCreate(&e.b) != hipSuccess)) return;
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(e.name, sizeof e.name, fmt, ap);
    va_end(ap);
    (void)hipEventRecord(e.a, st);
    ++c->n_lev;
}
void bb_launch_timed_end(bb_ctx* c, hipStream_t st) {
    if (c->timing && c->n_lev) (void)hipEventRecord(c->lev[c->n_lev - 1].b, st);
}

// pass 0 of a split (group, strand): k_barcode_lane (which computes the shared rows itself) or k_barcode_pfx (which reads k_bar_prefix's
// records).  wide: the hits whose window exceeds 48 columns — their 64-column k_barcode_lane exists for the default order only.
bool bb_takes_lane(const bb_ctx* c, uint32_t g, uint32_t strand, bool wide) {
    // a small batch: one lane per (hit, barcode) finishes in a fraction of the time one lane per hit takes to walk the group's barcodes
    return !c->batch_no_lane && bb_lane_eligible(c, g, strand, wide);
}
bool bb_lane_eligible(const bb_ctx* c, uint32_t g, uint32_t strand, bool wide) {
    const bb_group_dev& D = c->gdev[g];
    if (!bb_class_unit_of(c->prio_class).lane || (wide && c->prio_class != 0)) return false;
    // any flank budget since round 4: above BB_LANE_MAX_FLANK_K the kernel's bound uses the Match columns of the shared rows' walk (the NM instantiation);
    // BARBELL_AMD_LANE_NM=0 restores the round-3 choice (k_barcode_pfx for those groups)
    const bool big_k = c->groups[g].info.flank_k > BB_LANE_MAX_FLANK_K;
    return c->fast_path && D.pfx[strand] <= 16 &&
           (c->lane_kernel == 2 || (c->lane_kernel == 1 && (!big_k || c->lane_nm) && c->lane_off[g][strand] == 0));
}
void bb_launch_bar_prefix(bb_ctx* c, uint32_t n_hits, hipStream_t st, const uint32_t* n_hits_dev) {
    hipLaunchKernelGGL(k_bar_prefix, dim3((n_hits + 127) / 128), dim3(128), 0, st, (const uint8_t*)c->d_tables, (const bb_group_dev*)c->d_groups,
                       (const bb_hit*)c->d_hits, n_hits, c->d_pfx, (uint32_t)c->groups.size(), n_hits_dev);
}

namespace {
template <int WB, int CW>
void launch_barcode_reg(bb_ctx* c, const uint8_t* d_bases, const uint64_t* d_offsets, uint32_t n_hits, uint32_t g, const uint32_t* list, const uint32_t* cnt) {
    const bb_group_dev& D = c->gdev[g];
    const uint32_t N = (uint32_t)D.n_seqs;
    const uint32_t hpb = c->reg_threads / N;
    const uint32_t threads = ((hpb * N + 63) / 64) * 64;
    const size_t smem = (size_t)2 * 16 * N * WB * 4 + (size_t)hpb * 24 + (size_t)hpb * sizeof(bb_hit) + 16;
    const uint32_t n_iter = (n_hits + hpb - 1) / hpb;
    // persistent blocks: the Peq table is loaded once per block and the next hit records are prefetched
    const uint32_t resident = (uint32_t)c->n_cus * (threads > 256 ? 1u : 2u) * c->reg_blocks_mult;
    const uint32_t blocks = n_iter < resident ? n_iter : resident;
    (void)d_bases; (void)d_offsets;
    hipLaunchKernelGGL((k_barcode_reg<WB, CW>), dim3(blocks), dim3(threads), smem, c->stream,
                       (const uint8_t*)c->d_tables, (const bb_group_dev*)c->d_groups, g, (const bb_hit*)c->d_hits, list,
                       cnt, n_hits, hpb, c->params.min_score, c->params.min_score_diff, c->d_rows);
}

template <int CW>
void launch_barcode_pfx(bb_ctx* c, uint32_t n_hits, uint32_t g, uint32_t strand, const uint32_t* list, const uint32_t* cnt, bool fast) {
    const bb_group_dev& D = c->gdev[g];
    const uint32_t N = (uint32_t)D.n_seqs;
    const bb_class_unit& U = bb_class_unit_of(c->prio_class);
    hipStream_t st = (c->use_side && strand == 1) ? c->side : c->stream;
    if (fast && bb_takes_lane(c, g, strand, CW > 48)) {
        const uint32_t T = (uint32_t)D.tail[strand];
        const uint32_t use_nm = (c->groups[g].info.flank_k > BB_LANE_MAX_FLANK_K && c->lane_nm && D.pfx[strand] > 0) ? 1u : 0u;
        const size_t smem = (((size_t)N * 64 + 31) & ~(size_t)31) + 256 * 32 + 16 + (size_t)T * 2 * 256 * 8 + 2 * 256 * 16 + (use_nm ? (size_t)BB_LANE_NM_COLS * 256 * 2 : 0);
        const bb_lane_args a{(const uint8_t*)c->d_tables, (const bb_group_dev*)c->d_groups, g, strand, (const bb_hit*)c->d_hits, (const uint32_t*)c->d_hitmeta, list, cnt, n_hits, c->d_rows,
                             c->params.min_score, c->params.min_score_diff, c->fast_margin, c->d_fb_lists, c->cap_hits, c->d_fbcnt, use_nm};
        const uint32_t lev0 = c->n_lev;
        bb_launch_timed_begin(c, st, "k_barcode_lane<%d, %s, %uu, %s>", CW, T > 0 ? "true" : "false", BB_PRIO_TABLE.cls[c->prio_class], use_nm ? "true" : "false");
        if (U.lane(CW, T > 0, (n_hits + 255) / 256, smem, st, a)) { bb_launch_timed_end(c, st); c->lane_used[g][strand] = 1; return; }
        c->n_lev = lev0;
    }
    if (fast) ++c->pfx_fast_launches;  // these leave records for k_rows
    // CW = 48 fits 168 VGPRs -> 3 waves per SIMD: 768-thread blocks (8 hits x 96 barcodes use every lane); measured
    // 22.8 vs 26.8 ms against 512-thread blocks at 2 waves per SIMD
    uint32_t tmax = (uint32_t)BB_PFX_MAX_THREADS(CW, fast);
    if (c->pfx_threads && c->pfx_threads <= tmax) tmax = c->pfx_threads;  // BARBELL_AMD_PFX_THREADS (tuning knob)
    uint32_t hpb = std::max(1u, tmax / N);
    uint32_t threads = ((hpb * N + 63) / 64) * 64;
    // two halves of everything a set of hpb hits owns (records, reduction cells, per-column tables) + Peq + trailing-row planes
    auto smem_for = [&](uint32_t h, uint32_t t) {
        return (size_t)2 * h * 40 + 16 + (size_t)2 * h * (sizeof(bb_hit) + sizeof(bb_hit_pfx)) + (size_t)2 * h * CW * 24 + (size_t)16 * N * 4 +
               (size_t)D.tail[strand] * 2 * t * 8 + 64 + (fast ? 256 * 32 + 32 : 0);
    };
    // groups of few barcodes put many hits into a block: fewer of them when the block's LDS would not fit (the per-hit
    // share is ~3.6 KB; 64 KB is what a launch gets without asking, BB_LDS_MAX what the CU has to give)
    while (hpb > 1 && smem_for(hpb, threads) > BB_LDS_MAX) { --hpb; threads = ((hpb * N + 63) / 64) * 64; }
    const size_t smem = smem_for(hpb, threads);
    const uint32_t n_iter = (n_hits + hpb - 1) / hpb;
    const uint32_t per_cu = std::max(1u, (uint32_t)BB_PFX_MAX_THREADS(CW, fast) / threads);  // blocks that fit a CU at this kernel's register count
    const uint32_t resident = (uint32_t)c->n_cus * per_cu * c->reg_blocks_mult;
    const uint32_t blocks = n_iter < resident ? n_iter : resident;
    // the 48-column variants, fast and exact, come from the class's unit (compile-time traceback order); the 64-column kernel reads the
    // order from the group (BB_PRIO_RT)
    if (CW == 48 && U.pfx) {
        const bb_pfx_args a{(const uint8_t*)c->d_tables, (const bb_group_dev*)c->d_groups, g, strand, (const bb_hit*)c->d_hits, (const bb_hit_pfx*)c->d_pfx, list,
                            cnt, n_hits, hpb, c->params.min_score, c->params.min_score_diff, c->d_rows};
        const bool defpol = fast && c->policy.lm_rule == BB_LM_PLATEAU_RIGHT && c->policy.bar_tie == BB_TIE_FIRST;
        const uint32_t lev0 = c->n_lev;   // restored, not decremented: bb_launch_timed_begin may have added nothing (timing off, table full)
        bb_launch_timed_begin(c, st, "k_barcode_pfx<48, %s, %s, %s, %uu>", D.tail[strand] > 0 ? "true" : "false", fast ? "true" : "false",
                              defpol && c->prio_class == 0 ? "true" : "false", BB_PRIO_TABLE.cls[c->prio_class]);
        if (U.pfx(D.tail[strand] > 0, fast, defpol, blocks, threads, smem, st, a)) { bb_launch_timed_end(c, st); return; }
        c->n_lev = lev0;
    }
#define BB_PFX_ARGS (const uint8_t*)c->d_tables, (const bb_group_dev*)c->d_groups, g, strand, (const bb_hit*)c->d_hits, (const bb_hit_pfx*)c->d_pfx, list, \
                    cnt, n_hits, hpb, c->params.min_score, c->params.min_score_diff, c->d_rows
#define BB_PFX_LAUNCH(TAIL_, FAST_)                                                                                          \
    do {                                                                                                                    \
        if (smem > 64 * 1024)                                                                                               \
            (void)hipFuncSetAttribute((const void*)k_barcode_pfx<CW, TAIL_, FAST_, false, BB_PRIO_RT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem); \
        hipLaunchKernelGGL((k_barcode_pfx<CW, TAIL_, FAST_, false, BB_PRIO_RT>), dim3(blocks), dim3(threads), smem, st, BB_PFX_ARGS);   \
    } while (0)
    bb_launch_timed_begin(c, st, "k_barcode_pfx<%d, %s, %s, false, %uu>", CW, D.tail[strand] > 0 ? "true" : "false", fast ? "true" : "false", (unsigned)BB_PRIO_RT);
    if (D.tail[strand] > 0) { if (fast) BB_PFX_LAUNCH(true, true); else BB_PFX_LAUNCH(true, false); }
    else { if (fast) BB_PFX_LAUNCH(false, true); else BB_PFX_LAUNCH(false, false); }
    bb_launch_timed_end(c, st);
#undef BB_PFX_LAUNCH
#undef BB_PFX_ARGS
}

// Barcode stage of one query group: the hits of each strand and window class come from their own list (k_hit_lists,
// slot 4g + 2 wide + strand): windows of at most 48 columns run the 48-column instantiations whatever the widest
// possible window of the group is.
// The kernels index list_cnt with g; handing them list_cnt + (slot - g) makes that the slot's counter.
// pass 0: every hit of the group — split strands through the fast kernel when enabled (bounds + k_rows), the others
// through the exact kernels; pass 1 (after k_rows): the exact split kernel on the hits the bounds left undecided.
template <int WB>
void launch_barcode(bb_ctx* c, const uint8_t* d_bases, const uint64_t* d_offsets, uint32_t n_hits, uint32_t g, int pass) {
    const bb_group_dev& D = c->gdev[g];
    const bb_group_info& I = c->groups[g].info;
    const uint32_t N = (uint32_t)D.n_seqs;
    // widest barcode window the flank traceback can produce: (mask_len - 1 + flank_k) + 2*PADDING
    const uint32_t win_max = I.mask_len + (uint32_t)I.flank_k + 2 * BB_PADDING - 1;
    const size_t peq_bytes = (size_t)2 * 16 * N * WB * 4;
    const bool reg_ok = !c->force_generic && !c->generic_barcode && D.m_bar <= 48 && N <= c->reg_threads && peq_bytes <= 48 * 1024 && win_max <= 63;
    for (uint32_t sw = 0; sw < 4; ++sw) {
        const uint32_t strand = sw & 1u, wide = sw >> 1;
        if (wide && win_max <= 48) continue;  // no such hits
        const uint32_t slot = 4 * g + sw;
        const uint32_t* list = c->d_lists + (size_t)slot * c->cap_hits;
        const uint32_t* cnt = c->d_listcnt + slot - g;  // the kernels index list_cnt with g
        if constexpr (WB == 2) {
            if (!c->force_generic && !c->generic_barcode && D.split[strand] && win_max <= 63) {  // one word per barcode lane (end positions 0..wn live in a 64-bit mask: wn <= 63)
                if (pass == 1) {
                    if (!c->fast_path) continue;
                    list = c->d_fb_lists + (size_t)slot * c->cap_hits;
                    cnt = c->d_fbcnt + slot - g;
                    if (c->lazy_prefix)  // no k_bar_prefix has run over all hits: the records of the undecided ones, now
                        hipLaunchKernelGGL(k_bar_prefix_list, dim3(256), dim3(128), 0, (c->use_side && strand == 1) ? c->side : c->stream, (const uint8_t*)c->d_tables,
                                           (const bb_group_dev*)c->d_groups, (const bb_hit*)c->d_hits, list, c->d_fbcnt + slot, c->d_pfx);
                }
                const bool fast = pass == 0 && c->fast_path;
                if (!wide) launch_barcode_pfx<48>(c, n_hits, g, strand, list, cnt, fast);
                else launch_barcode_pfx<64>(c, n_hits, g, strand, list, cnt, fast);
                continue;
            }
        }
        if (pass == 1) continue;
        if constexpr (WB <= 2) {
            if (reg_ok) {
                if (!wide) launch_barcode_reg<WB, 48>(c, d_bases, d_offsets, n_hits, g, list, cnt);
                else launch_barcode_reg<WB, 64>(c, d_bases, d_offsets, n_hits, g, list, cnt);
                continue;
            }
        }
        const uint32_t hpb = N >= 256 ? 1 : 256 / N;
        const uint32_t threads = ((hpb * N + 63) / 64) * 64;
        const bool lds = peq_bytes <= 48 * 1024;
        const size_t smem = (lds ? peq_bytes : 0) + (size_t)hpb * N * 8 + (size_t)hpb * 16 + (size_t)hpb * BB_MAX_WIN;
        const uint32_t blocks = (n_hits + hpb - 1) / hpb;
        if (lds)
            hipLaunchKernelGGL((k_barcode<WB, true>), dim3(blocks), dim3(threads), smem, c->stream, d_bases, d_offsets,
                               (const uint8_t*)c->d_tables, (const bb_group_dev*)c->d_groups, g, (const bb_hit*)c->d_hits, list, cnt, n_hits, hpb,
                               c->params.min_score, c->params.min_score_diff, c->d_rows);
        else
            hipLaunchKernelGGL((k_barcode<WB, false>), dim3(blocks), dim3(threads), smem, c->stream, d_bases, d_offsets,
                               (const uint8_t*)c->d_tables, (const bb_group_dev*)c->d_groups, g, (const bb_hit*)c->d_hits, list, cnt, n_hits, hpb,
                               c->params.min_score, c->params.min_score_diff, c->d_rows);
    }
}

}  // namespace

void bb_launch_barcode(bb_ctx* c, const uint8_t* d_bases, const uint64_t* d_offsets, uint32_t n_hits, uint32_t g, int pass) {
    switch (c->gdev[g].WB) {
        case 1: launch_barcode<1>(c, d_bases, d_offsets, n_hits, g, pass); break;
        case 2: launch_barcode<2>(c, d_bases, d_offsets, n_hits, g, pass); break;
        case 3: launch_barcode<3>(c, d_bases, d_offsets, n_hits, g, pass); break;
        default: launch_barcode<4>(c, d_bases, d_offsets, n_hits, g, pass); break;
    }
}
