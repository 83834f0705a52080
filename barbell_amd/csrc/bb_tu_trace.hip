// bb_tu_trace.hip — k_flank_trace (bb_k_trace.h) and its launch: one translation unit of libbarbell_amd.so (bb_launch.h).
#include <algorithm>
#include <cstdlib>

#include "bb_launch.h"
#include "bb_k_trace.h"

// Which variant of k_flank_trace a group takes (bb_kernels.h): 4 = 8-row band in LDS (k <= 3), 2 = 16-row band in LDS (k <= 6), 1 = every row in LDS,
// 3 = checkpointed columns, 0 = private memory.
int bb_trace_mode(const bb_ctx* c, uint32_t g) {
    const bb_group_dev& D = c->gdev[g];
    const int W = D.W;
    const size_t lds = (size_t)(D.m + D.flank_k + 2) * 2 * W * 64 * 4;  // columns 0..m+k, lo+hi, W words, 64 lanes
    const size_t lds_ck = (size_t)(((D.m + D.flank_k) / BB_TRACE_CKB + 1) * 2 * W) * 64 * 4;   // checkpoints only: a block's move bits live in registers
    const size_t lds_band = (size_t)(D.m + D.flank_k + 2) * 64 * 4;
    if (D.flank_k <= 3 && lds_band <= 64 * 1024 && !c->force_generic && !getenv("BARBELL_AMD_TRACE_FULL") && !getenv("BARBELL_AMD_TRACE_BAND16")) return 4;  // 2(k+1) <= 8 rows in 8 bits
    // (measured, round 6: three-word flanks at k <= 6 — the custom dual-end set's — through the checkpointed variant instead: 2.10 -> 1.94 ms per 2 M-read
    // step, but 2.4 -> 4.0 GB of HBM traffic: twelve waves per CU instead of six evict each other's text lines from L2 between the two passes.  Not taken.)
    if (D.flank_k <= 6 && lds_band <= 64 * 1024 && !c->force_generic && !getenv("BARBELL_AMD_TRACE_FULL")) return 2;  // band of 2(k+1)+1 <= 15 rows in 16 bits
    if (W <= 4 && lds <= 64 * 1024 && !c->force_generic) return 1;
    if (lds_ck <= 64 * 1024 && !c->force_generic && !getenv("BARBELL_AMD_TRACE_NOCKPT")) return 3;
    return 0;
}
namespace {
// One launch for every group of the same width and variant (the raw hits of all groups share one array: a launch per
// group walks it once per group with the other groups' lanes idle).
template <int W>
void launch_trace(bb_ctx* c, const uint8_t* d_bases, const uint64_t* d_offsets, uint32_t n_hits, uint32_t gmask, int mode, const uint32_t* n_hits_dev) {
    const size_t lds_rec = (size_t)64 * BB_TRACE_REC_STRIDE * 4;  // the staged hit records share the move bits' LDS
    size_t lds = lds_rec;
    int mk_max = 0;
    for (uint32_t g = 0; g < c->groups.size(); ++g) {
        if (!((gmask >> g) & 1u)) continue;
        const bb_group_dev& D = c->gdev[g];
        const int mk = D.m + D.flank_k;
        mk_max = std::max(mk_max, mk);
        const size_t need = mode == 4 ? (size_t)(mk + 2) * 64 * 2                                   // 16 bits per column and lane
                          : mode == 2 ? (size_t)(mk + 2) * 64 * 4                                   // one word per column and lane
                          : mode == 1 ? (size_t)(mk + 2) * 2 * W * 64 * 4                            // columns 0..m+k, lo+hi, W words
                          : mode == 3 ? (size_t)((mk / BB_TRACE_CKB + 1) * 2 * W) * 64 * 4                         // checkpoints
                          : 0;
        lds = std::max(lds, need);
    }
#define BB_TRACE_ARGS d_bases, d_offsets, (const uint8_t*)c->d_tables, (const bb_group_dev*)c->d_groups, (uint32_t)c->groups.size(), \
                      (const bb_hit_raw*)c->d_raw, n_hits, (const uint32_t*)c->d_base, c->d_hits, c->d_hitmeta, gmask, mk_max, n_hits_dev
    if (mode == 4) hipLaunchKernelGGL((k_flank_trace<W, 4>), dim3((n_hits + 63) / 64), dim3(64), lds, c->stream, BB_TRACE_ARGS);
    else if (mode == 2) hipLaunchKernelGGL((k_flank_trace<W, 2>), dim3((n_hits + 63) / 64), dim3(64), lds, c->stream, BB_TRACE_ARGS);
    else if (mode == 1) {
        if constexpr (W <= 4)  // the full-height LDS variant never fits beyond 4 words
            hipLaunchKernelGGL((k_flank_trace<W, 1>), dim3((n_hits + 63) / 64), dim3(64), lds, c->stream, BB_TRACE_ARGS);
    } else if (mode == 3) hipLaunchKernelGGL((k_flank_trace<W, 3>), dim3((n_hits + 63) / 64), dim3(64), lds, c->stream, BB_TRACE_ARGS);
    else hipLaunchKernelGGL((k_flank_trace<W, 0>), dim3((n_hits + 63) / 64), dim3(64), lds, c->stream, BB_TRACE_ARGS);
#undef BB_TRACE_ARGS
}
}  // namespace

void bb_launch_trace(bb_ctx* c, const uint8_t* d_bases, const uint64_t* d_offsets, uint32_t n_hits, uint32_t gmask, int mode, int W, const uint32_t* n_hits_dev) {
    switch (W) {
        case 1: launch_trace<1>(c, d_bases, d_offsets, n_hits, gmask, mode, n_hits_dev); break;
        case 2: launch_trace<2>(c, d_bases, d_offsets, n_hits, gmask, mode, n_hits_dev); break;
        case 3: launch_trace<3>(c, d_bases, d_offsets, n_hits, gmask, mode, n_hits_dev); break;
        case 4: launch_trace<4>(c, d_bases, d_offsets, n_hits, gmask, mode, n_hits_dev); break;
        case 5: launch_trace<5>(c, d_bases, d_offsets, n_hits, gmask, mode, n_hits_dev); break;
        case 6: launch_trace<6>(c, d_bases, d_offsets, n_hits, gmask, mode, n_hits_dev); break;
        case 7: launch_trace<7>(c, d_bases, d_offsets, n_hits, gmask, mode, n_hits_dev); break;
        default: launch_trace<8>(c, d_bases, d_offsets, n_hits, gmask, mode, n_hits_dev); break;
    }
}
