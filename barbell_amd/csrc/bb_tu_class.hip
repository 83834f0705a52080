// bb_tu_class.hip — the register-resident fast barcode kernels for ONE class of traceback orders (policy [H3], bb_prio.h): compiled
// once per class with -DBB_TU_CLASS=0..17 into its own object (Makefile).  The move planes of a class are two v_bitop3 truth tables —
// compile-time constants — so every order the real sassy could turn out to use runs the same kernels at the same speed as the default.
#include "bb_launch.h"
#include "bb_lane.h"
#include "bb_k_bar_pfx.h"

#ifndef BB_TU_CLASS
#error "compile with -DBB_TU_CLASS=<0..17>"
#endif
#define BB_CAT2(a, b) a##b
#define BB_CAT(a, b) BB_CAT2(a, b)

namespace {
constexpr uint32_t PRIO = BB_PRIO_TABLE.cls[BB_TU_CLASS];

template <int CW, bool TAIL, bool NM>
void lane_go(uint32_t blocks, size_t smem, hipStream_t st, const bb_lane_args& a) {
    if (smem > 64 * 1024) (void)hipFuncSetAttribute((const void*)k_barcode_lane<CW, TAIL, PRIO, NM>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    hipLaunchKernelGGL((k_barcode_lane<CW, TAIL, PRIO, NM>), dim3(blocks), dim3(256), smem, st, a.tables, a.groups, a.g, a.strand, a.hits, a.hit_meta, a.list, a.cnt, a.n_hits,
                       a.rows, a.min_score, a.min_score_diff, a.margin, a.fb_lists, a.list_stride, a.fb_cnt);
}
template <int CW>
void lane_pick(bool tail, bool nm, uint32_t blocks, size_t smem, hipStream_t st, const bb_lane_args& a) {
    if (nm) { if (tail) lane_go<CW, true, true>(blocks, smem, st, a); else lane_go<CW, false, true>(blocks, smem, st, a); }
    else { if (tail) lane_go<CW, true, false>(blocks, smem, st, a); else lane_go<CW, false, false>(blocks, smem, st, a); }
}
template <bool TAIL, bool FAST, bool DEFPOL>
void pfx_go(uint32_t blocks, uint32_t threads, size_t smem, hipStream_t st, const bb_pfx_args& a) {
    if (smem > 64 * 1024) (void)hipFuncSetAttribute((const void*)k_barcode_pfx<48, TAIL, FAST, DEFPOL, PRIO>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    hipLaunchKernelGGL((k_barcode_pfx<48, TAIL, FAST, DEFPOL, PRIO>), dim3(blocks), dim3(threads), smem, st, a.tables, a.groups, a.g, a.strand, a.hits, a.pfxs,
                       a.list, a.cnt, a.n_hits, a.hpb, a.min_score, a.min_score_diff, a.rows);
}
}  // namespace

bool BB_CAT(bb_class_lane_, BB_TU_CLASS)(int cw, bool tail, uint32_t blocks, size_t smem, hipStream_t st, const bb_lane_args& a) {
    if (cw == 48) { lane_pick<48>(tail, a.use_nm != 0u, blocks, smem, st, a); return true; }
#if BB_TU_CLASS == 0
    if (cw == 64) { lane_pick<64>(tail, a.use_nm != 0u, blocks, smem, st, a); return true; }
#endif
    return false;
}
// The 48-column k_barcode_pfx of the class: the fast variant (bounds) and the exact one (every lane scored: the hits the bounds leave
// undecided — under a non-default order the run-time form of the move planes made that pass 1.3 x slower).  defpol (the default
// local-minimum and tie rules as compile-time constants, worth 1 %) exists for the default order's fast variant only.
bool BB_CAT(bb_class_pfx_, BB_TU_CLASS)(bool tail, bool fast, bool defpol, uint32_t blocks, uint32_t threads, size_t smem, hipStream_t st, const bb_pfx_args& a) {
    if (!fast) { if (tail) pfx_go<true, false, false>(blocks, threads, smem, st, a); else pfx_go<false, false, false>(blocks, threads, smem, st, a); return true; }
#if BB_TU_CLASS == 0
    if (defpol) { if (tail) pfx_go<true, true, true>(blocks, threads, smem, st, a); else pfx_go<false, true, true>(blocks, threads, smem, st, a); return true; }
#endif
    (void)defpol;
    if (tail) pfx_go<true, true, false>(blocks, threads, smem, st, a); else pfx_go<false, true, false>(blocks, threads, smem, st, a);
    return true;
}
