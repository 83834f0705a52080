// bb_launch.h — what the batch pipeline (barbell_amd.hip) calls in the translation units that hold the kernels of a stage.
// One stage per unit so that they compile side by side (make -j) and a change to one kernel rebuilds one object:
//   bb_tu_scan.hip    k_flank_filter / k_flank_verify / k_flank_scan2 (W = 1..8), the u32 scans
//   bb_tu_trace.hip   k_flank_trace<W, MODE>
//   bb_tu_bar.hip     k_bar_prefix(_list), k_barcode, k_barcode_reg, the run-time-order k_barcode_pfx variants; dispatch of a group's
//                     barcode stage, incl. the per-class fast kernels below
//   bb_tu_class.hip   compiled once per class of traceback orders (-DBB_TU_CLASS=0..17, bb_prio.h): k_barcode_lane<48, TAIL, PRIO, NM> and
//                     the 48-column k_barcode_pfx<48, TAIL, FAST, *, PRIO> (fast and exact); class 0 (the default order) also the 64-column k_barcode_lane
#pragma once
#include "bb_ctx.h"
#include "bb_prio.h"

#define BB_LANE_MAX_FLANK_K 8   // flank edit budget up to which k_barcode_lane's walk-free bound (all P shared rows matched) decides as often as the traced one
                                // (measured: k = 3, 5 yes; k = 20 no); above it the NM instantiation follows the walk per entry column (bb_lane.h)

// timing on: events around one kernel launch on its stream (bb_ctx::lev); bb_launch_timed_end closes the pair opened last
void bb_launch_timed_begin(bb_ctx* c, hipStream_t st, const char* fmt, ...);
void bb_launch_timed_end(bb_ctx* c, hipStream_t st);
int bb_scan_u32(bb_ctx* c, const uint32_t* in, uint32_t* out, uint64_t n, uint32_t* total = nullptr);   // in[n - 1]: place holder; out[n - 1] (and *total) = the sum
// the batch's read lengths (bb_len.h): sets c->vtab / c->n_virtual for the scans of this batch, *off0 / *off1 = offsets[0] / offsets[n]; one round trip
int bb_prepare_lengths(bb_ctx* c, const uint8_t* d_bases, const uint64_t* d_offsets, uint32_t n, uint64_t* off0, uint64_t* off1);
int bb_launch_scans(bb_ctx* c, const uint8_t* d_bases, const uint64_t* d_offsets, uint32_t n, uint64_t flag_words, uint64_t batch_bytes, bool deferred);   // the flank scan of every group
void bb_note_flag_counts(bb_ctx* c, const unsigned long long* nflag);
int bb_trace_mode(const bb_ctx* c, uint32_t g);
void bb_launch_trace(bb_ctx* c, const uint8_t* d_bases, const uint64_t* d_offsets, uint32_t n_hits, uint32_t gmask, int mode, int W, const uint32_t* n_hits_dev);
// bb_fastq.hip: reads that came two bases per byte (bb_pack.h) into the one-byte-per-base batch every kernel takes
void bb_launch_unpack_reads(hipStream_t st, const uint8_t* d_packed, const uint64_t* d_poff, const uint64_t* d_off, uint32_t n, uint8_t* d_bases);
bool bb_takes_lane(const bb_ctx* c, uint32_t g, uint32_t strand, bool wide);      // the batch in hand
bool bb_lane_eligible(const bb_ctx* c, uint32_t g, uint32_t strand, bool wide);   // ... any batch large enough for one lane per hit to pay
void bb_launch_bar_prefix(bb_ctx* c, uint32_t n_hits, hipStream_t st, const uint32_t* n_hits_dev);
void bb_launch_barcode(bb_ctx* c, const uint8_t* d_bases, const uint64_t* d_offsets, uint32_t n_hits, uint32_t g, int pass);

// ---- the per-class units ----
#define BB_LANE_NM_COLS 32   // NM: entry columns with a stored walk mask (the lane's 32 rows consume ~32 of the window's columns first: an entry
                             // further right is rare and gets the sentinel); 16 KB of LDS instead of 2 x CW x 256 keep three blocks per CU
struct bb_lane_args {
    const uint8_t* tables; const bb_group_dev* groups; uint32_t g, strand; const bb_hit* hits; const uint32_t* hit_meta; const uint32_t* list; const uint32_t* cnt;
    uint32_t n_hits; bb_rowtmp* rows; double min_score, min_score_diff, margin; uint32_t* fb_lists; uint32_t list_stride; uint32_t* fb_cnt; uint32_t use_nm;
};
struct bb_pfx_args {
    const uint8_t* tables; const bb_group_dev* groups; uint32_t g, strand; const bb_hit* hits; const bb_hit_pfx* pfxs; const uint32_t* list;
    const uint32_t* cnt; uint32_t n_hits, hpb; double min_score, min_score_diff; bb_rowtmp* rows;
};
// false: this unit has no such instantiation (cw = 64 outside class 0) — the caller takes another kernel
typedef bool (*bb_lane_launch_fn)(int cw, bool tail, uint32_t blocks, size_t smem, hipStream_t st, const bb_lane_args& a);
typedef bool (*bb_pfx_launch_fn)(bool tail, bool fast, bool defpol, uint32_t blocks, uint32_t threads, size_t smem, hipStream_t st, const bb_pfx_args& a);
struct bb_class_unit { bb_lane_launch_fn lane; bb_pfx_launch_fn pfx; };   // pfx: the 48-column k_barcode_pfx, fast and exact
// null members: the class was not built into this library (a development build with fewer classes)
const bb_class_unit& bb_class_unit_of(int cls);
