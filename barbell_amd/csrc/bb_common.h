// bb_common.h — shared host/device definitions of the MI355X annotate hot path.
// Everything here is first-party; nothing is shared with oracle/.
#pragma once
#include <stddef.h>
#include <stdint.h>

#include "../../include/barbell_amd.h"
#include "../../include/barbell_amd_filter.h"
#include "../../include/barbell_amd_policy.h"

#if defined(__HIPCC__)
#define BB_HD __host__ __device__ __forceinline__
#else
#define BB_HD inline
#endif

#define BB_PADDING 10      // src/lib.rs:10
#define BB_MAX_W 8         // flank pattern words (m <= 256); W <= 4 are the tuned instantiations, 5..8 the wide ones
#define BB_MAX_WB 4        // padded barcode pattern words (m_bar <= 128; <= 64 run the tuned kernels, longer ones the any-geometry kernel)
#define BB_MAX_WIN 256     // barcode window columns (<= 64 run the tuned kernels)
#define BB_MAX_FLANK_K 127 // flank error budget
#define BB_MAX_GROUPS 32   // a group is an index in a byte, a bit of a 32-bit launch mask (k_flank_trace) and four list slots of 128 (bb_hit_meta)
#define BB_MAX_OPS (32 * BB_MAX_WB + BB_MAX_WIN)  // unit ops of one barcode alignment

// IUPAC base sets A=1 C=2 G=4 T=8 (case-insensitive, U=T, X = empty, non-letters invalid = 0xFF)
BB_HD uint8_t bb_iupac(uint8_t c) {
    switch (c & 0xDF) {  // fold case for letters
        case 'A': return 1;
        case 'C': return 2;
        case 'G': return 4;
        case 'T': case 'U': return 8;
        case 'R': return 5;
        case 'Y': return 10;
        case 'S': return 6;
        case 'W': return 9;
        case 'K': return 12;
        case 'M': return 3;
        case 'B': return 14;
        case 'D': return 13;
        case 'H': return 11;
        case 'V': return 7;
        case 'N': return 15;
        case 'X': return 0;
        default: return 0xFF;
    }
}
BB_HD bool bb_is_letter(uint8_t c) { uint8_t u = c & 0xDF; return u >= 'A' && u <= 'Z'; }
// code of a read character: invalid characters match nothing
BB_HD uint8_t bb_text_code(uint8_t c) {
    if (!bb_is_letter(c)) return 0;
    uint8_t k = bb_iupac(c);
    return k == 0xFF ? 0 : k;
}
BB_HD uint8_t bb_comp_code(uint8_t k) { return (uint8_t)(((k & 1) << 3) | ((k & 8) >> 3) | ((k & 2) << 1) | ((k & 4) >> 1)); }

// Per-group constants the kernels need (one per query group, in HBM, copied to LDS/SGPRs).
struct bb_group_dev {
    int32_t m;          // flank length (scan pattern)
    int32_t W;          // ceil(m/32)
    int32_t flank_k;
    int32_t bar_lo, bar_hi;   // inclusive
    int32_t m_bar, WB, n_seqs;
    int32_t k1, k2;
    int32_t rel_lo, rel_hi;   // bar_lo - pad_lo, bar_hi - pad_lo (searcher.rs:381-382)
    int32_t type;             // BB_FTAG / BB_RTAG
    int32_t score0;           // D[m][0] of the flank scan (floor(alpha*m))
    int32_t count_off;        // offset of this group's counters in the histogram
    int32_t filt_rows;        // > 0: the flank scan runs as filter (filt_rows <= 15 rows of the flank, both strands in one pass) + windowed verification
    double perfect;
    // byte offsets into the table blob
    uint32_t off_peq_flank[2];   // [strand] 256 entries x PEQ_STRIDE(W) words, indexed by read byte
    uint32_t off_pv0;            // W words: vertical +1 deltas of column 0 (left overhang)
    uint32_t off_ovh;            // (m+1) int32: floor(alpha*o)
    uint32_t off_pcode[2];       // [strand] m bytes: flank base sets (fwd, complemented)
    uint32_t off_peq_bar[2];     // [strand] [16 codes][n_seqs] x WB words
    uint32_t off_lut;            // 256 bytes: read byte -> 4-bit base set (bb_text_code)
    // Row split of the padded barcodes (k_barcode_pfx), per strand (the rc patterns are the reverse complements, so
    // their leading rows are the forward patterns' trailing ones): the first pfx[s] rows and the last tail[s] rows are
    // the same for every barcode of the group, the 32 rows between them fit one word per barcode lane.
    //   rows 1..pfx            computed once per hit, column-wise (k_bar_prefix)
    //   rows pfx+1..pfx+32     one Myers word per (hit, barcode) lane
    //   last tail rows         row-wise (bit-vectors along the window's columns) per lane after the forward pass
    int32_t split[2];            // [strand] 1: hits of this strand take the split kernel
    int32_t pfx[2];              // 0..16
    int32_t tail[2];             // 0..BB_MAX_TAIL
    uint32_t off_peq_pfx[2];     // [strand] 16 words: Peq of the leading shared rows
    uint32_t off_peq_sub[2];     // [strand] [16 codes][n_seqs] words: Peq of rows pfx..pfx+31
    uint32_t off_tail_lut[2];    // [strand] 16 bytes: bit t of byte[code] = trailing row t matches base set `code`
    int32_t filt_off;            // first flank row of the filter's window (rows filt_off .. filt_off + filt_rows - 1)
    int32_t filt_mode;           // BB_FILT_* bits
    int32_t ovh_steps;           // overhang positions worth visiting after the last column: min(m, 1 + max{o : floor(alpha*o) <= k})
    // The context's policy (include/barbell_amd_policy.h), the same in every group; wave-uniform scalars in the kernels.
    int32_t pol_lm;              // [H1] BB_LM_*
    int32_t pol_rc_fwd;          // [H2] 1: rc matches of a read in ascending forward position
    int32_t pol_prio;            // [H3] traceback preference, 2 bits per rank (first choice in bits 0-1); BB_PRIO_DEFAULT = M, I, S, D
    int32_t pol_tie_last;        // [H7] 1: the last of several equally cheap local minima of a barcode pattern
    int32_t pol_lodhi_exp;       // [H8] decay exponents, one byte per op (M, S, I, D); 0x01010101 by default
    int32_t pol_lodhi_p;         // [H8] subsequence length (the register-resident kernels: 3 only)
    int32_t pol_rc_mirror;       // [H5] 1: path pattern indices of rc flank matches are mirrored (get_matching_region selects rows m-1-bar_hi .. m-1-bar_lo)
    double pol_lambda;           // [H8] (the register-resident kernels: 0.5 only)
};
#define BB_PRIO_DEFAULT (BB_OP_MATCH | (BB_OP_INS << 2) | (BB_OP_SUB << 4) | (BB_OP_DEL << 6))
#define BB_LODHI_EXP_DEFAULT 0x01010101
// Which fixed intervals k_flank_verify scans besides the flagged ones (o_max = most rows that can hang over a read end at
// a cost <= k; u, R = the window; see upload_tables):
#define BB_FILT_TRUE_INIT 1u         /* u == 0: the forward block is rows 1..R of the scan's own matrix, overhang column included */
#define BB_FILT_FWD_BEGIN_ALWAYS 2u  /* 0 < u < o_max: a forward match hanging over the read's start may leave the window unflagged */
#define BB_FILT_RC_BEGIN_ALWAYS 4u   /* the same for the rc strand (u < o_max and no hint available: o_max >= R or u > 0) */
#define BB_FILT_RC_BEGIN_HINT 8u     /* u == 0, o_max < R: k_flank_filter says per read whether the rc strand's start needs scanning */
#define BB_FILT_WIDE 32u             /* one word per strand (windows of up to 31 rows) instead of both strands' 15-row blocks in one word */
#define BB_FILT_END_ALWAYS 16u       /* o_max > m - u - R: a match may hang over a strand's end with the window outside the read */

#define BB_MAX_TAIL 4
// per-hit output of k_bar_prefix: what the lanes of k_barcode_pfx need of the shared rows
struct __attribute__((aligned(16))) bb_hit_pfx {
    uint64_t ph, mh;             // bit c: horizontal +1 / -1 delta of row pfx at window column c (carry into row pfx+1)
    uint64_t teq[BB_MAX_TAIL];   // bit c: trailing row t matches the window's column c
    uint32_t sh[64];             // per column: move bit planes of rows 1..pfx, row r <-> bit pfx-r; lo plane | hi plane << 16
};
static_assert(sizeof(bb_hit_pfx) == 304, "bb_hit_pfx: 19 x 16 bytes");

BB_HD int bb_peq_stride_words(int W) { return W <= 2 ? 2 : (W <= 4 ? 4 : 8); }

// unordered flank hit, written by the scan kernel
struct bb_hit_raw {
    uint32_t read_idx;
    uint32_t e;        // scan end index 0..n+m (in scan direction of the strand)
    int16_t cost;
    uint8_t group, strand;
    uint32_t ordinal;  // order of discovery within (read, group, strand)
};
// ordered flank match after traceback + window computation: 32-byte header + the window's base-set
// codes (zero padded), so k_barcode_reg fetches everything it needs about a hit with six 16-byte loads
struct __attribute__((aligned(16))) bb_hit {
    uint32_t read_idx;
    uint32_t text_start, text_end;  // forward coordinates
    uint32_t ws, we;                // barcode window [ws, we)
    int16_t cost;
    uint8_t group, strand;
    uint8_t valid;                  // 0: get_matching_region returned None -> no row
    uint8_t _pad[3];
    uint32_t read_len;
    uint8_t win[64];                // filled when we - ws <= 64
};
static_assert(sizeof(bb_hit) == 96, "bb_hit: 32-byte header + 64 window codes");
// one word per ordered flank match, written next to the record by k_flank_trace: valid (bits 0-7), list slot 4 g + 2 wide + strand of
// k_hit_lists (bits 8-15; wide = window of more than 48 columns), window width min(we - ws, 255) (bits 16-23)
BB_HD uint32_t bb_hit_meta(uint32_t valid, uint32_t group, uint32_t strand, uint32_t wn) {
    return (valid ? 1u : 0u) | ((4u * group + (wn > 48u ? 2u : 0u) + (strand & 1u)) << 8) | ((wn > 255u ? 255u : wn) << 16);
}

// ---- row slots of the barcode stage (one per flank hit) ----
struct __attribute__((aligned(16))) bb_rowtmp {  // one per flank hit: the provisional row; row._pad[0] = 1 when the hit has a row
    bb_row row;
};
static_assert(sizeof(bb_rowtmp) == 48, "bb_rowtmp is three 16-byte pieces");
// What the fast barcode kernel leaves in a hit's row slot for k_rows: the traced path of the barcode with the highest
// score BOUND (column planes, consumed rows) and the second-highest bound.  `marker` sits where bb_row keeps the
// pipeline's row flag (_pad[0], byte 45): 0 = no row, 1 = row, 2 = this record.
struct __attribute__((aligned(16))) bb_winrec {
    unsigned long long plo, phi, diagrow;
    double ub_second;
    uint8_t tstart, best_pos;
    uint16_t top;
    uint8_t flags;
    uint8_t _pad0[8];
    uint8_t marker;
    uint8_t _pad[2];
};
static_assert(sizeof(bb_winrec) == 48 && offsetof(bb_winrec, marker) == 45, "bb_winrec overlays bb_rowtmp");

// ---- filter patterns on the device (k_filter; pattern.rs:96-240) ----
struct bb_pat_elem_dev {
    uint8_t match_type; int8_t orientation; uint8_t relative_to; uint8_t n_cuts;
    int32_t placeholder;
    int64_t lo, hi;
    uint32_t label_off;  // byte offset into the label_ok blob, 0xFFFFFFFF = any label
    bb_cut cuts[BB_MAX_CUTS];
};
struct bb_pat_dev { uint32_t first, n; };

// Kernels over the batch's flank hits take the hit count from the host — or, in a DEFERRED batch (no round trip between the scans and the
// kernels after them: barbell_amd.hip), an upper bound from the host and the count from device memory: blocks beyond the count leave, and so
// does every block if the count exceeds the bound (= the hit buffers' capacity: slots lie beyond them; the host grows them and runs the batch again).
#define BB_HITS_ON_DEVICE(n_hits, n_hits_dev, block_items)                                \
    do {                                                                                  \
        if (n_hits_dev) {                                                                 \
            const uint32_t nd_ = *(n_hits_dev);                                           \
            if (nd_ > (n_hits)) return;                                                   \
            (n_hits) = nd_;                                                               \
        }                                                                                 \
        if ((uint64_t)blockIdx.x * (block_items) >= (n_hits)) return;                     \
    } while (0)
