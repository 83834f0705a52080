// bb_prio.h — the traceback preference (policy [H3], include/barbell_amd_policy.h) as COMPILE-TIME move-plane functions.
//
// At a DP cell the ops that can lead to it are known from four bits of the column step:
//     Match  eq        (characters match; eq implies d0 in every step here: d0 = ... | eq)
//     Sub    ~d0       (the diagonal neighbour is one cheaper)
//     Ins    ph        (the left neighbour is one cheaper: horizontal +1)
//     Del    pvn       (the neighbour above is one cheaper: the NEW column's vertical +1)
// and the traceback takes the first applicable op of the policy's order.  The LAST op of an order is never tested (some op
// always applies), so the two move planes (lo = Sub | Del, hi = Ins | Del) are functions of THREE of the four bits whatever
// the order — one v_bitop3 each, with a truth table that is a compile-time constant of the order:
//     last = Del: (d0, eq, ph)      last = Ins: (d0, eq, pvn)      last = Match: (d0, ph, pvn)      last = Sub: (eq, ph, pvn)
// Match and Sub exclude each other, so two orders that differ by swapping ADJACENT M and S take the same op everywhere:
// 24 permutations, 18 distinguishable classes.  A class is named by its canonical order (M before S where adjacent),
// packed like bb_group_dev::pol_prio (first choice in bits 0-1).  The register-resident barcode kernels are instantiated
// per class (k_barcode_lane, the fast k_barcode_pfx); the others take the order at run time (move_bits_prio).
#pragma once
#include <cstdint>

#include "../../include/barbell_amd_policy.h"

#define BB_PRIO_PACK(a, b, c, d) ((uint32_t)(a) | ((uint32_t)(b) << 2) | ((uint32_t)(c) << 4) | ((uint32_t)(d) << 6))

constexpr uint32_t bb_prio_canon(uint32_t prio) {
    uint32_t o[4] = {prio & 3u, (prio >> 2) & 3u, (prio >> 4) & 3u, (prio >> 6) & 3u};
    for (int q = 0; q < 3; ++q)
        if (o[q] == BB_OP_SUB && o[q + 1] == BB_OP_MATCH) { o[q] = BB_OP_MATCH; o[q + 1] = BB_OP_SUB; }
    return BB_PRIO_PACK(o[0], o[1], o[2], o[3]);
}
constexpr bool bb_prio_valid(uint32_t prio) {
    uint32_t seen = 0;
    for (int q = 0; q < 4; ++q) seen |= 1u << ((prio >> (2 * q)) & 3u);
    return seen == 15u && prio < 256u;
}

// the 18 classes, the default (M, I, S, D) first
#define BB_PRIO_CLASSES 18
#define BB_PRIO_RT 0xFFFFFFFFu  /* template argument: the order is a run-time value */
struct bb_prio_table {
    uint32_t cls[BB_PRIO_CLASSES];
    int n;
};
constexpr bb_prio_table bb_prio_build() {
    bb_prio_table t{};
    t.n = 0;
    t.cls[t.n++] = BB_PRIO_PACK(BB_OP_MATCH, BB_OP_INS, BB_OP_SUB, BB_OP_DEL);
    for (uint32_t p = 0; p < 256u; ++p) {
        if (!bb_prio_valid(p) || bb_prio_canon(p) != p) continue;
        bool have = false;
        for (int i = 0; i < t.n; ++i) have = have || t.cls[i] == p;
        if (!have && t.n < BB_PRIO_CLASSES) t.cls[t.n++] = p;
    }
    return t;
}
constexpr bb_prio_table BB_PRIO_TABLE = bb_prio_build();
static_assert(BB_PRIO_TABLE.n == BB_PRIO_CLASSES, "24 orders, 6 pairs that differ by an adjacent M/S swap");
constexpr int bb_prio_class(uint32_t prio) {  // -1: not a permutation
    if (!bb_prio_valid(prio)) return -1;
    const uint32_t c = bb_prio_canon(prio);
    for (int i = 0; i < BB_PRIO_CLASSES; ++i)
        if (BB_PRIO_TABLE.cls[i] == c) return i;
    return -1;
}

// which three bits the planes of an order are functions of (see the head of this file)
enum { BB_PK_D0_EQ_PH = 0, BB_PK_D0_EQ_PVN = 1, BB_PK_D0_PH_PVN = 2, BB_PK_EQ_PH_PVN = 3 };
constexpr int bb_prio_kind(uint32_t prio) {
    const uint32_t last = (prio >> 6) & 3u;
    return last == BB_OP_DEL ? BB_PK_D0_EQ_PH : last == BB_OP_INS ? BB_PK_D0_EQ_PVN : last == BB_OP_MATCH ? BB_PK_D0_PH_PVN : BB_PK_EQ_PH_PVN;
}
// truth table of a plane (0: lo = Sub | Del, 1: hi = Ins | Del) over the kind's inputs (a, b, c), bit index (a << 2) | (b << 1) | c
constexpr uint32_t bb_prio_tt(uint32_t prio, int plane) {
    const int kind = bb_prio_kind(prio);
    uint32_t tt = 0;
    for (uint32_t idx = 0; idx < 8u; ++idx) {
        const bool a = (idx >> 2) & 1u, b = (idx >> 1) & 1u, c = idx & 1u;
        bool ap[4] = {false, false, false, false};  // applicable: indexed by BB_OP_*; the last op of the order is never looked up
        if (kind == BB_PK_D0_EQ_PH) { ap[BB_OP_MATCH] = a && b; ap[BB_OP_SUB] = !a; ap[BB_OP_INS] = c; }
        else if (kind == BB_PK_D0_EQ_PVN) { ap[BB_OP_MATCH] = a && b; ap[BB_OP_SUB] = !a; ap[BB_OP_DEL] = c; }
        else if (kind == BB_PK_D0_PH_PVN) { ap[BB_OP_SUB] = !a; ap[BB_OP_INS] = b; ap[BB_OP_DEL] = c; }
        else { ap[BB_OP_MATCH] = a; ap[BB_OP_INS] = b; ap[BB_OP_DEL] = c; }
        uint32_t op = (prio >> 6) & 3u;
        for (int q = 2; q >= 0; --q) {
            const uint32_t cand = (prio >> (2 * q)) & 3u;
            if (ap[cand]) op = cand;
        }
        const bool bit = plane == 0 ? (op == BB_OP_SUB || op == BB_OP_DEL) : (op == BB_OP_INS || op == BB_OP_DEL);
        tt |= bit ? 1u << idx : 0u;
    }
    return tt;
}
static_assert(bb_prio_tt(BB_PRIO_TABLE.cls[0], 0) == 0x15 && bb_prio_tt(BB_PRIO_TABLE.cls[0], 1) == 0x3A, "the default order's planes (move_bits)");

#ifdef __HIPCC__
// The two move planes of one 32-bit word of a column, for a compile-time order.  pvn is the vertical +1 vector of the column
// just computed (not needed — and not read — when Del is the order's last op, as in the default).
template <uint32_t PRIO>
__device__ __forceinline__ void move_planes(uint32_t d0, uint32_t eq, uint32_t ph, uint32_t pvn, uint32_t& lo, uint32_t& hi) {
    constexpr int K = bb_prio_kind(PRIO);
    constexpr int TL = (int)bb_prio_tt(PRIO, 0), TH = (int)bb_prio_tt(PRIO, 1);
    const uint32_t a = K == BB_PK_EQ_PH_PVN ? eq : d0;
    const uint32_t b = (K == BB_PK_D0_EQ_PH || K == BB_PK_D0_EQ_PVN) ? eq : ph;
    const uint32_t c = K == BB_PK_D0_EQ_PH ? ph : pvn;
    lo = __builtin_amdgcn_bitop3_b32(a, b, c, TL);
    hi = __builtin_amdgcn_bitop3_b32(a, b, c, TH);
}
template <uint32_t PRIO>
__device__ __forceinline__ void move_planes64(unsigned long long d0, unsigned long long eq, unsigned long long ph, unsigned long long pvn,
                                              unsigned long long& lo, unsigned long long& hi) {
    uint32_t l0, h0, l1, h1;
    move_planes<PRIO>((uint32_t)d0, (uint32_t)eq, (uint32_t)ph, (uint32_t)pvn, l0, h0);
    move_planes<PRIO>((uint32_t)(d0 >> 32), (uint32_t)(eq >> 32), (uint32_t)(ph >> 32), (uint32_t)(pvn >> 32), l1, h1);
    lo = ((unsigned long long)l1 << 32) | l0;
    hi = ((unsigned long long)h1 << 32) | h0;
}
constexpr bool bb_prio_needs_pvn(uint32_t prio) { return bb_prio_kind(prio) != BB_PK_D0_EQ_PH; }
#endif
