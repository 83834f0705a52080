// bb_myers.h — primitives of the bit-parallel kernels (gfx950 / CDNA4, wave64): three-input booleans, one-instruction 64-bit
// shift / add, the Myers / Hyyro column step on W words, the traceback's move planes, the streaming local-minimum state.
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>
#include <type_traits>

#include "../../include/barbell_amd_filter.h"
#include "../../include/barbell_amd_inspect.h"
#include "bb_common.h"
#include "bb_prio.h"
#include "bb_synth.h"

// ------------------------------------------------------------------------------------------------
// Myers / Hyyro column step on a W-word (32-bit) vertical bit-vector.  Row r (1-based) <-> bit r-1.
// pv/mv: vertical +1/-1 deltas of the previous column, updated in place to the new column.
// d0: diagonal-zero vector, ph/mh: horizontal deltas (before the shift), all for the new column.
// Top boundary row is all zero (text is free: D[0][i] = 0), so the horizontal carry-in is 0.
// ------------------------------------------------------------------------------------------------
// gfx950 three-input boolean: result bit = TT[(a << 2) | (b << 1) | c].  The compiler finds some of these on its
// own but leaves e.g. pv = mhs | ~(d0 | phs) as or + not + or; spelled out they are one instruction each.
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
template <int TT>
__device__ __forceinline__ uint32_t bitop3(uint32_t a, uint32_t b, uint32_t c) { return __builtin_amdgcn_bitop3_b32(a, b, c, TT); }
#define BB_TT_XOR_OR 0xBE    /* (a ^ b) | c   */
#define BB_TT_OR_NOR 0xF1    /* a | ~(b | c)  */

// 64-bit shift by one in ONE instruction (v_lshlrev_b64, half rate like v_lshlrev_b32 / v_alignbit_b32 — measured in
// profiles/valu_ceiling.json — but it does both words); left to itself the compiler splits it into lshl + alignbit
__device__ __forceinline__ unsigned long long shl1_64(unsigned long long x) {
    unsigned long long r;
    asm("v_lshlrev_b64 %0, 1, %1" : "=v"(r) : "v"(x));
    return r;
}
// 64-bit add in ONE instruction without the carry flag (the compiler's v_add_co / v_addc pair needs a wait state between
// its halves on gfx950 and both are half rate)
__device__ __forceinline__ unsigned long long add_64(unsigned long long x, unsigned long long y) {
    unsigned long long r;
    asm("v_lshl_add_u64 %0, %1, 0, %2" : "=v"(r) : "v"(x), "v"(y));
    return r;
}
// x << 1 as x + x: v_add_u32 issues at the full rate, v_lshlrev_b32 at half of it (profiles/valu_ceiling.json); written as inline
// assembly because the compiler canonicalises x + x back into the shift
__device__ __forceinline__ uint32_t shl1_32(uint32_t x) {
    uint32_t r;
    asm("v_add_u32 %0, %1, %1" : "=v"(r) : "v"(x));
    return r;
}
#ifndef BB_MYERS64
#define BB_MYERS64 1  // two-word step: carry chain as one 64-bit add (v_lshl_add_u64), the two shifts as v_lshlrev_b64
#endif
template <int W>
__device__ __forceinline__ void myers_step(uint32_t (&pv)[W], uint32_t (&mv)[W], const uint32_t (&eq)[W],
                                           uint32_t (&d0)[W], uint32_t (&ph)[W], uint32_t (&mh)[W]) {
    if constexpr (W == 2 && BB_MYERS64) {
        const unsigned long long x = ((unsigned long long)(eq[1] & pv[1]) << 32) | (eq[0] & pv[0]);
        const unsigned long long s = add_64(x, ((unsigned long long)pv[1] << 32) | pv[0]);
        d0[0] = bitop3<BB_TT_XOR_OR>((uint32_t)s, pv[0], eq[0]) | mv[0];
        d0[1] = bitop3<BB_TT_XOR_OR>((uint32_t)(s >> 32), pv[1], eq[1]) | mv[1];
        ph[0] = bitop3<BB_TT_OR_NOR>(mv[0], d0[0], pv[0]); ph[1] = bitop3<BB_TT_OR_NOR>(mv[1], d0[1], pv[1]);
        mh[0] = pv[0] & d0[0]; mh[1] = pv[1] & d0[1];
        const unsigned long long phs = shl1_64(((unsigned long long)ph[1] << 32) | ph[0]);
        const unsigned long long mhs = shl1_64(((unsigned long long)mh[1] << 32) | mh[0]);
        pv[0] = bitop3<BB_TT_OR_NOR>((uint32_t)mhs, d0[0], (uint32_t)phs); pv[1] = bitop3<BB_TT_OR_NOR>((uint32_t)(mhs >> 32), d0[1], (uint32_t)(phs >> 32));
        mv[0] = (uint32_t)phs & d0[0]; mv[1] = (uint32_t)(phs >> 32) & d0[1];
        return;
    }
    uint32_t phs[W], mhs[W];
    if constexpr (W >= 3 && W <= 4 && BB_MYERS64) {
        // words in pairs: 64-bit adds, the carry out of a pair from the majority function of its high words' bit 31
        // (x, pv, ~sum), the shifts as 64-bit shifts with one v_alignbit across the pair boundary
        uint32_t carry = 0;
#pragma unroll
        for (int w = 0; w < W; w += 2) {
            if (w + 1 < W) {
                const uint32_t x0 = eq[w] & pv[w], x1 = eq[w + 1] & pv[w + 1];
                unsigned long long sum = add_64(((unsigned long long)x1 << 32) | x0, ((unsigned long long)pv[w + 1] << 32) | pv[w]);
                if (w) sum = add_64(sum, (unsigned long long)carry);  // carry of the pair below (0/1)
                d0[w] = bitop3<BB_TT_XOR_OR>((uint32_t)sum, pv[w], eq[w]) | mv[w];
                d0[w + 1] = bitop3<BB_TT_XOR_OR>((uint32_t)(sum >> 32), pv[w + 1], eq[w + 1]) | mv[w + 1];
                carry = bitop3<0xD4>(x1, pv[w + 1], (uint32_t)(sum >> 32)) >> 31;  // (x & pv) | ((x | pv) & ~sum)
            } else {
                const uint32_t x = eq[w] & pv[w];
                const uint32_t sum = x + pv[w] + carry;
                d0[w] = bitop3<BB_TT_XOR_OR>(sum, pv[w], eq[w]) | mv[w];
            }
        }
#pragma unroll
        for (int w = 0; w < W; ++w) {
            ph[w] = bitop3<BB_TT_OR_NOR>(mv[w], d0[w], pv[w]);
            mh[w] = pv[w] & d0[w];
        }
#pragma unroll
        for (int w = 0; w < W; w += 2) {
            if (w + 1 < W) {
                const unsigned long long p2 = shl1_64(((unsigned long long)ph[w + 1] << 32) | ph[w]);
                const unsigned long long m2 = shl1_64(((unsigned long long)mh[w + 1] << 32) | mh[w]);
                phs[w] = (uint32_t)p2 | (w ? (ph[w - 1] >> 31) : 0u); phs[w + 1] = (uint32_t)(p2 >> 32);
                mhs[w] = (uint32_t)m2 | (w ? (mh[w - 1] >> 31) : 0u); mhs[w + 1] = (uint32_t)(m2 >> 32);
            } else {
                phs[w] = __builtin_amdgcn_alignbit(ph[w], ph[w - 1], 31);
                mhs[w] = __builtin_amdgcn_alignbit(mh[w], mh[w - 1], 31);
            }
        }
    } else {
    uint32_t carry = 0;
#pragma unroll
    for (int w = 0; w < W; ++w) {
        uint32_t x = eq[w] & pv[w];
        uint64_t s = (uint64_t)x + (uint64_t)pv[w] + (uint64_t)carry;
        carry = (uint32_t)(s >> 32);
        d0[w] = bitop3<BB_TT_XOR_OR>((uint32_t)s, pv[w], eq[w]) | mv[w];
    }
#pragma unroll
    for (int w = 0; w < W; ++w) {
        ph[w] = bitop3<BB_TT_OR_NOR>(mv[w], d0[w], pv[w]);
        mh[w] = pv[w] & d0[w];
    }
#pragma unroll
    for (int w = W - 1; w >= 0; --w) {
        phs[w] = (ph[w] << 1) | (w ? (ph[w - 1] >> 31) : 0u);
        mhs[w] = (mh[w] << 1) | (w ? (mh[w - 1] >> 31) : 0u);
    }
    }
#pragma unroll
    for (int w = 0; w < W; ++w) {
        pv[w] = bitop3<BB_TT_OR_NOR>(mhs[w], d0[w], phs[w]);
        mv[w] = phs[w] & d0[w];
    }
}

// Move bits of the traceback preference (oracle [H3]): at cell (row, column) with cost g
//   Match if diagonal-zero and characters match      (d0 & eq)
//   Ins   else if D[j][i-1] == g-1                    (ph)
//   Sub   else if D[j-1][i-1] == g-1                  (~d0)
//   Del   otherwise
// encoded as 2 bits per cell: 0 Match, 1 Sub, 2 Ins, 3 Del  ->  lo = Sub|Del, hi = Ins|Del.
template <int W>
__device__ __forceinline__ void move_bits(const uint32_t (&eq)[W], const uint32_t (&d0)[W], const uint32_t (&ph)[W],
                                          uint32_t (&lo)[W], uint32_t (&hi)[W]) {
#pragma unroll
    for (int w = 0; w < W; ++w) {  // both planes are three-input functions of (d0, eq, ph)
        lo[w] = bitop3<0x15>(d0[w], eq[w], ph[w]);  // ~((d0 & eq) | ph)
        hi[w] = bitop3<0x3A>(d0[w], eq[w], ph[w]);  // (ph & ~(d0 & eq)) | (lo & d0)
    }
}

// The same planes for any preference order (policy [H3]): prio holds the four ops, first choice in bits 0-1; an op is
// applicable at a cell iff  Match: d0 & eq,  Sub: ~d0 (diagonal is g-1),  Ins: ph (left is g-1),  Del: pvn, the NEW
// column's vertical +1 delta (above is g-1).  Used by the kernels that honour every policy (k_flank_trace, k_barcode);
// the default order takes the two-instruction form above.
template <int W>
__device__ __forceinline__ void move_bits_prio(uint32_t prio, const uint32_t (&eq)[W], const uint32_t (&d0)[W], const uint32_t (&ph)[W],
                                               const uint32_t (&pvn)[W], uint32_t (&lo)[W], uint32_t (&hi)[W]) {
    if (prio == (uint32_t)BB_PRIO_DEFAULT) { move_bits<W>(eq, d0, ph, lo, hi); return; }  // wave-uniform
#pragma unroll
    for (int w = 0; w < W; ++w) {
        const uint32_t vM = d0[w] & eq[w], vS = ~d0[w], vI = ph[w], vD = pvn[w];
        uint32_t taken = 0u, sS = 0u, sI = 0u, sD = 0u;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const uint32_t op = (prio >> (2 * q)) & 3u;
            const uint32_t v = (op == BB_OP_MATCH ? vM : op == BB_OP_SUB ? vS : op == BB_OP_INS ? vI : vD) & ~taken;
            taken |= v;
            sS |= op == BB_OP_SUB ? v : 0u; sI |= op == BB_OP_INS ? v : 0u; sD |= op == BB_OP_DEL ? v : 0u;
        }
        lo[w] = sS | sD; hi[w] = sI | sD;
    }
}

// One word's planes for an order that is either a compile-time constant (PRIO = the class's canonical order: one v_bitop3 per
// plane, bb_prio.h) or taken at run time (PRIO = BB_PRIO_RT: the order in `prio`).
template <uint32_t PRIO>
__device__ __forceinline__ void move_planes_any(uint32_t prio, uint32_t d0, uint32_t eq, uint32_t ph, uint32_t pvn, uint32_t& lo, uint32_t& hi) {
    if constexpr (PRIO == BB_PRIO_RT) {
        const uint32_t e1[1] = {eq}, d1[1] = {d0}, p1[1] = {ph}, v1[1] = {pvn};
        uint32_t l1[1], h1[1];
        move_bits_prio<1>(prio, e1, d1, p1, v1, l1, h1);
        lo = l1[0]; hi = h1[0];
    } else {
        move_planes<PRIO>(d0, eq, ph, pvn, lo, hi);
    }
}
template <uint32_t PRIO>
__device__ __forceinline__ void move_planes_any64(uint32_t prio, unsigned long long d0, unsigned long long eq, unsigned long long ph, unsigned long long pvn,
                                                  unsigned long long& lo, unsigned long long& hi) {
    uint32_t l0, h0, l1, h1;
    move_planes_any<PRIO>(prio, (uint32_t)d0, (uint32_t)eq, (uint32_t)ph, (uint32_t)pvn, l0, h0);
    move_planes_any<PRIO>(prio, (uint32_t)(d0 >> 32), (uint32_t)(eq >> 32), (uint32_t)(ph >> 32), (uint32_t)(pvn >> 32), l1, h1);
    lo = ((unsigned long long)l1 << 32) | l0;
    hi = ((unsigned long long)h1 << 32) | h0;
}

template <int W>
__device__ __forceinline__ uint32_t get_bit(const uint32_t (&v)[W], int bit) {
    uint32_t word = v[0];
#pragma unroll
    for (int w = 1; w < W; ++w) word = (bit >> 5) == w ? v[w] : word;
    return (word >> (bit & 31)) & 1u;
}

template <int W, int S>
__device__ __forceinline__ void load_eq(const uint32_t* tab, uint32_t c, uint32_t (&eq)[W]) {
    if constexpr (S == 2) {
        uint2 v = *reinterpret_cast<const uint2*>(tab + c * 2);
        eq[0] = v.x;
        if constexpr (W > 1) eq[1] = v.y;
    } else if constexpr (S == 4) {
        uint4 v = *reinterpret_cast<const uint4*>(tab + c * 4);
        eq[0] = v.x;
        if constexpr (W > 1) eq[1] = v.y;
        if constexpr (W > 2) eq[2] = v.z;
        if constexpr (W > 3) eq[3] = v.w;
    } else {
        const uint4 v = *reinterpret_cast<const uint4*>(tab + c * 8), u = *reinterpret_cast<const uint4*>(tab + c * 8 + 4);
        eq[0] = v.x; eq[1] = v.y; eq[2] = v.z; eq[3] = v.w;
        eq[4] = u.x;
        if constexpr (W > 5) eq[5] = u.y;
        if constexpr (W > 6) eq[6] = u.z;
        if constexpr (W > 7) eq[7] = u.w;
    }
}

// streaming local-minimum rule (policy [H1], include/barbell_amd_policy.h); evaluated lazily: only steps that touch the
// <= k zone matter, and entering the zone from above is a strict decrease, so `dec` and `cand` (the position of the
// last strict decrease: the left end of the plateau in progress) are always fresh when they are read.
struct lm_lane {
    int32_t prev;
    uint32_t dec;
    uint32_t nrep;
    uint32_t cand;
};

__device__ __forceinline__ void emit_hit(bb_hit_raw* hits, uint32_t cap, uint32_t* count, uint32_t read, uint32_t e,
                                         int32_t cost, uint32_t g, uint32_t strand, uint32_t ordinal) {
    uint32_t slot = atomicAdd(count, 1u);
    if (slot < cap) {
        bb_hit_raw h;
        h.read_idx = read; h.e = e; h.cost = (int16_t)cost; h.group = (uint8_t)g; h.strand = (uint8_t)strand; h.ordinal = ordinal;
        hits[slot] = h;
    }
}

