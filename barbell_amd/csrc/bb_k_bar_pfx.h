// bb_k_bar_pfx.h — the row split with one lane per (flank hit, barcode): k_bar_prefix (shared leading rows once per hit) and
// k_barcode_pfx (fast: score bounds + top-2 per hit; exact: every lane scored).
#pragma once
#include "bb_k_bar_common.h"

// Wave-wide maximum of a u32 on the VALU's data-parallel primitives (no LDS): quad swaps, half-row and row mirrors give
// every lane of a 16-lane row the row's maximum, two row broadcasts carry it to the last row; the result is lane 63's.
__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v) {
    v = max(v, (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0xB1, 0xF, 0xF, false));   // quad_perm [1,0,3,2]
    v = max(v, (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x4E, 0xF, 0xF, false));   // quad_perm [2,3,0,1]
    v = max(v, (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x141, 0xF, 0xF, false));  // row_half_mirror
    v = max(v, (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x140, 0xF, 0xF, false));  // row_mirror
    v = max(v, (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x142, 0xA, 0xF, false));  // row_bcast15 -> rows 1, 3
    v = max(v, (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x143, 0xC, 0xF, false));  // row_bcast31 -> rows 2, 3
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}
// The two largest keys among the wave's lanes with `in` (keys = value bits : 0xFFFF - p with p ascending along the lanes
// of a hit, so the first lane holding the largest value also holds the largest key); 0 where there is none.  Wave-uniform.
__device__ __forceinline__ void wave_top2(bool in, uint32_t vbits, unsigned long long key, unsigned long long& k1, unsigned long long& k2) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t v = in ? vbits + 1u : 0u;  // members are > 0 (value bits are those of a finite non-negative float)
    const uint32_t m1 = wave_max_u32(v);
    k1 = 0ull; k2 = 0ull;
    if (m1 == 0u) return;
    const int l1 = (int)__ffsll((long long)__ballot(v == m1)) - 1;
    k1 = ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(key >> 32), l1) << 32) |
         (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)key, l1);
    const uint32_t v2 = (int)lane == l1 ? 0u : v;
    const uint32_t m2 = wave_max_u32(v2);
    if (m2 == 0u) return;
    const int l2 = (int)__ffsll((long long)__ballot(v2 == m2)) - 1;
    k2 = ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(key >> 32), l2) << 32) |
         (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)key, l2);
}
#define BB_PFX_SYNC() __syncthreads()
// Block size the kernel is compiled for.  The fast 48-column variants fit 168 VGPRs: 768-lane blocks, 3 waves per SIMD (8 hits x 96 barcodes
// use every lane).  The exact variants (every lane scored in f64: the hits the bounds leave undecided, 0.5 % on the headline workload) need
// ~176: compiled for 512-lane blocks they hold everything in registers (8 spilled VGPRs at 768 through round 4).
#define BB_PFX_MAX_THREADS(CW_, FAST_) (((CW_) <= 48 && (FAST_)) ? 768 : 512)
// DEFPOL: the default local-minimum and tie rules as compile-time constants (measured: the run-time form costs the 48-column
// fast variants 1 % — 16.40 against 16.24 ms per 2 M-read step); the host launches it when the context's policy has them
// PRIO: the class of the policy's traceback order (bb_prio.h) as a compile-time constant — the fast variants of the 48-column kernel,
// one instantiation per class —, or BB_PRIO_RT: the order is read from the group (the exact variants and the 64-column kernel).
template <int CW, bool TAIL, bool FAST, bool DEFPOL, uint32_t PRIO>
__global__ __launch_bounds__(BB_PFX_MAX_THREADS(CW, FAST)) void k_barcode_pfx(const uint8_t* __restrict__ tables, const bb_group_dev* __restrict__ groups,
                                                     uint32_t g, uint32_t strand, const bb_hit* __restrict__ hits, const bb_hit_pfx* __restrict__ pfxs,
                                                     const uint32_t* __restrict__ hit_list, const uint32_t* __restrict__ list_cnt,
                                                     uint32_t n_hits_all, uint32_t hpb, double min_score, double min_score_diff,
                                                     bb_rowtmp* __restrict__ rows) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const bb_group_dev G = groups[g];
    const uint32_t n_list = hit_list ? list_cnt[g] : n_hits_all;
    const uint32_t n_iter = (n_list + hpb - 1) / hpb;
    if (blockIdx.x >= n_iter) return;
    const int N = G.n_seqs, m = G.m_bar, P = groups[g].pfx[strand], T = TAIL ? groups[g].tail[strand] : 0;  // scalar loads: no dynamic index into G
    const uint32_t prio_rt = (uint32_t)G.pol_prio;  // read by the BB_PRIO_RT instantiations only
    // rows per lane: 32 = m_bar - P - T (row P+1 <-> bit 31 of the bit-reversed planes, row P+32 <-> bit 0)
    constexpr int PIECES_H = (int)(sizeof(bb_hit) / 16), PIECES_P = (int)(sizeof(bb_hit_pfx) / 16), PIECES = PIECES_H + PIECES_P;
    constexpr int SH_PIECE = PIECES_H + 1 + BB_MAX_TAIL / 2;  // first piece of sh[] inside a hit's record pair
    // LDS carve: [hit + prefix records: hpb x 400 B][max u64[hpb]][second u64[hpb]][cnt1 i32[hpb]][top i32[hpb]][walk table]
    // [peq 16*N words][move planes of the trailing rows: T x 2 x blockDim u64]
    // Everything a set of hpb hits owns exists twice ([2][..]): while the lanes work on one set, the next set's records
    // land in the other half and its per-column tables are built there, so an iteration needs two barriers, not five.
    uint4* s_hit2 = reinterpret_cast<uint4*>(smem);
    size_t o = (size_t)2 * hpb * PIECES * 16;
    unsigned long long* s_max2 = reinterpret_cast<unsigned long long*>(smem + o);
    o += (size_t)2 * hpb * 8;
    unsigned long long* s_sec2 = reinterpret_cast<unsigned long long*>(smem + o);
    o += (size_t)2 * hpb * 8;
    unsigned long long* s_maxB2 = reinterpret_cast<unsigned long long*>(smem + o);  // fast variant: top-2 of the pass-2 candidate set
    o += (size_t)2 * hpb * 8;
    unsigned long long* s_secB2 = reinterpret_cast<unsigned long long*>(smem + o);
    o += (size_t)2 * hpb * 8;
    int32_t* s_cnt12 = reinterpret_cast<int32_t*>(smem + o);
    o += (size_t)2 * hpb * 4;
    int32_t* s_top2 = reinterpret_cast<int32_t*>(smem + o);
    o += (size_t)2 * hpb * 4;
    o = (o + 15) & ~(size_t)15;
    uint2* s_tab2 = reinterpret_cast<uint2*>(smem + o);  // [2][hpb][CW]: the walk through the shared rows per entry column
    o += (size_t)2 * hpb * CW * 8;
    uint4* s_col2 = reinterpret_cast<uint4*>(smem + o);  // [2][hpb][CW]: what every barcode lane of a hit needs of a column
    o += (size_t)2 * hpb * CW * 16;
    uint32_t* s_peq = reinterpret_cast<uint32_t*>(smem + o);
    o += (size_t)16 * N * 4;
    o = (o + 15) & ~(size_t)15;
    o = (o + 31) & ~(size_t)31;
    bb_lb_entry* s_lb = reinterpret_cast<bb_lb_entry*>(smem + o);  // FAST: the bound's table (one entry per byte of Match bits)
    o += FAST ? 256 * sizeof(bb_lb_entry) : 0;
    unsigned long long* s_tail = reinterpret_cast<unsigned long long*>(smem + o);  // [t][lo|hi][thread]
    {
        const uint32_t* gp = reinterpret_cast<const uint32_t*>(tables + groups[g].off_peq_sub[strand]);
        const int words = 16 * N;
        for (int i = threadIdx.x; i < words; i += blockDim.x) s_peq[i] = gp[i];
        if constexpr (FAST)
            for (uint32_t i = threadIdx.x; i < 256u; i += blockDim.x) { bb_lb_entry e; lodhi_bound_table_entry(i, (uint32_t)G.pol_lodhi_exp, e); lb_put(s_lb, i, e); }
    }
    const int hl = threadIdx.x / N;
    const int p = threadIdx.x - hl * N;
    const bool in_blk = hl < (int)hpb;
    const int hls = in_blk ? hl : 0;
    // prefetch of the next iteration's records: lane p of a hit fetches piece p (hit record pieces first, then the
    // prefix record).  Groups with fewer barcodes than pieces (2 N >= PIECES) fetch pieces N.. at the start of the
    // iteration instead, unprefetched — they have many hits per block iteration to hide it behind.
    uint4 pre = make_uint4(0u, 0u, 0u, 0u);
    auto piece = [&](uint32_t idx, int pc) -> uint4 {
        return pc < PIECES_H ? reinterpret_cast<const uint4*>(hits + idx)[pc] : reinterpret_cast<const uint4*>(pfxs + idx)[pc - PIECES_H];
    };
    auto prefetch = [&](uint32_t it) {
        const uint32_t li = it * hpb + (uint32_t)hl;
        if (in_blk && p < PIECES && it < n_iter && li < n_list) pre = piece(hit_list ? hit_list[li] : li, p);
    };
    // set `it` -> half h: lane p of a hit stores piece p of its record pair (prefetched in `pre`)
    auto store_set = [&](uint32_t it, uint32_t h) {
        const uint32_t li = it * hpb + (uint32_t)hl;
        if (in_blk && p < PIECES && it < n_iter && li < n_list) {
            uint4* dst = s_hit2 + ((size_t)h * hpb + hl) * PIECES;
            dst[p] = pre;
            if (p + N < PIECES) dst[p + N] = piece(hit_list ? hit_list[li] : li, p + N);
        }
    };
    // per-column table and reduction cells of the set in half h (all lanes)
    auto build_cols = [&](uint32_t h) {
        const uint4* hitb = s_hit2 + (size_t)h * hpb * PIECES;
        uint4* colb = s_col2 + (size_t)h * hpb * CW;
        if (in_blk && p == 0) {
            const uint32_t x = h * hpb + (uint32_t)hl;
            s_max2[x] = 0ull; s_sec2[x] = 0ull; s_cnt12[x] = 0; s_top2[x] = 0x7FFFFFFF; s_maxB2[x] = 0ull; s_secB2[x] = 0ull;
        }
    // Per (hit, column), once for the hit's N barcode lanes: x = byte offset of the column's base-set row in the Peq table,
    // y / z = carry-in of the shared rows (horizontal +1 / -1 of row P) as words of their own.  The lanes then spend one
    // 16-byte LDS read (a broadcast: the lanes of a hit read the same address) and one addition per column instead of
    // three bit-field extractions and a multiply-add (all half rate, profiles/valu_ceiling.json).
    for (uint32_t l = threadIdx.x; l < hpb * (uint32_t)CW; l += blockDim.x) {
        const uint32_t hw = l / (uint32_t)CW, c = l % (uint32_t)CW;
        const uint32_t* rec = reinterpret_cast<const uint32_t*>(hitb + hw * PIECES);
        const uint32_t code = (rec[8 + (c >> 2)] >> (8u * (c & 3u))) & 0xFu;
        const uint32_t* hv = reinterpret_cast<const uint32_t*>(hitb + hw * PIECES + PIECES_H);  // {ph lo, ph hi, mh lo, mh hi}
        const uint32_t hp = (hv[c >> 5] >> (c & 31u)) & 1u, hm = (hv[2 + (c >> 5)] >> (c & 31u)) & 1u;
        colb[l] = make_uint4(code * (uint32_t)N * 4u, hp, hm, 0u);  // the carry-in bits as words of their own: no extraction per lane
    }
    };
    auto build_walks = [&](uint32_t h) {
        const uint4* hitb = s_hit2 + (size_t)h * hpb * PIECES;
        uint2* tabb = s_tab2 + (size_t)h * hpb * CW;
    // The walk of a traced path through the shared rows depends only on the hit and on the column in which the
    // path enters row P, not on the barcode: the first hpb * CW lanes of the block each walk one (hit, entry column)
    // once — 16 columns from independent LDS reads, static register indices — and every barcode lane later looks
    // its entry up instead of walking (the walk was 12 % of this kernel).  Entry: x = text-op planes of the columns
    // cx, cx-1, .. (bit i <-> column cx - i; lo | hi << 16), y = consumed rows (bits 0..15) | text ops (bits 16..20) |
    // bit position of a cursor still alive after the 16 columns (bits 24..27, flag in bit 31: the lane then finishes
    // in a loop).
    {
        const uint32_t pm = (1u << P) - 1u;  // P <= 16
        for (uint32_t l = threadIdx.x; l < hpb * (uint32_t)CW; l += blockDim.x) {
            const uint32_t hw = l / (uint32_t)CW;
            const int32_t cxw = (int32_t)(l % (uint32_t)CW) + 1;
            const uint32_t* shw = reinterpret_cast<const uint32_t*>(hitb + hw * PIECES + SH_PIECE);
            uint32_t bh = 1u, lo2 = 0u, hi2 = 0u, dgw = 0u, n2 = 0u;
#pragma unroll 1
            for (int i = 0; i < 16 && bh != 0u && cxw - i >= 1; ++i) {  // rolled: short, and the registers are wanted elsewhere
                const uint32_t w = shw[cxw - 1 - i];
                const uint32_t Lr = w & 0xFFFFu, Hr = w >> 16;
                const uint32_t Dr = Lr & Hr;
                const uint32_t nb = ((Dr + bh) & ~Dr) & pm;
                const bool has = nb != 0u, lo = (Lr & nb) != 0u, hi = (Hr & nb) != 0u;
                lo2 |= lo ? (1u << i) : 0u;
                hi2 |= hi ? (1u << i) : 0u;
                const bool consume = has & !hi;
                dgw |= consume ? nb : 0u;
                bh = consume ? ((nb << 1) & pm) : nb;
                n2 += has ? 1u : 0u;
            }
            if (cxw - 16 < 1) bh = 0u;
            tabb[l] = make_uint2(lo2 | (hi2 << 16), dgw | (n2 << 16) | (bh ? 0x80000000u | ((uint32_t)(__ffs(bh) - 1) << 24) : 0u));
        }
    }
    };
    prefetch(blockIdx.x);
    store_set(blockIdx.x, 0u);
    prefetch(blockIdx.x + gridDim.x);
    BB_PFX_SYNC();
    build_cols(0u);
    build_walks(0u);
    BB_PFX_SYNC();
    uint32_t half = 0u;
  for (uint32_t it = blockIdx.x; it < n_iter; it += gridDim.x, half ^= 1u) {
    const uint32_t li = it * hpb + (uint32_t)hl;
    const bool exists = in_blk && li < n_list;
    const uint32_t hit_idx = hit_list ? (exists ? hit_list[li] : 0u) : li;
    // the next set's records go to the other half (its last readers finished before the barrier this wave just left)
    store_set(it + gridDim.x, half ^ 1u);
    prefetch(it + 2u * gridDim.x);
    const uint4* s_hit = s_hit2 + (size_t)half * hpb * PIECES;
    const uint4* s_col = s_col2 + (size_t)half * hpb * CW;
    const uint2* s_tab = s_tab2 + (size_t)half * hpb * CW;
    unsigned long long* s_max = s_max2 + half * hpb, *s_sec = s_sec2 + half * hpb, *s_maxB = s_maxB2 + half * hpb, *s_secB = s_secB2 + half * hpb;
    int32_t* s_cnt1 = s_cnt12 + half * hpb, *s_top = s_top2 + half * hpb;
    bb_hit H;  // header only
    {
        const uint4 h0 = s_hit[hls * PIECES], h1 = s_hit[hls * PIECES + 1];
        H.read_idx = h0.x; H.text_start = h0.y; H.text_end = h0.z; H.ws = h0.w;
        H.we = h1.x; H.cost = (int16_t)(h1.y & 0xFFFFu); H.group = (uint8_t)((h1.y >> 16) & 0xFFu); H.strand = (uint8_t)(h1.y >> 24);
        H.valid = (uint8_t)(h1.z & 0xFFu); H.read_len = h1.w;
    }
    bool active = exists && H.valid != 0;
    if (exists && !H.valid && p == 0) rows[hit_idx].row._pad[0] = 0;
    const int32_t wn = active ? (int32_t)(H.we - H.ws) : 0;
    const uint32_t* s_sh = reinterpret_cast<const uint32_t*>(s_hit + hls * PIECES + SH_PIECE);  // sh[64] of the prefix record

    int wmax = wn;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) wmax = max(wmax, __shfl_xor(wmax, d, 64));
    wmax = __builtin_amdgcn_readfirstlane(wmax);

    // ---- forward pass on the lane's own rows (one word), carry-in from the shared rows ----
    uint32_t L0[CW], H0[CW];
    int32_t best_cost = 0x7FFFFFFF, best_pos = -1;
    {
        const uint4* colv = s_col + hls * CW;
        const uint32_t pb4 = (uint32_t)p * 4u;
        const uint8_t* s_peq_b = reinterpret_cast<const uint8_t*>(s_peq);
        uint32_t pv = 0xFFFFFFFFu, mv = 0u;
        // bottom-row deltas (bit 31 of ph / mh), newest column at bit 0: one v_alignbit per column and plane;
        // the column order is restored after the loop
        uint32_t upr[2] = {0u, 0u}, dnr[2] = {0u, 0u};
#pragma unroll
        for (int c0 = 0; c0 < CW; c0 += BB_CG) {
            if (c0 < BB_FIXED_COLS || c0 < wmax) {  // wave-uniform; the first BB_FIXED_COLS columns unconditionally (straight-line code)
#pragma unroll
                for (int c = c0; c < c0 + BB_CG; ++c) {
                    const uint4 cv = colv[c];
                    const uint32_t eq = *reinterpret_cast<const uint32_t*>(s_peq_b + (cv.x + pb4));
                    const uint32_t hp = cv.y, hm = cv.z;
                    // Every boolean step as ONE three-input v_bitop3 (at three waves per SIMD v_bitop3 issues at 941 G/s, v_and / v_or
                    // at 760: profiles/valu_ceiling.json): 10 v_bitop3 + 1 add + 2 v_bfrev + 2 v_lshlrev_b64 per column (was 8 + 5 + 2 + 2)
                    const uint32_t x = bitop3<0xC8>(eq, pv, hm);                       // (eq | hm) & pv
                    const uint32_t t = bitop3<BB_TT_XOR_OR>(x + pv, pv, eq);           // ((x + pv) ^ pv) | eq
                    const uint32_t d0 = bitop3<0xFE>(t, hm, mv);                       // t | hm | mv
                    const uint32_t ph = bitop3<BB_TT_OR_NOR>(mv, d0, pv), mh = bitop3<0xC0>(pv, d0, 0u);   // mv | ~(d0 | pv),  pv & d0
                    uint32_t l, hh;  // move planes (bb_prio.h; the default order: two three-input functions of (d0, eq, ph))
                    if constexpr (PRIO != BB_PRIO_RT && !bb_prio_needs_pvn(PRIO)) move_planes<PRIO>(d0, eq, ph, 0u, l, hh);
                    // {accumulator : vector} shifted as ONE 64-bit value: the vector's top bit (the bottom row's delta) lands in
                    // the accumulator and the vector is shifted, in one half-rate instruction instead of v_alignbit + v_lshl_or
                    const unsigned long long tp = shl1_64(((unsigned long long)upr[c >> 5] << 32) | ph);
                    const unsigned long long tm = shl1_64(((unsigned long long)dnr[c >> 5] << 32) | mh);
                    upr[c >> 5] = (uint32_t)(tp >> 32); dnr[c >> 5] = (uint32_t)(tm >> 32);
                    // with phs = (ph << 1) | hp and mhs = (mh << 1) | hm (carry-in of the shared rows in bit 0):
                    const uint32_t nph = bitop3<0x01>((uint32_t)tp, hp, d0);           // ~(phs | d0)
                    mv = bitop3<0xA8>((uint32_t)tp, hp, d0);                           // phs & d0
                    pv = bitop3<0xFE>(nph, (uint32_t)tm, hm);                          // mhs | ~(d0 | phs)
                    if constexpr (PRIO == BB_PRIO_RT || bb_prio_needs_pvn(PRIO)) move_planes_any<PRIO>(prio_rt, d0, eq, ph, pv, l, hh);  // Del's bit: the new column's vertical +1
                    L0[c] = __brev(l); H0[c] = __brev(hh);  // row P+1 <-> bit 31, row P+32 <-> bit 0
                }
            }
        }
        // columns processed (wave-uniform): the groups below wmax; word w holds its columns newest-first
        const int pc = min(CW, ((max(wmax, BB_FIXED_COLS) + BB_CG - 1) / BB_CG) * BB_CG);
        const int n0 = min(pc, 32), n1 = pc - n0;
        uint32_t up[2], dn[2];
        up[0] = n0 ? __brev(upr[0]) >> (32 - n0) : 0u; dn[0] = n0 ? __brev(dnr[0]) >> (32 - n0) : 0u;
        up[1] = n1 ? __brev(upr[1]) >> (32 - n1) : 0u; dn[1] = n1 ? __brev(dnr[1]) >> (32 - n1) : 0u;
        const unsigned long long wmask = wn >= 64 ? ~0ull : ((1ull << wn) - 1ull);
        unsigned long long Pm = (((unsigned long long)up[1] << 32) | up[0]) & wmask;   // horizontal deltas of row P+32
        unsigned long long Mm = (((unsigned long long)dn[1] << 32) | dn[0]) & wmask;
        // The trailing shared rows, row-wise: the same recurrence with the roles of rows and columns exchanged — bit-vectors
        // run along the window's columns, the state is the horizontal deltas of the row above, the carry-in is the vertical
        // delta +1 of column 0 (D[r][0] = r), Eq comes from the prefix record (the rows' characters are the same for every
        // barcode).  ~20 64-bit operations per row and lane instead of a second word in every column step.  The rows' move
        // planes (as column masks) are parked in LDS for the start of the traceback.
        if constexpr (TAIL) {
            const uint2* teq = reinterpret_cast<const uint2*>(s_hit + hls * PIECES + PIECES_H + 1);
#pragma unroll 1
            for (int t = 0; t < T; ++t) {
                const uint2 e2 = teq[t];
                const unsigned long long Eq = ((unsigned long long)e2.y << 32) | e2.x;
                const unsigned long long D0 = (((Eq & Pm) + Pm) ^ Pm) | Eq | Mm;
                const unsigned long long Pvv = Mm | ~(D0 | Pm), Mvv = Pm & D0;
                const unsigned long long Pvs = (Pvv << 1) | 1ull, Mvs = Mvv << 1;
                const unsigned long long Ph = Mvs | ~(D0 | Pvs), Mh = Pvs & D0;
                unsigned long long tl, th;  // row-wise: Ins tests the new row's horizontal +1 (Ph), Del the vertical +1 between the two rows (Pvv)
                move_planes_any64<PRIO>(prio_rt, D0, Eq, Ph, Pvv, tl, th);
                s_tail[(size_t)(2 * t) * blockDim.x + threadIdx.x] = tl;
                s_tail[(size_t)(2 * t + 1) * blockDim.x + threadIdx.x] = th;
                Pm = Ph & wmask; Mm = Mh & wmask;
            }
        }
        pick_minimum(Pm, Mm, wn, m, active, DEFPOL ? BB_LM_PLATEAU_RIGHT : G.pol_lm, DEFPOL ? false : G.pol_tie_last != 0, best_cost, best_pos);
        if constexpr (!FAST) { if (active && best_pos >= 0 && best_cost <= G.k1) atomicAdd(&s_cnt1[hl], 1); }
    }
    bool cand = active && best_pos >= 0 && best_cost <= G.k2;
    // ---- traceback, phase 1: the lane's own rows, one-hot cursor on one 32-bit word (row P+1 <-> bit 31).
    // The cursor leaves the word either by the carry of the Del-run addition (the run continues in the shared
    // rows at the same column, which then has no text op in this phase) or by a Match/Sub out of row P+1 (next
    // column); either way phase 2 starts with the cursor entering row P. ----
    unsigned long long plo = 0ull, phi = 0ull;
    uint32_t b = 0u, dg = 0u;
    // ---- phase 0: the trailing shared rows, from (row m, column best_pos) on their column masks: one step per loop
    // iteration (Match/Sub: row and column, Ins: column, Del: row) until the cursor reaches row P+32 — typically T
    // iterations.  Rows left over when the window's first column is passed are deleted, like everything above them. ----
    int32_t c_ent = best_pos;   // column in which the cursor enters the lane's word
    int32_t tr = cand ? T - 1 : -1;
    uint32_t dgt = 0u;          // trailing rows consumed by a Match/Sub
    while (TAIL && __any(tr >= 0 && c_ent >= 1)) {
        const bool on = tr >= 0 && c_ent >= 1;
        const int rr = on ? tr : 0, sh = on ? c_ent - 1 : 0;
        const unsigned long long l64 = s_tail[(size_t)(2 * rr) * blockDim.x + threadIdx.x], h64 = s_tail[(size_t)(2 * rr + 1) * blockDim.x + threadIdx.x];
        const uint32_t lo = (uint32_t)(l64 >> sh) & 1u, hi = (uint32_t)(h64 >> sh) & 1u;
        const bool del = on && (lo & hi) != 0u, text = on && !del, diag = on && hi == 0u;
        plo |= text ? (unsigned long long)lo << sh : 0ull;
        phi |= text ? (unsigned long long)hi << sh : 0ull;
        dgt |= diag ? 1u << rr : 0u;
        tr -= (del || diag) ? 1 : 0;
        c_ent -= text ? 1 : 0;
    }
    // the cursor enters at row P+32 = bit 0 of the word in column c_ent: that column's bit of this mask is simply
    // added in with the Del-run sum (v_add3)
    const unsigned long long smask = (cand && tr < 0 && c_ent >= 1) ? 1ull << (c_ent - 1) : 0ull;  // column 0: nothing to walk
    const uint32_t sm_w[2] = {(uint32_t)smask, (uint32_t)(smask >> 32)};
    // Mask arithmetic only (profiles/valu_ceiling.json: v_cmp / v_cndmask / shifts issue at half the rate of and/or/add):
    // nb is one-hot or zero, so "the landing cell has lo" is (Lr & nb) != 0 — brought to bit 31 by negation and shifted
    // into the column accumulators with one v_alignbit per plane (word 1: columns 33.., word 0: columns 1..32, newest
    // column at bit 0 = its final place); a Match/Sub step is cm = nb & ~Hr (one-hot or zero): consumed rows |= cm,
    // and the cursor moves by b = nb + cm (nb << 1 when it consumed, nb when it did not).
    uint32_t pl_acc[2] = {0u, 0u}, ph_acc[2] = {0u, 0u};
#pragma unroll
    for (int c0 = CW; c0 >= BB_CG; c0 -= BB_CG) {
        if (c0 <= BB_FIXED_COLS || c0 - (BB_CG - 1) <= wmax) {  // wave-uniform
#pragma unroll
            for (int c = c0; c > c0 - BB_CG; --c) {
                const uint32_t Lr = L0[c - 1], Hr = H0[c - 1];
                const uint32_t Dr = Lr & Hr;
                const uint32_t nb = bitop3<0x0C>(Dr, Dr + b + ((sm_w[(c - 1) >> 5] >> ((c - 1) & 31)) & 1u), 0u);  // ~Dr & sum
                const uint32_t tl = Lr & nb, th = Hr & nb;
                const uint32_t cm = bitop3<0x0C>(Hr, nb, 0u);  // ~Hr & nb
                pl_acc[(c - 1) >> 5] = __builtin_amdgcn_alignbit(pl_acc[(c - 1) >> 5], 0u - tl, 31);
                ph_acc[(c - 1) >> 5] = __builtin_amdgcn_alignbit(ph_acc[(c - 1) >> 5], 0u - th, 31);
                dg |= cm;
                b = nb + cm;
            }
        }
    }
    plo |= ((unsigned long long)pl_acc[1] << 32) | pl_acc[0];
    phi |= ((unsigned long long)ph_acc[1] << 32) | ph_acc[0];
    // Text ops of phase 1 = rows it consumed + its Ins columns.  Whichever way the cursor left the word, the
    // columns best_pos .. cx+1 carry exactly those ops: phase 2 starts at column cx = best_pos - ntext.
    int32_t ntext = cand ? __popc(dg) + __popc(dgt) + __popcll(phi & ~plo) : 0;
    const int32_t cx = cand ? best_pos - ntext : 0;
    // ---- phase 2: the shared rows (row r <-> bit P - r): looked up in the block's walk table; a cursor still
    // alive after the table's 16 columns (more than 16 - P insertions inside the shared rows) finishes in the
    // loop underneath on the move bits of the hit's prefix record. ----
    uint32_t dgh = 0u;
    BB_PFX_SYNC();  // barrier A: every wave is done with the previous set; the next set's records are in place
    {
        const uint32_t pm = (1u << P) - 1u;
        const uint2 e = (cand && cx >= 1) ? s_tab[hls * CW + cx - 1] : make_uint2(0u, 0u);
        const uint32_t lo2 = e.x & 0xFFFFu, hi2 = e.x >> 16;
        dgh = e.y & 0xFFFFu;
        uint32_t bh = (e.y >> 31) ? 1u << ((e.y >> 24) & 0xFu) : 0u;
        ntext += (int32_t)((e.y >> 16) & 0x1Fu);
        // local bit i <-> column cx - i <-> plane bit cx - i - 1: reverse the 16 bits and slide them under cx
        const unsigned long long rl = (unsigned long long)(__brev(lo2) >> 16), rh = (unsigned long long)(__brev(hi2) >> 16);
        plo |= cx >= 16 ? (rl << (cx - 16)) : (rl >> (16 - cx));
        phi |= cx >= 16 ? (rh << (cx - 16)) : (rh >> (16 - cx));
        int32_t col = cx - 16;
        if (col < 1) bh = 0u;
        while (__any(bh != 0u)) {  // rare: more than 16 columns inside the shared rows
            const uint32_t w = s_sh[col >= 1 ? col - 1 : 0];
            const uint32_t Lr = w & 0xFFFFu, Hr = w >> 16;
            const uint32_t Dr = Lr & Hr;
            const uint32_t nb = bh ? (((Dr + bh) & ~Dr) & pm) : 0u;
            const bool has = nb != 0u, lo = (Lr & nb) != 0u, hi = (Hr & nb) != 0u;
            const unsigned long long bit = 1ull << (col >= 1 ? col - 1 : 0);
            plo |= lo ? bit : 0ull;
            phi |= hi ? bit : 0ull;
            const bool consume = has & !hi;
            dgh |= consume ? nb : 0u;
            bh = consume ? ((nb << 1) & pm) : nb;
            ntext += has ? 1 : 0;
            col -= has ? 1 : 0;
            if (col < 1) bh = 0u;
        }
    }
    const int32_t tstart = cand ? best_pos - ntext : 0;
    // consumed rows in natural order (row r <-> bit r-1)
    const unsigned long long diagrow = ((unsigned long long)__brev(dg) << P) | (P ? (unsigned long long)(__brev(dgh) >> (32 - P)) : 0ull) |
                                       ((unsigned long long)dgt << (P + 32));
    const unsigned long long delrow = cand ? (low64(m) & ~diagrow) : 0ull;
    if constexpr (FAST) {
        // A bound for every lane; the exact score of the best-bounded lane only, later (k_rows).  Per hit the two highest
        // bounds of BOTH candidate sets of searcher.rs:303-328 — pass 1: lowest cost <= k1, pass 2: every lane with a local
        // minimum — are collected with one pair of returning LDS atomics per set and lane (key = bound bits : 0xFFFF - p, so
        // the maximum is also the FIRST maximum; whatever a lane's atomicMax displaces or fails to displace, min(old, key),
        // is a candidate for second place, and the true second always shows up as one).  Which set counts is known after
        // the single barrier: pass 2 iff pass 1 has fewer than two members, i.e. its second place is empty.
        const float ubf = lodhi_bound_tab<CW>(cand ? plo : 0ull, cand ? phi : 0ull, cand ? tstart : 0, cand ? best_pos : 0, wmax, s_lb);
        (void)delrow;
        const unsigned long long key = ((unsigned long long)__float_as_uint(ubf) << 16) | (unsigned long long)(0xFFFFu - (uint32_t)p);
        if (N >= 64) {
            // A wave holds lanes of at most two hits.  96 lanes posting to one LDS address serialise inside the LDS unit (and
            // hold up the other waves' table reads): the wave finds its own top-2 per (hit, candidate set) on the VALU first
            // and four lanes post them.
            const uint32_t lane = threadIdx.x & 63u;
            const int hA = __builtin_amdgcn_readfirstlane(hl);
            const bool c2 = cand, c1 = cand && best_cost <= G.k1;
            const uint32_t vb = __float_as_uint(ubf);
            unsigned long long t[4], u[4];  // combos: 0 = (hit A, pass 1), 1 = (A, pass 2), 2 = (B, pass 1), 3 = (B, pass 2)
            // the two candidate sets differ only if some candidate costs more than k1, and two waves in three hold one hit:
            // usually one reduction serves all
            const bool sets_differ = __any(c2 && !c1), two_hits = __any(hl != hA);
            wave_top2(c2 && hl == hA, vb, key, t[1], u[1]);
            if (sets_differ) wave_top2(c1 && hl == hA, vb, key, t[0], u[0]);
            else { t[0] = t[1]; u[0] = u[1]; }
            t[2] = t[3] = u[2] = u[3] = 0ull;
            if (two_hits) {
                wave_top2(c2 && hl != hA, vb, key, t[3], u[3]);
                if (sets_differ) wave_top2(c1 && hl != hA, vb, key, t[2], u[2]);
                else { t[2] = t[3]; u[2] = u[3]; }
            }
            if (lane < 4u) {
                const unsigned long long kt = lane == 0u ? t[0] : lane == 1u ? t[1] : lane == 2u ? t[2] : t[3];
                const unsigned long long ku = lane == 0u ? u[0] : lane == 1u ? u[1] : lane == 2u ? u[2] : u[3];
                if (kt != 0ull) {
                    const int hx = hA + (int)(lane >> 1);
                    unsigned long long* pm = (lane & 1u) ? &s_maxB[hx] : &s_max[hx];
                    unsigned long long* ps = (lane & 1u) ? &s_secB[hx] : &s_sec[hx];
                    const unsigned long long o = atomicMax(pm, kt);
                    atomicMax(ps, o < kt ? o : kt);
                    if (ku != 0ull) atomicMax(ps, ku);
                }
            }
        } else if (cand) {
            const unsigned long long o2 = atomicMax(&s_maxB[hl], key);
            atomicMax(&s_secB[hl], o2 < key ? o2 : key);
            if (best_cost <= G.k1) {
                const unsigned long long o1 = atomicMax(&s_max[hl], key);
                atomicMax(&s_sec[hl], o1 < key ? o1 : key);
            }
        }
        build_cols(half ^ 1u);
        BB_PFX_SYNC();  // barrier B: the set's candidates are posted, the next set's column table is complete
        if (active) {
            const bool pass2 = s_sec[hl] == 0ull && G.k1 < G.k2;
            const unsigned long long mx = pass2 ? s_maxB[hl] : s_max[hl], sx = pass2 ? s_secB[hl] : s_sec[hl];
            if (mx != 0ull) {
                if ((uint32_t)p == 0xFFFFu - (uint32_t)(mx & 0xFFFFull)) {
                    bb_winrec W;
                    W.plo = plo; W.phi = phi; W.diagrow = diagrow;
                    W.ub_second = sx ? (double)__uint_as_float((uint32_t)(sx >> 16)) / G.perfect : -1.0;   // -1: no other candidate
                    W.tstart = (uint8_t)tstart; W.best_pos = (uint8_t)best_pos; W.top = (uint16_t)p;
                    W.flags = 0; W.marker = 2; W._pad[0] = W._pad[1] = 0;
#pragma unroll
                    for (int q = 0; q < 8; ++q) W._pad0[q] = 0;
                    *reinterpret_cast<bb_winrec*>(rows + hit_idx) = W;
                }
            } else if (p == 0) {  // no candidate at all: flank-only row (searcher.rs:353-362)
                bb_rowtmp R;
                bb_row& r = R.row;
                r.read_idx = H.read_idx; r.read_len = H.read_len;
                r.rel_dist_to_end = rel_dist_to_end((int64_t)H.text_start, (int64_t)H.read_len);
                r.read_start_flank = H.text_start; r.read_end_flank = H.text_end;
                r.flank_cost = H.cost; r.group_idx = H.group; r.strand = H.strand;
                r._pad[0] = 1; r._pad[1] = r._pad[2] = 0;
                r.read_start_bar = H.text_start; r.read_end_bar = H.text_end;
                r.bar_start = 0; r.bar_end = 0;
                r.match_type = (uint8_t)(G.type == BB_FTAG ? BB_FFLANK : BB_RFLANK);
                r.barcode_cost = (int16_t)G.m_bar; r.barcode_idx = -1;
                rows[hit_idx] = R;
            }
        }
        build_walks(half ^ 1u);  // read after the next barrier A; built while the slower waves finish this set
    } else {
        build_cols(half ^ 1u);
        build_walks(half ^ 1u);
        double s_norm = -1.0;
        {
            const bool on = cand;  // the loop is wave-uniform: idle lanes walk it with empty masks
            const double sc = (uint32_t)G.pol_lodhi_exp == (uint32_t)BB_LODHI_EXP_DEFAULT
                                  ? lodhi_replay<CW>(on ? plo : 0ull, on ? phi : 0ull, on ? delrow : 0ull, on ? tstart : 0, on ? best_pos : 0, wmax)
                                  : lodhi_replay<CW, true>(on ? plo : 0ull, on ? phi : 0ull, on ? delrow : 0ull, on ? tstart : 0, on ? best_pos : 0, wmax, (uint32_t)G.pol_lodhi_exp);
            if (cand) s_norm = G.perfect > 0.0 ? sc / G.perfect : 0.0;
        }
        pick_and_emit(active, cand, best_cost, s_norm, p, hl, H, hit_idx, G, plo, phi, diagrow, tstart, best_pos, s_cnt1, s_max, s_sec, s_top,
                      min_score, min_score_diff, rows);
        BB_PFX_SYNC();
    }
  }
}
