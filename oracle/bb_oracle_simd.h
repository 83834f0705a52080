/*
 * bb_oracle_simd.h — AVX-512 forms of the two inner loops of the checker's TIMING path (bbo_annotate_batch_fast), included by bb_oracle.c.
 * TEST INFRASTRUCTURE like the rest of oracle/: bench.py's cpu_baseline times this path on the GPU box's host cores so that the CPU figure
 * reported beside the GPU's is not a scalar loop against a reference that runs SIMD sassy (/root/reference/.cargo/config.toml:1-3 builds
 * Barbell with target-cpu=native; sassy's `search` is text-parallel — chunks of the text in the lanes of a vector, an overlap of pattern
 * length + k between them — and its `search_encoded_patterns` pattern-parallel: SURVEY.md Appendix B).  Same results as the scalar
 * functions of bb_oracle.c: tests/test_oracle_fast.py compares them row by row, under every local-minimum / tie / traceback policy.
 *
 *   scan_strand_simd    searcher.rs:438, one strand of the flank scan: the read cut into 8 chunks, one per 64-bit lane, each started
 *                       m + k columns early from the all-insertions column (bottom-row values <= k are exact after that, larger ones stay
 *                       larger — Ukkonen's cut-off argument, the one k_flank_verify relies on).  The vector pass only FLAGS the columns whose
 *                       cost is <= k; the local-minimum machine (lm_step) then runs, with the scalar words, over each valley (maximal run of
 *                       flagged columns) plus m + k columns of lead-in, and over the read's end with its overhang columns.  The machine emits
 *                       at a rise out of a column <= k only, and the step into a valley is a strict decrease whatever came before: valleys can
 *                       be replayed one by one, in order, and give the full scan's end positions.
 *   best_matches_simd   searcher.rs:279-301, every padded barcode of a group against one window: 8 patterns per vector, one Myers word each,
 *                       the local-minimum pick of best_match_for_pattern_fast as mask arithmetic, both move planes of every column stored
 *                       for the walks back (scalar, per pattern that matched).
 */
#include <immintrin.h>

#define BBO_SIMD_TARGET __attribute__((target("avx512f,avx512bw,avx512dq,avx512vl")))
static int bbo_have_avx512(void) {
    static int have = -1;
    if (have < 0) { __builtin_cpu_init(); have = __builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512bw") && __builtin_cpu_supports("avx512dq") && !getenv("BBO_NO_SIMD"); }
    return have;
}

/* ---- scalar words over a window of columns (the replay of a valley) ------------------------------------------------------------------ */
typedef struct { uint64_t pv[BBO_MAXW64], mv[BBO_MAXW64]; int32_t score; } bbo_colstate;
/* column 0 of the scan (left overhang deltas, scan_strand_fast's start) or the all-insertions column of a window that starts inside the text */
static int bbo_win_init(const bb_policy* P, int W, int m, float alpha, int at_zero, bbo_colstate* cs) {
    for (int w = 0; w < W; ++w) { cs->pv[w] = 0; cs->mv[w] = 0; }
    if (at_zero) {
        for (int j = 1; j <= m; ++j) {
            const int dlt = alpha >= 0.f ? overhang_cost(P, alpha, j) - overhang_cost(P, alpha, j - 1) : 1;
            if (dlt < 0 || dlt > 1) return 0;
            if (dlt) cs->pv[(j - 1) >> 6] |= 1ull << ((j - 1) & 63);
        }
        cs->score = alpha >= 0.f ? overhang_cost(P, alpha, m) : m;
    } else {
        for (int j = 1; j <= m; ++j) cs->pv[(j - 1) >> 6] |= 1ull << ((j - 1) & 63);
        cs->score = m;
    }
    return 1;
}
static inline void bbo_win_step(const uint64_t* peq, int W, int TW, int TB, uint8_t code, bbo_colstate* cs) {
    const uint64_t* e = peq + (size_t)code * W;
    uint64_t carry = 0, pin = 0, min_ = 0;
    for (int w = 0; w < W; ++w) {
        const uint64_t eq = e[w], x = eq & cs->pv[w];
        const unsigned __int128 sum = (unsigned __int128)x + cs->pv[w] + carry;
        carry = (uint64_t)(sum >> 64);
        const uint64_t d0 = (((uint64_t)sum ^ cs->pv[w]) | eq | cs->mv[w]);
        const uint64_t ph = cs->mv[w] | ~(d0 | cs->pv[w]), mh = cs->pv[w] & d0;
        if (w == TW) cs->score += (int32_t)((ph >> TB) & 1u) - (int32_t)((mh >> TB) & 1u);
        const uint64_t phs = (ph << 1) | pin, mhs = (mh << 1) | min_;
        pin = ph >> 63; min_ = mh >> 63;
        cs->pv[w] = mhs | ~(d0 | phs);
        cs->mv[w] = phs & d0;
    }
}

/* ---- the vector pass: flags[c] = 1 for the columns c in 1..n whose bottom-row cost is <= k --------------------------------------------- */
/* sc: the text's base-set codes in SCAN order with PAD readable bytes before sc[0] and behind sc[n - 1] */
#define BBO_SIMD_PAD 1024
#define BBO_SIMD_BODY(W)                                                                                                                    \
    __m512i pv[W], mv[W];                                                                                                                   \
    for (int w = 0; w < W; ++w) {                                                                                                           \
        uint64_t a[8];                                                                                                                      \
        a[0] = z0->pv[w];                                                                                                                   \
        for (int l = 1; l < 8; ++l) a[l] = zi->pv[w];                                                                                       \
        pv[w] = _mm512_loadu_si512((const void*)a);                                                                                         \
        for (int l = 0; l < 8; ++l) a[l] = l ? zi->mv[w] : z0->mv[w];                                                                       \
        mv[w] = _mm512_loadu_si512((const void*)a);                                                                                         \
    }                                                                                                                                       \
    __m512i score = _mm512_set_epi64(zi->score, zi->score, zi->score, zi->score, zi->score, zi->score, zi->score, z0->score);              \
    const __m512i one = _mm512_set1_epi64(1), kv = _mm512_set1_epi64(k), nib = _mm512_set1_epi64(15), topbit = _mm512_set1_epi64((long long)(1ull << TB)); \
    /* lane l's stream begins at sc + l C - mk: eight bytes of every stream per gather, a byte of them per step */                         \
    const __m512i stream = _mm512_sub_epi64(_mm512_mullo_epi64(_mm512_set_epi64(7, 6, 5, 4, 3, 2, 1, 0), _mm512_set1_epi64(C)), _mm512_set1_epi64(mk)); \
    const int T = C + mk;                                                                                                                   \
    __m512i text = _mm512_setzero_si512();                                                                                                  \
    for (int t = 1; t <= T; ++t) {                                                                                                          \
        const int tb = (t - 1) & 7;                                                                                                         \
        if (!tb) text = _mm512_i64gather_epi64(_mm512_add_epi64(stream, _mm512_set1_epi64(t - 1)), (const void*)sc, 1);                     \
        const __m512i code = _mm512_and_si512(_mm512_srli_epi64(text, 8 * tb), nib);                                                        \
        __m512i carry = _mm512_setzero_si512(), pin = carry, min_ = carry, npv[W], nmv[W], nscore = score;                                  \
        for (int w = 0; w < W; ++w) {                                                                                                       \
            const __m512i eq = _mm512_permutex2var_epi64(tab[2 * w], code, tab[2 * w + 1]);                                                 \
            const __m512i x = _mm512_and_si512(eq, pv[w]);                                                                                  \
            const __m512i s1 = _mm512_add_epi64(x, pv[w]);                                                                                  \
            const __m512i sum = _mm512_add_epi64(s1, carry);                                                                                \
            if (w + 1 < W) {                                                                                                                \
                const __mmask8 c1 = _mm512_cmplt_epu64_mask(s1, x), c2 = _mm512_cmplt_epu64_mask(sum, s1);                                  \
                carry = _mm512_maskz_mov_epi64((__mmask8)(c1 | c2), one);                                                                   \
            }                                                                                                                               \
            const __m512i d0 = _mm512_ternarylogic_epi64(_mm512_xor_si512(sum, pv[w]), eq, mv[w], 0xFE);          /* a | b | c */           \
            const __m512i ph = _mm512_ternarylogic_epi64(mv[w], d0, pv[w], 0xF1);                                 /* a | ~(b | c) */        \
            const __m512i mh = _mm512_and_si512(pv[w], d0);                                                                                 \
            if (w == TW)                                                                                                                    \
                nscore = _mm512_mask_sub_epi64(_mm512_mask_add_epi64(score, _mm512_test_epi64_mask(ph, topbit), score, one),                \
                                               _mm512_test_epi64_mask(mh, topbit), _mm512_mask_add_epi64(score, _mm512_test_epi64_mask(ph, topbit), score, one), one); \
            const __m512i phs = _mm512_or_si512(_mm512_slli_epi64(ph, 1), pin), mhs = _mm512_or_si512(_mm512_slli_epi64(mh, 1), min_);      \
            pin = _mm512_srli_epi64(ph, 63); min_ = _mm512_srli_epi64(mh, 63);                                                              \
            npv[w] = _mm512_ternarylogic_epi64(mhs, d0, phs, 0xF1);                                                                         \
            nmv[w] = _mm512_and_si512(phs, d0);                                                                                             \
        }                                                                                                                                   \
        if (t <= mk) { /* lane 0 starts at column 0 with the scan's own start: it waits while the others run their lead-in */             \
            for (int w = 0; w < W; ++w) { pv[w] = _mm512_mask_mov_epi64(pv[w], 0xFE, npv[w]); mv[w] = _mm512_mask_mov_epi64(mv[w], 0xFE, nmv[w]); } \
            score = _mm512_mask_mov_epi64(score, 0xFE, nscore);                                                                             \
            masks[t] = 0;                                                                                                                   \
        } else {                                                                                                                            \
            for (int w = 0; w < W; ++w) { pv[w] = npv[w]; mv[w] = nmv[w]; }                                                                 \
            score = nscore;                                                                                                                 \
            masks[t] = (uint8_t)_mm512_cmple_epi64_mask(score, kv);                                                                         \
        }                                                                                                                                   \
    }

BBO_SIMD_TARGET static void bbo_simd_pass1(const __m512i* tab, int TW, int TB, const uint8_t* sc, int C, int mk, int k, const bbo_colstate* z0,
                                           const bbo_colstate* zi, uint8_t* masks) { BBO_SIMD_BODY(1) }
BBO_SIMD_TARGET static void bbo_simd_pass2(const __m512i* tab, int TW, int TB, const uint8_t* sc, int C, int mk, int k, const bbo_colstate* z0,
                                           const bbo_colstate* zi, uint8_t* masks) { BBO_SIMD_BODY(2) }
BBO_SIMD_TARGET static void bbo_simd_pass3(const __m512i* tab, int TW, int TB, const uint8_t* sc, int C, int mk, int k, const bbo_colstate* z0,
                                           const bbo_colstate* zi, uint8_t* masks) { BBO_SIMD_BODY(3) }
BBO_SIMD_TARGET static void bbo_simd_pass4(const __m512i* tab, int TW, int TB, const uint8_t* sc, int C, int mk, int k, const bbo_colstate* z0,
                                           const bbo_colstate* zi, uint8_t* masks) { BBO_SIMD_BODY(4) }

/* the local-minimum machine over columns [from, to] of one window; the window's words start m + k columns before `from` */
static void bbo_replay(const bb_policy* P, const uint64_t* peq, int W, int m, const uint8_t* sc, int n, int k, float alpha, int from, int to,
                       int with_tail, end_list* out) {
    const int mk = m + k, TW = (m - 1) >> 6, TB = (m - 1) & 63;
    int s = from - mk;
    if (s < 0) s = 0;
    bbo_colstate cs;
    bbo_win_init(P, W, m, alpha, s == 0, &cs);
    lm_state st = {0, k + 1, 1, 0, P->lm_rule};                 /* the column before a valley is > k: whatever it holds, the step in is a strict decrease */
    if (from == 0) { lm_state fresh = {0, 0, 0, 0, P->lm_rule}; st = fresh; lm_step(&st, 0, cs.score, k, out); }
    for (int i = s + 1; i <= to; ++i) {
        bbo_win_step(peq, W, TW, TB, sc[i - 1], &cs);
        if (i >= from && i > 0) lm_step(&st, i, cs.score, k, out);
    }
    if (!with_tail) return;
    int last = n;
    if (alpha >= 0.f) {
        int32_t d = cs.score;
        for (int o = 1; o <= m; ++o) {
            const int b = m - o;
            d -= (int32_t)((cs.pv[b >> 6] >> (b & 63)) & 1u) - (int32_t)((cs.mv[b >> 6] >> (b & 63)) & 1u);
            lm_step(&st, n + o, d + overhang_cost(P, alpha, o), k, out);
        }
        last = n + m;
    }
    lm_finish(&st, last, k, out);
}

/* one strand of the flank scan; 0: not applicable here (short read, odd overhang costs, no AVX-512): the caller takes scan_strand_fast */
static int scan_strand_simd(const bb_policy* P, const uint64_t* peq, int W, int m, const uint8_t* sc /* scan order, padded */, int n, int k, float alpha,
                            end_list* out, uint8_t** scratch, size_t* scratch_cap) {
    const int mk = m + k;
    if (!bbo_have_avx512() || W > BBO_MAXW64 || n < 16 * mk || mk + 8 > BBO_SIMD_PAD) return 0;
    bbo_colstate z0, zi;
    if (!bbo_win_init(P, W, m, alpha, 1, &z0) || z0.score <= k) return 0;
    bbo_win_init(P, W, m, alpha, 0, &zi);
    const int C = (n + 7) / 8, T = C + mk;
    const size_t need = (size_t)T + 2 + (size_t)n + 2;
    if (*scratch_cap < need) { free(*scratch); *scratch = (uint8_t*)malloc(need); *scratch_cap = need; }
    uint8_t* masks = *scratch;
    uint8_t* flag = *scratch + T + 2;
    __m512i tab[2 * BBO_MAXW64];
    for (int w = 0; w < W; ++w) {
        uint64_t a[16];
        for (int c = 0; c < 16; ++c) a[c] = peq[(size_t)c * W + w];
        memcpy(&tab[2 * w], a, 64); memcpy(&tab[2 * w + 1], a + 8, 64);
    }
    const int TW = (m - 1) >> 6, TB = (m - 1) & 63;
    switch (W) {
        case 1: bbo_simd_pass1(tab, TW, TB, sc, C, mk, k, &z0, &zi, masks); break;
        case 2: bbo_simd_pass2(tab, TW, TB, sc, C, mk, k, &z0, &zi, masks); break;
        case 3: bbo_simd_pass3(tab, TW, TB, sc, C, mk, k, &z0, &zi, masks); break;
        default: bbo_simd_pass4(tab, TW, TB, sc, C, mk, k, &z0, &zi, masks); break;
    }
    memset(flag, 0, (size_t)n + 2);
    for (int t = mk + 1; t <= T; ++t) {
        unsigned mm = masks[t];
        while (mm) {
            const int l = __builtin_ctz(mm);
            mm &= mm - 1;
            const long c = (long)l * C - mk + t;                /* lane l's column at step t: inside its own chunk here */
            if (c >= 1 && c <= n) flag[c] = 1;
        }
    }
    /* valleys in order; the one that touches the read's end is replayed with the end */
    int c = 1;
    while (c <= n) {
        if (!flag[c]) { ++c; continue; }
        const int a = c;
        while (c <= n && flag[c]) ++c;
        const int b = c - 1;
        if (b == n) { bbo_replay(P, peq, W, m, sc, n, k, alpha, a - 1, n, 1, out); return 1; }
        bbo_replay(P, peq, W, m, sc, n, k, alpha, a - 1, b + 1, 0, out);
    }
    bbo_replay(P, peq, W, m, sc, n, k, alpha, n, n, 1, out);
    return 1;
}

/* ---- every padded barcode of a group against one window, 8 patterns per vector ------------------------------------------------------------ */
typedef struct { int n_vec; __m512i* eq; } bbo_bartab;   /* eq[(v * 16 + code)]: the match masks of patterns 8 v .. 8 v + 7 for a text code */

BBO_SIMD_TARGET static void bbo_bartab_build(bbo_bartab* t, const uint64_t* bpeq16, uint32_t n_seqs) {
    t->n_vec = (int)((n_seqs + 7) / 8);
    t->eq = (__m512i*)aligned_alloc(64, sizeof(__m512i) * (size_t)t->n_vec * 16);
    for (int v = 0; v < t->n_vec; ++v)
        for (int code = 0; code < 16; ++code) {
            uint64_t a[8];
            for (int l = 0; l < 8; ++l) { const uint32_t p = (uint32_t)(8 * v + l); a[l] = p < n_seqs ? bpeq16[(size_t)p * 16 + code] : 0; }
            t->eq[v * 16 + code] = _mm512_loadu_si512((const void*)a);
        }
}

/* best_match_for_pattern_fast for all patterns: has[p], best[p] (ops in ops_arena + p * ops_stride); planes: 2 * n_vec * (wn + 1) vectors of scratch */
/* opsT (may be NULL: scalar walks into ops_arena): n_vec x ops_cap vectors, ops_cap >= m + wn + 2; nopsT: n_vec vectors */
BBO_SIMD_TARGET static int best_matches_simd(const bb_policy* P, const bbo_bartab* bt, uint32_t n_seqs, int m, const uint8_t* wcode, int wn, int k,
                                             bbo_match* best, uint8_t* has, uint8_t* ops_arena, size_t ops_stride, __m512i* planes,
                                             __m512i* opsT, size_t ops_cap, __m512i* nopsT) {
    const int NV = bt->n_vec, TB = m - 1;
    const __m512i one = _mm512_set1_epi64(1), kv = _mm512_set1_epi64(k), ones = _mm512_set1_epi64(-1), topbit = _mm512_set1_epi64((long long)(1ull << TB));
    const int left = P->lm_rule == BB_LM_PLATEAU_LEFT, strict = P->lm_rule == BB_LM_STRICT, tie_last = P->bar_tie == BB_TIE_LAST;
    int matched = 0;
    for (int v = 0; v < NV; ++v) {
        __m512i pv = _mm512_set1_epi64(m >= 64 ? -1ll : (long long)((1ull << m) - 1ull)), mv = _mm512_setzero_si512();
        __m512i score = _mm512_set1_epi64(m), prev = score, best_cost = _mm512_set1_epi64(0x7FFFFFFF), best_pos = _mm512_set1_epi64(-1), cand = _mm512_setzero_si512();
        __mmask8 dec = 0xFF;
        __m512i* lo = planes + (size_t)v * 2 * (wn + 1);
        __m512i* hi = lo + (wn + 1);
        for (int c = 1; c <= wn; ++c) {
            const __m512i eq = bt->eq[v * 16 + wcode[c - 1]];
            const __m512i x = _mm512_and_si512(eq, pv);
            const __m512i d0 = _mm512_ternarylogic_epi64(_mm512_xor_si512(_mm512_add_epi64(x, pv), pv), eq, mv, 0xFE);
            const __m512i ph = _mm512_ternarylogic_epi64(mv, d0, pv, 0xF1), mh = _mm512_and_si512(pv, d0);
            score = _mm512_mask_add_epi64(score, _mm512_test_epi64_mask(ph, topbit), score, one);
            score = _mm512_mask_sub_epi64(score, _mm512_test_epi64_mask(mh, topbit), score, one);
            const __m512i phs = _mm512_slli_epi64(ph, 1), mhs = _mm512_slli_epi64(mh, 1);
            pv = _mm512_ternarylogic_epi64(mhs, d0, phs, 0xF1); mv = _mm512_and_si512(phs, d0);
            {   /* policy [H3]: per cell the first applicable op of the order (best_match_for_pattern_fast) */
                __m512i vv[4], s[4], taken = _mm512_setzero_si512();
                vv[BBO_MATCH] = _mm512_and_si512(d0, eq); vv[BBO_SUB] = _mm512_andnot_si512(d0, ones); vv[BBO_INS] = ph; vv[BBO_DEL] = pv;
                s[0] = s[1] = s[2] = s[3] = taken;
                for (int q = 0; q < 4; ++q) { const int op = P->trace_prio[q]; const __m512i y = _mm512_andnot_si512(taken, vv[op]); taken = _mm512_or_si512(taken, y); s[op] = _mm512_or_si512(s[op], y); }
                lo[c] = _mm512_or_si512(s[BBO_SUB], s[BBO_DEL]); hi[c] = _mm512_or_si512(s[BBO_INS], s[BBO_DEL]);
            }
            const __mmask8 rise = _mm512_cmpgt_epi64_mask(score, prev), fall = _mm512_cmplt_epi64_mask(score, prev);
            const __mmask8 better = (__mmask8)(_mm512_cmplt_epi64_mask(prev, best_cost) | (tie_last ? _mm512_cmpeq_epi64_mask(prev, best_cost) : 0));
            const __mmask8 take = (__mmask8)(rise & dec & _mm512_cmple_epi64_mask(prev, kv) & better);
            best_cost = _mm512_mask_mov_epi64(best_cost, take, prev);
            best_pos = _mm512_mask_mov_epi64(best_pos, take, left ? cand : _mm512_set1_epi64(c - 1));
            dec = (__mmask8)((dec & ~rise) | fall);
            if (strict) dec = (__mmask8)(dec & (rise | fall));   /* a plateau clears it */
            cand = _mm512_mask_mov_epi64(cand, fall, _mm512_set1_epi64(c));
            prev = score;
        }
        {
            const __mmask8 better = (__mmask8)(_mm512_cmplt_epi64_mask(prev, best_cost) | (tie_last ? _mm512_cmpeq_epi64_mask(prev, best_cost) : 0));
            const __mmask8 take = (__mmask8)(dec & _mm512_cmple_epi64_mask(prev, kv) & better);
            best_cost = _mm512_mask_mov_epi64(best_cost, take, prev);
            best_pos = _mm512_mask_mov_epi64(best_pos, take, left ? cand : _mm512_set1_epi64(wn));
        }
        long long bc[8], bp[8];
        _mm512_storeu_si512((void*)bc, best_cost); _mm512_storeu_si512((void*)bp, best_pos);
        if (opsT) {
            /* the eight walks back in lockstep: a lane's cell (row j, column i) -> its two move bits by gather from the planes; step t of every lane in
             * opsT[v][t] (a walk's ops in REVERSE order; lodhi_polT and bbo_ops_from_T read them back to front) */
            __m512i* oT = opsT + (size_t)v * ops_cap;
            __mmask8 active = (__mmask8)(_mm512_cmpge_epi64_mask(best_pos, _mm512_setzero_si512()) & (8 * v + 8 <= (int)n_seqs ? 0xFF : (0xFF >> (8 * v + 8 - (int)n_seqs))));
            const __mmask8 todo = active;
            const __m512i lanes = _mm512_set_epi64(7, 6, 5, 4, 3, 2, 1, 0), three = _mm512_set1_epi64(BBO_DEL), two = _mm512_set1_epi64(BBO_INS);
            __m512i i = best_pos, j = _mm512_set1_epi64(m), nops = _mm512_setzero_si512();
            int t = 0;
            while (active) {
                const __m512i idx = _mm512_add_epi64(_mm512_slli_epi64(i, 3), lanes);
                const __m512i lw = _mm512_mask_i64gather_epi64(_mm512_setzero_si512(), active, idx, (const void*)lo, 8);
                const __m512i hw = _mm512_mask_i64gather_epi64(_mm512_setzero_si512(), active, idx, (const void*)hi, 8);
                const __m512i sh = _mm512_sub_epi64(j, one);
                __m512i op = _mm512_or_si512(_mm512_and_si512(_mm512_srlv_epi64(lw, sh), one), _mm512_slli_epi64(_mm512_and_si512(_mm512_srlv_epi64(hw, sh), one), 1));
                op = _mm512_mask_mov_epi64(op, _mm512_cmpeq_epi64_mask(i, _mm512_setzero_si512()), three);   /* column 0: the rows left are deleted */
                oT[t++] = op;
                nops = _mm512_mask_add_epi64(nops, active, nops, one);
                j = _mm512_mask_sub_epi64(j, (__mmask8)(active & _mm512_cmpneq_epi64_mask(op, two)), j, one);
                i = _mm512_mask_sub_epi64(i, (__mmask8)(active & _mm512_cmpneq_epi64_mask(op, three)), i, one);
                active = (__mmask8)(active & _mm512_cmpgt_epi64_mask(j, _mm512_setzero_si512()));
            }
            long long ts[8], no[8];
            _mm512_storeu_si512((void*)ts, i); _mm512_storeu_si512((void*)no, nops);
            nopsT[v] = nops;
            for (int l = 0; l < 8; ++l) {
                const uint32_t p = (uint32_t)(8 * v + l);
                if (p >= n_seqs) break;
                has[p] = (uint8_t)((todo >> l) & 1);
                if (!has[p]) continue;
                bbo_match* b = &best[p];
                memset(b, 0, sizeof(*b));
                b->pattern_start = 0; b->pattern_end = m; b->text_start = (int)ts[l]; b->text_end = (int)bp[l];
                b->cost = (int)bc[l]; b->n_ops = (int)no[l]; b->strand = BB_FWD; b->rc_text_len = wn;
                b->ops = NULL;   /* bbo_ops_from_T for the one candidate that needs its string */
                ++matched;
            }
            continue;
        }
        for (int l = 0; l < 8; ++l) {
            const uint32_t p = (uint32_t)(8 * v + l);
            if (p >= n_seqs) break;
            has[p] = 0;
            if (bp[l] < 0) continue;
            uint8_t rev[64 + BB_FAST_MAXWIN + 2];
            int nops = 0, j = m, i = (int)bp[l];
            while (j > 0) {
                uint8_t op = BBO_DEL;
                if (i > 0) {
                    const uint64_t lw = ((const uint64_t*)&lo[i])[l], hw = ((const uint64_t*)&hi[i])[l];
                    op = (uint8_t)(((lw >> (j - 1)) & 1u) | (((hw >> (j - 1)) & 1u) << 1));
                }
                rev[nops++] = op;
                if (op != BBO_INS) --j;
                if (op != BBO_DEL) --i;
            }
            bbo_match* b = &best[p];
            memset(b, 0, sizeof(*b));
            b->pattern_start = 0; b->pattern_end = m; b->text_start = i; b->text_end = (int)bp[l];
            b->cost = (int)bc[l]; b->n_ops = nops; b->strand = BB_FWD; b->rc_text_len = wn;
            b->ops = ops_arena + ops_stride * p;
            for (int t = 0; t < nops; ++t) b->ops[t] = rev[nops - 1 - t];
            has[p] = 1; ++matched;
        }
    }
    return matched;
}

/* ---- Lodhi scores of a vector's eight candidates at once: the SAME sequence of f64 multiplications and additions per candidate as lodhi_pol (no FMA:
 * -ffp-contract=off), one candidate per lane, lanes past their op string's end left alone — bit-identical scores (tests/test_oracle_fast.py) ---------- */
/* the op string of pattern p (lane p & 7 of vector p >> 3) in forward order, from the lockstep walks' matrix */
static void bbo_ops_from_T(const __m512i* opsT, size_t ops_cap, uint32_t p, int n_ops, uint8_t* out) {
    const long long* col = (const long long*)(opsT + (size_t)(p >> 3) * ops_cap) + (p & 7);
    for (int c = 0; c < n_ops; ++c) out[c] = (uint8_t)col[(size_t)(n_ops - 1 - c) * 8];
}

/* lodhi_pol8 on the eight walks of one vector: lane l's column c is row n_ops[l] - 1 - c of the matrix (one gather per column) */
BBO_SIMD_TARGET static void lodhi_polT(const bb_policy* P, const __m512i* oT, const __m512i* nops_p, double* out) {
    const __m512i nops = *nops_p;
    const int p = P->lodhi_p;
    double dk[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int o = 0; o < 4; ++o) {
        double d = 1.0;
        for (int e = 0; e < P->lodhi_exp[o]; ++e) d = e == 0 ? P->lodhi_lambda : d * P->lodhi_lambda;
        dk[o] = d;
    }
    const __m512d dkv = _mm512_loadu_pd(dk), onev = _mm512_set1_pd(1.0);
    __m512d A[4] = {_mm512_setzero_pd(), _mm512_setzero_pd(), _mm512_setzero_pd(), _mm512_setzero_pd()}, score = _mm512_setzero_pd();
    long long no[8];
    _mm512_storeu_si512((void*)no, nops);
    int maxn = 0;
    for (int l = 0; l < 8; ++l) if (no[l] > maxn) maxn = (int)no[l];
    const __m512i lanes = _mm512_set_epi64(7, 6, 5, 4, 3, 2, 1, 0), one = _mm512_set1_epi64(1);
    __m512i row = _mm512_sub_epi64(nops, one);                     /* the matrix row of every lane's column 0 */
    for (int c = 0; c < maxn; ++c) {
        const __mmask8 live = _mm512_cmpge_epi64_mask(row, _mm512_setzero_si512());
        const __m512i opv = _mm512_mask_i64gather_epi64(_mm512_setzero_si512(), live, _mm512_add_epi64(_mm512_slli_epi64(row, 3), lanes), (const void*)oT, 8);
        row = _mm512_sub_epi64(row, one);
        const __m512d d = _mm512_permutexvar_pd(opv, dkv);
        const __mmask8 isM = (__mmask8)(_mm512_cmpeq_epi64_mask(opv, _mm512_set1_epi64(BBO_MATCH)) & live), notM = (__mmask8)(live & ~isM);
        const __m512d prod = _mm512_mul_pd(d, p >= 2 ? A[p - 2] : onev);
        score = _mm512_mask_add_pd(score, isM, score, prod);
        for (int qq = p - 2; qq >= 1; --qq) {
            const __m512d m_ = _mm512_mul_pd(d, _mm512_add_pd(A[qq], A[qq - 1])), o_ = _mm512_mul_pd(d, A[qq]);
            A[qq] = _mm512_mask_mov_pd(_mm512_mask_mov_pd(A[qq], isM, m_), notM, o_);
        }
        if (p >= 2) {
            const __m512d m_ = _mm512_mul_pd(d, _mm512_add_pd(A[0], onev)), o_ = _mm512_mul_pd(d, A[0]);
            A[0] = _mm512_mask_mov_pd(_mm512_mask_mov_pd(A[0], isM, m_), notM, o_);
        }
    }
    _mm512_storeu_pd(out, score);
}
