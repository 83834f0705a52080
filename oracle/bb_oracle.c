/*
 * bb_oracle.c — CPU restatement of Barbell's annotate hot path (Demuxer::demux and everything
 * under it).  TEST INFRASTRUCTURE: the checker for the HIP path, never the thing shipped.
 *
 * PINNING STATUS: "parity unpinned" beyond the reference's own known-answer tests.
 *   The first-party logic (searcher.rs, cigar_parse.rs, interval.rs, barcodes.rs, edit_model.rs)
 *   is restated line by line below with file:line citations.  The arithmetic underneath it lives
 *   in third-party crates that are NOT in /root/reference and cannot be fetched or built here
 *   (no cargo/rustc, no network):
 *       sassy 0.2.1          (Cargo.lock:1060-1079)  Searcher::search / search_encoded_patterns
 *       cigar-lodhi-rs 0.1.0 (Cargo.lock:242-248)    Lodhi::new(3, 0.5).compute(&Cigar)
 *       pa-types 1.2.0       (Cargo.lock:764-772)    Cigar / CigarOp / Pos
 *   Their published behaviour is restated in the functions tagged [H1]..[H9] (hazard numbers of
 *   SURVEY.md §8c).  These are pinned ONLY by the five sassy known-answer tests of
 *   src/annotate/cigar_parse.rs:104-176 (tests/test_oracle_kat.py) plus what Barbell's own code
 *   lets one deduce.  Every tie-break that those do not pin is a documented
 *   assumption; the HIP path must be bit-identical to THIS file.
 *
 * Published algorithm being restated (sassy): semi-global unit-cost edit distance of a pattern
 * against every end position of a text (pattern global, text local: D[0][i] = 0, D[j][0] = j),
 * IUPAC matching (two characters match iff their base sets intersect), reverse complement handled
 * by searching complement(pattern) in reversed(text), optional overhang: pattern characters that
 * fall outside the text cost alpha each (rounded down), matches reported at local minima of the
 * end-position cost that are <= k, each with a traceback (CIGAR).
 */
#define _GNU_SOURCE   /* sched_setaffinity: the batch entry points pin their OpenMP workers */
#include "bb_oracle.h"

#include <malloc.h>
#include <math.h>
#include <sched.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define PADDING 10 /* src/lib.rs:10 */

static int g_full_trace = 0;
void bbo_set_full_trace(int on) { g_full_trace = on; }
static int g_pin_threads = 1;   /* bbo_annotate_batch*: one pinned worker per CPU (see annotate_batch_impl) */
void bbo_set_pin_threads(int on) { g_pin_threads = on; }

/* The switchable assumptions (include/barbell_amd_policy.h).  g_pol serves the stand-alone entry points (bbo_search,
 * bbo_lodhi) and is the policy of contexts made by bbo_create; bbo_create_policy carries its own. */
static bb_policy g_pol;
static int g_pol_set = 0;
static const bb_policy* cur_pol(void) {
    if (!g_pol_set) { bb_policy_default(&g_pol); g_pol_set = 1; }
    return &g_pol;
}
int bbo_set_policy(const bb_policy* p) {
    if (!p) { bb_policy_default(&g_pol); g_pol_set = 1; return BB_OK; }
    if (bb_policy_validate(p)) return BB_E_INVALID;
    g_pol = *p; g_pol_set = 1;
    return BB_OK;
}

/* ------------------------------------------------------------------------------------------ */
/* [H6] IUPAC profile.  4-bit base sets A=1 C=2 G=4 T=8; case-insensitive; X = empty set;      */
/* anything that is not an IUPAC letter is invalid (0xFF) for queries (barcodes.rs:45-47) and  */
/* matches nothing when it appears in a read.                                                  */
/* ------------------------------------------------------------------------------------------ */
uint8_t bbo_iupac_code(uint8_t c) {
    switch (c) {
        case 'A': case 'a': return 1;
        case 'C': case 'c': return 2;
        case 'G': case 'g': return 4;
        case 'T': case 't': case 'U': case 'u': return 8;
        case 'R': case 'r': return 1 | 4;
        case 'Y': case 'y': return 2 | 8;
        case 'S': case 's': return 4 | 2;
        case 'W': case 'w': return 1 | 8;
        case 'K': case 'k': return 4 | 8;
        case 'M': case 'm': return 1 | 2;
        case 'B': case 'b': return 2 | 4 | 8;
        case 'D': case 'd': return 1 | 4 | 8;
        case 'H': case 'h': return 1 | 2 | 8;
        case 'V': case 'v': return 1 | 2 | 4;
        case 'N': case 'n': return 15;
        case 'X': case 'x': return 0;
        default: return 0xFF;
    }
}
static inline uint8_t text_code(uint8_t c) {
    uint8_t k = bbo_iupac_code(c);
    return k == 0xFF ? 0 : k;
}
/* the same as a table (the timing path converts every base of every read) */
static const uint8_t* text_code_table(void) {
    static uint8_t T[256];
    static int made = 0;
    if (!made) {
#pragma omp critical(bbo_text_code_table)
        { if (!made) { for (int c = 0; c < 256; ++c) T[c] = text_code((uint8_t)c); __atomic_store_n(&made, 1, __ATOMIC_RELEASE); } }
    }
    return T;
}
/* complement of a base set: A<->T, C<->G */
static inline uint8_t comp_code(uint8_t k) {
    return (uint8_t)(((k & 1) << 3) | ((k & 8) >> 3) | ((k & 2) << 1) | ((k & 4) >> 1));
}
/* barcodes.rs:394-441 RC table: complement that keeps case and IUPAC codes */
static uint8_t rc_char(uint8_t c) {
    static const char from[] = "ACTGactgRYSWKMBDHVNXryswkmbdhvnx";
    static const char to[]   = "TGACtgacYRSWMKVHDBNXyrswmkvhdbnx";
    for (int i = 0; from[i]; ++i)
        if ((uint8_t)from[i] == c) return (uint8_t)to[i];
    return c;
}

/* [H4] overhang cost of `len` pattern characters outside the text: round(alpha * len), rounding mode and width of
 * the product by policy (default: floor, f32) */
static inline int overhang_cost(const bb_policy* P, float alpha, int len) {
    if (P->ovh_round & BB_OVH_F64) {
        const double v = (double)len * (double)alpha;
        switch (P->ovh_round & 3) { case BB_OVH_CEIL: return (int)ceil(v); case BB_OVH_NEAR: return (int)nearbyint(v); default: return (int)floor(v); }
    }
    const float v = (float)len * alpha;
    switch (P->ovh_round & 3) { case BB_OVH_CEIL: return (int)ceilf(v); case BB_OVH_NEAR: return (int)nearbyintf(v); default: return (int)floorf(v); }
}

/* ------------------------------------------------------------------------------------------ */
/* DP primitives                                                                               */
/* ------------------------------------------------------------------------------------------ */
/* advance one text column in place: col[j] = D[j][i-1] -> D[j][i]; col[0] stays 0 */
static void dp_column(const uint8_t* pcode, int m, uint8_t tcode, int32_t* col) {
    int32_t diag = col[0];
    for (int j = 1; j <= m; ++j) {
        int32_t left = col[j];
        int32_t best = diag + ((pcode[j - 1] & tcode) ? 0 : 1);
        if (left + 1 < best) best = left + 1;
        if (col[j - 1] + 1 < best) best = col[j - 1] + 1;
        diag = left;
        col[j] = best;
    }
}

typedef struct { int32_t e, cost; } end_hit;
typedef struct { end_hit* v; int n, cap; } end_list;
static void end_push(end_list* l, int e, int cost) {
    if (l->n == l->cap) { l->cap = l->cap ? 2 * l->cap : 8; l->v = (end_hit*)realloc(l->v, sizeof(end_hit) * (size_t)l->cap); }
    l->v[l->n].e = e; l->v[l->n].cost = cost; l->n++;
}

/*
 * [H1] local-minimum rule.  C[0..imax] is the cost of the best match ENDING at position i
 * (i characters of the text consumed).  A position is reported when the cost sequence stops
 * decreasing there: state `decreasing` becomes true on a strict decrease, false on a strict
 * increase, is unchanged on a plateau; on a strict increase the PREVIOUS position is reported if
 * `decreasing` and its cost <= k (so a plateau is reported at its right end); at the end of the
 * sequence the last position is reported under the same condition.  `decreasing` starts true.
 * This streaming form is what the scan kernels implement.
 * Policy [H1]: BB_LM_PLATEAU_LEFT reports the same minima at the LEFT end of their plateau (the position of the last
 * strict decrease, `cand`); BB_LM_STRICT lets a plateau clear `decreasing`, so only strict minima are reported.
 */
typedef struct { int decreasing; int32_t prev; int started; int cand; int rule; } lm_state;
static inline void lm_step(lm_state* s, int idx, int32_t cur, int k, end_list* out) {
    if (!s->started) { s->started = 1; s->decreasing = 1; s->prev = cur; s->cand = idx; return; }
    if (cur > s->prev) {
        if (s->decreasing && s->prev <= k) end_push(out, s->rule == BB_LM_PLATEAU_LEFT ? s->cand : idx - 1, s->prev);
        s->decreasing = 0;
    } else if (cur < s->prev) {
        s->decreasing = 1; s->cand = idx;
    } else if (s->rule == BB_LM_STRICT) {
        s->decreasing = 0;
    }
    s->prev = cur;
}
static inline void lm_finish(lm_state* s, int last_idx, int k, end_list* out) {
    if (s->started && s->decreasing && s->prev <= k) end_push(out, s->rule == BB_LM_PLATEAU_LEFT ? s->cand : last_idx, s->prev);
}

/*
 * One strand of Searcher::search.  pcode: pattern base sets (m), tcode: text base sets in scan
 * direction (n).  alpha < 0: no overhang (D[j][0] = j, ends 0..n).  alpha >= 0 [H4]:
 *   left : D[j][0] = floor(alpha*j)           (pattern prefix of length j before the text)
 *   right: C[n+o] = D[m-o][n] + floor(alpha*o), o = 1..m (pattern suffix of length o after it)
 */
static void scan_strand(const bb_policy* P, const uint8_t* pcode, int m, const uint8_t* tcode, int n, int k, float alpha,
                        end_list* out) {
    int32_t* col = (int32_t*)malloc(sizeof(int32_t) * (size_t)(m + 1));
    for (int j = 0; j <= m; ++j) col[j] = alpha >= 0.f ? overhang_cost(P, alpha, j) : j;
    lm_state st = {0, 0, 0, 0, P->lm_rule};
    lm_step(&st, 0, col[m], k, out);
    for (int i = 1; i <= n; ++i) {
        dp_column(pcode, m, tcode[i - 1], col);
        lm_step(&st, i, col[m], k, out);
    }
    int last = n;
    if (alpha >= 0.f) {
        for (int o = 1; o <= m; ++o) lm_step(&st, n + o, col[m - o] + overhang_cost(P, alpha, o), k, out);
        last = n + m;
    }
    lm_finish(&st, last, k, out);
    free(col);
}

/*
 * [H3] traceback (sassy get_trace restated).  Walk from the end cell back to row 0, preferring, at
 * cell (j,i) with cost g:  (1) diagonal Match if D[j-1][i-1]==g and the characters match,
 * (2) Ins  (text char only)   if D[j][i-1]   == g-1,
 * (3) Sub  (diagonal)         if D[j-1][i-1] == g-1,
 * (4) Del  (pattern char only) if D[j-1][i]  == g-1.
 * KAT cigar_parse.rs:163-176 pins (3) before (4).  At text column 0 with pattern left: with
 * overhang the walk stops (pattern_start = j, those characters are outside the text); without,
 * the rest of the pattern is consumed as Del at column 0 (KAT cigar_parse.rs:137-148).
 * The DP matrix is recomputed on the last m+k text columns only; that window provably contains
 * every cell the walk can visit (tests/test_oracle_props.py checks it against the full matrix).
 * `e` is the scan end index (0..n, or n+o for a right overhang of o characters).
 */
static void trace_match(const bb_policy* P, const uint8_t* pcode, int m, const uint8_t* tcode, int n, int k, float alpha,
                        int e, int cost, bbo_match* out) {
    int o = e > n ? e - n : 0;
    int j0 = m - o, i0 = e > n ? n : e;
    int g = cost - (o ? overhang_cost(P, alpha, o) : 0);
    int s0 = i0 - (m + k);
    if (s0 < 0 || g_full_trace) s0 = 0;
    int w = i0 - s0;
    size_t stride = (size_t)(w + 1);
    int32_t* D = (int32_t*)malloc(sizeof(int32_t) * (size_t)(m + 1) * stride);
    int32_t* col = (int32_t*)malloc(sizeof(int32_t) * (size_t)(m + 1));
    for (int j = 0; j <= m; ++j) {
        col[j] = (s0 == 0 && alpha >= 0.f) ? overhang_cost(P, alpha, j) : j;
        D[(size_t)j * stride] = col[j];
    }
    for (int c = 1; c <= w; ++c) {
        dp_column(pcode, m, tcode[s0 + c - 1], col);
        for (int j = 0; j <= m; ++j) D[(size_t)j * stride + (size_t)c] = col[j];
    }
#define DD(j, i) D[(size_t)(j) * stride + (size_t)(i)]
    if (DD(j0, w) != g) { fprintf(stderr, "bb_oracle: trace cost mismatch %d vs %d\n", DD(j0, w), g); abort(); }
    uint8_t* rev = (uint8_t*)malloc((size_t)(m + w + 2));
    int nops = 0, j = j0, i = w;
    while (j > 0) {
        if (i == 0 && s0 == 0 && alpha >= 0.f) break; /* left overhang: rest of pattern is outside */
        int took = 0;
        for (int q = 0; q < 4 && !took; ++q) {  /* policy [H3]: the first applicable op in order of preference */
            switch (P->trace_prio[q]) {
                case BBO_MATCH: if (i > 0 && DD(j - 1, i - 1) == g && (pcode[j - 1] & tcode[s0 + i - 1])) { rev[nops++] = BBO_MATCH; --j; --i; took = 1; } break;
                case BBO_INS:   if (i > 0 && DD(j, i - 1) == g - 1) { rev[nops++] = BBO_INS; --i; --g; took = 1; } break;
                case BBO_SUB:   if (i > 0 && DD(j - 1, i - 1) == g - 1) { rev[nops++] = BBO_SUB; --j; --i; --g; took = 1; } break;
                default:        if (DD(j - 1, i) == g - 1) { rev[nops++] = BBO_DEL; --j; --g; took = 1; } break;
            }
        }
        if (!took) { fprintf(stderr, "bb_oracle: trace failed at (%d,%d)\n", j, i); abort(); }
    }
#undef DD
    out->pattern_start = j; out->pattern_end = j0;
    out->text_start = s0 + i; out->text_end = i0;
    out->cost = cost; out->n_ops = nops;
    out->ops = (uint8_t*)malloc((size_t)(nops ? nops : 1));
    for (int t = 0; t < nops; ++t) out->ops[t] = rev[nops - 1 - t];
    free(rev); free(col); free(D);
}

/*
 * sassy Searcher::<Iupac>::search(pattern, text, k) restated.
 * [H2] order of the result: forward-strand matches by ascending end position, then
 * reverse-complement matches in the order the rc scan finds them (ascending end position in the
 * REVERSED text).  KAT cigar_parse.rs:163-176 pins that the forward match comes first.
 * [H5] rc: complement(pattern) (same index order) is searched in reversed(text); text_start/end are
 * mirrored back to forward coordinates, ops stay in pattern order.
 */
static int search_pol(const bb_policy* P, const uint8_t* pat, int m, const uint8_t* text, int n, int k, float alpha, int rc, bbo_match** out) {
    uint8_t* pc = (uint8_t*)malloc((size_t)(m ? m : 1));
    uint8_t* tc = (uint8_t*)malloc((size_t)(n ? n : 1));
    for (int j = 0; j < m; ++j) pc[j] = text_code(pat[j]);
    for (int i = 0; i < n; ++i) tc[i] = text_code(text[i]);
    end_list ends = {0, 0, 0};
    scan_strand(P, pc, m, tc, n, k, alpha, &ends);
    int nf = ends.n, total = nf;
    bbo_match* ms = (bbo_match*)calloc((size_t)(nf ? nf : 1), sizeof(bbo_match));
    for (int t = 0; t < nf; ++t) {
        trace_match(P, pc, m, tc, n, k, alpha, ends.v[t].e, ends.v[t].cost, &ms[t]);
        ms[t].strand = BB_FWD; ms[t].pattern_idx = 0; ms[t].rc_text_len = n;
    }
    if (rc) {
        uint8_t* pcc = (uint8_t*)malloc((size_t)(m ? m : 1));
        uint8_t* trv = (uint8_t*)malloc((size_t)(n ? n : 1));
        for (int j = 0; j < m; ++j) pcc[j] = comp_code(pc[j]);
        for (int i = 0; i < n; ++i) trv[i] = tc[n - 1 - i];
        end_list re = {0, 0, 0};
        scan_strand(P, pcc, m, trv, n, k, alpha, &re);
        total = nf + re.n;
        ms = (bbo_match*)realloc(ms, sizeof(bbo_match) * (size_t)(total ? total : 1));
        for (int t = 0; t < re.n; ++t) {
            /* policy [H2]: the rc matches as the rc scan finds them, or in ascending forward position */
            bbo_match* mm = &ms[nf + (P->rc_order == BB_RC_FWD_ORDER ? re.n - 1 - t : t)];
            memset(mm, 0, sizeof(*mm));
            trace_match(P, pcc, m, trv, n, k, alpha, re.v[t].e, re.v[t].cost, mm);
            int ts = mm->text_start, te = mm->text_end;
            mm->text_start = n - te; mm->text_end = n - ts;
            mm->strand = BB_RC; mm->pattern_idx = 0; mm->rc_text_len = n;
            mm->rc_mirror_len = P->rc_path == BB_RCPATH_MIRROR ? m : 0;
        }
        free(re.v); free(pcc); free(trv);
    }
    free(ends.v); free(pc); free(tc);
    *out = ms;
    return total;
}
int bbo_search(const uint8_t* pat, int m, const uint8_t* text, int n, int k, float alpha, int rc, bbo_match** out) {
    return search_pol(cur_pol(), pat, m, text, n, k, alpha, rc, out);
}
void bbo_free_matches(bbo_match* ms, int n) {
    if (!ms) return;
    for (int i = 0; i < n; ++i) free(ms[i].ops);
    free(ms);
}

/*
 * Match::to_path restated: exactly one Pos per unit op, path[t] = the cell at which op t is
 * applied (before advancing) — SURVEY Appendix A; KATs cigar_parse.rs:137-176 distinguish this
 * from "after advancing".  [H5] For Rc matches the walk runs in reversed-text coordinates
 * (i' ascending from n - text_end) and each text index is mirrored to max(0, n-1-i'), so pattern
 * indices stay comparable with the forward flank and text indices run downwards
 * (cigar_parse.rs:79-81 takes min/max).  Policy [H5] BB_RCPATH_MIRROR: the pattern indices of an Rc match are those of the
 * reverse-complemented pattern instead, i -> m - 1 - i (what a search of rc(pattern) in the forward text would yield), so
 * get_matching_region's bar_region selects the mirrored rows.
 */
int bbo_to_path(const bbo_match* m, bbo_pos* path) {
    int j = m->pattern_start;
    int i = m->strand == BB_RC ? m->rc_text_len - m->text_end : m->text_start;
    for (int t = 0; t < m->n_ops; ++t) {
        path[t].i = m->rc_mirror_len ? m->rc_mirror_len - 1 - j : j;   /* policy [H5]: mirrored pattern indices for Rc matches */
        if (m->strand == BB_RC) { int f = m->rc_text_len - 1 - i; path[t].j = f < 0 ? 0 : f; }
        else path[t].j = i;
        switch (m->ops[t]) {
            case BBO_MATCH: case BBO_SUB: ++j; ++i; break;
            case BBO_INS: ++i; break;
            default: ++j; break;
        }
    }
    return m->n_ops;
}

/* cigar_parse.rs:6-68 map_pat_to_text_with_cost + compute_subpath_cost. returns 0 when None */
int bbo_map_pat_to_text_with_cost(const bbo_match* m, int p_start, int p_end,
                                  int* pat_lo, int* pat_hi, int* txt_lo, int* txt_hi, int* cost) {
    bbo_pos* path = (bbo_pos*)malloc(sizeof(bbo_pos) * (size_t)(m->n_ops ? m->n_ops : 1));
    int n = bbo_to_path(m, path);
    int si = -1, ei = -1;
    for (int t = 0; t < n; ++t)
        if (path[t].i >= p_start && path[t].i < p_end) { if (si < 0) si = t; ei = t; }
    if (si < 0) { free(path); return 0; }
    int c = 0;
    for (int t = si; t <= ei; ++t) c += m->ops[t] != BBO_MATCH; /* cigar_parse.rs:47-68 */
    *pat_lo = path[si].i; *pat_hi = path[ei].i + 1;
    *txt_lo = path[si].j; *txt_hi = path[ei].j + 1;
    *cost = c;
    free(path);
    return 1;
}

/* cigar_parse.rs:71-82: first and last path cell with pattern idx in [start,end] (inclusive);
 * `next()` then `next_back()` on one filtered iterator => None unless there are >= 2 such cells */
int bbo_get_matching_region(const bbo_match* m, int start, int end, int* lo, int* hi) {
    bbo_pos* path = (bbo_pos*)malloc(sizeof(bbo_pos) * (size_t)(m->n_ops ? m->n_ops : 1));
    int n = bbo_to_path(m, path);
    int first = -1, last = -1, cnt = 0;
    for (int t = 0; t < n; ++t)
        if (path[t].i >= start && path[t].i <= end) { if (first < 0) first = t; last = t; ++cnt; }
    if (cnt < 2) { free(path); return 0; }
    int a = path[first].j, b = path[last].j;
    *lo = a < b ? a : b; *hi = a < b ? b : a;
    free(path);
    return 1;
}

/*
 * [H8] Lodhi::new(3, 0.5).compute(&cigar) restated.  The CIGAR is read as a string of alignment
 * columns (one per unit op), x_c = 1 for Match columns.  Score = gap-weighted count of length-3
 * subsequences of matched columns (Lodhi et al. 2002 string subsequence kernel, p = 3):
 *        sum over c1<c2<c3, all matched, of lambda^(c3 - c1 + 1),   lambda = 0.5
 * evaluated left to right in f64 by the standard recurrences
 *        score += lambda*A2 ; A2 = lambda*(A2 + A1) ; A1 = lambda*(A1 + 1)     (match column)
 *                             A2 = lambda*A2        ; A1 = lambda*A1           (other column)
 * The HIP kernel runs the identical sequence of f64 operations (no FMA contraction).
 * Policy [H8] generalises the three things the crate may do differently: the subsequence length p (A[q] = weighted count
 * of matched (q+1)-subsequences ending at or before the column), lambda, and the decay exponent of a column per op
 * (d = lambda^exp[op]; exp = 1,1,1,1 is the formula above, 2,2,1,1 weighs a triple by lambda^(span in the pattern +
 * span in the text)).  The default (p = 3, lambda = 0.5, exponents 1) performs exactly the operations written above.
 */
static double lodhi_pol(const bb_policy* P, const uint8_t* ops, int n_ops) {
    const int p = P->lodhi_p;
    double dk[4], A[4] = {0.0, 0.0, 0.0, 0.0}, score = 0.0;
    for (int o = 0; o < 4; ++o) {  /* lambda^exp by repeated multiplication (exp 1: lambda itself, exp 0: 1.0) */
        double d = 1.0;
        for (int e = 0; e < P->lodhi_exp[o]; ++e) d = e == 0 ? P->lodhi_lambda : d * P->lodhi_lambda;
        dk[o] = d;
    }
    for (int c = 0; c < n_ops; ++c) {
        const double d = dk[ops[c] & 3];
        if (ops[c] == BBO_MATCH) {
            score = score + d * (p >= 2 ? A[p - 2] : 1.0);
            for (int q = p - 2; q >= 1; --q) A[q] = d * (A[q] + A[q - 1]);
            if (p >= 2) A[0] = d * (A[0] + 1.0);
        } else {
            for (int q = p - 2; q >= 0; --q) A[q] = d * A[q];
        }
    }
    return score;
}
double bbo_lodhi(const uint8_t* ops, int n_ops) { return lodhi_pol(cur_pol(), ops, n_ops); }

/* edit_model.rs:2-11 */
int bbo_edit_cut_off(int l) {
    double a = (double)l;
    double value = 0.5100 * a - 1.7312 * sqrt(a);
    double c = ceil(value);
    return c > 0.0 ? (int)c : 0;
}

/* searcher.rs:183-199 */
int bbo_rel_dist_to_end(long pos, long read_len) {
    if (pos < 0) return 1;
    if (pos <= read_len / 2) return pos == 0 ? 1 : (int)pos;
    if (pos == read_len) return -1;
    return (int)-(read_len - pos);
}

/* ------------------------------------------------------------------------------------------ */
/* interval.rs:4-79 collapse_overlapping_matches                                               */
/* ------------------------------------------------------------------------------------------ */
static int is_overlap(const bb_row* a, const bb_row* b, float threshold) { /* interval.rs:30-42 */
    uint32_t start = a->read_start_flank > b->read_start_flank ? a->read_start_flank : b->read_start_flank;
    uint32_t end = a->read_end_flank < b->read_end_flank ? a->read_end_flank : b->read_end_flank;
    if (end <= start) return 0;
    uint32_t overlap = end - start;
    uint32_t la = a->read_end_flank - a->read_start_flank, lb = b->read_end_flank - b->read_start_flank;
    uint32_t min_len = la < lb ? la : lb;
    return ((float)overlap / (float)min_len) >= threshold;
}
/* interval.rs:48-76 comparator: <0 if a sorts before b */
static int best_cmp(const bb_row* a, const bb_row* b) {
    int pa = (a->match_type == BB_FTAG || a->match_type == BB_RTAG) ? 1 : 2;
    int pb = (b->match_type == BB_FTAG || b->match_type == BB_RTAG) ? 1 : 2;
    if (pa != pb) return pa < pb ? -1 : 1;
    if (pa == 1) {
        if (a->barcode_cost != b->barcode_cost) return a->barcode_cost < b->barcode_cost ? -1 : 1;
        if (a->flank_cost != b->flank_cost) return a->flank_cost < b->flank_cost ? -1 : 1;
        return 0;
    }
    uint32_t la = a->read_end_flank - a->read_start_flank, lb = b->read_end_flank - b->read_start_flank;
    if (la != lb) return la > lb ? -1 : 1; /* longer first */
    return 0;
}
int bbo_collapse(bb_row* rows, int n, float filter_overlap) {
    if (n == 0) return 0;
    /* stable sort by read_start_flank (interval.rs:12) */
    for (int i = 1; i < n; ++i) {
        bb_row x = rows[i];
        int j = i - 1;
        while (j >= 0 && rows[j].read_start_flank > x.read_start_flank) { rows[j + 1] = rows[j]; --j; }
        rows[j + 1] = x;
    }
    int out = 0, gs = 0; /* current group = rows[gs..i) */
    for (int i = 1; i <= n; ++i) {
        int joins = 0;
        if (i < n)
            for (int g = gs; g < i; ++g)
                if (is_overlap(&rows[g], &rows[i], filter_overlap)) { joins = 1; break; }
        if (!joins) {
            /* select_best_match: stable sort by best_cmp, take [0] = first minimal element */
            int best = gs;
            for (int g = gs + 1; g < i; ++g)
                if (best_cmp(&rows[g], &rows[best]) < 0) best = g;
            bb_row b = rows[best];
            rows[out++] = b; /* out <= gs, safe */
            gs = i;
        }
    }
    return out;
}

/* ------------------------------------------------------------------------------------------ */
/* query preparation: BarcodeGroup::new (barcodes.rs:105-197)                                  */
/* ------------------------------------------------------------------------------------------ */
typedef struct {
    uint32_t n_seqs, seq_len;
    uint8_t* flank; uint32_t flank_len, prefix_len, suffix_len, mask_len;
    uint32_t bar_lo, bar_hi, pad_lo, pad_hi, m_bar;
    uint8_t *pat_fwd, *pat_rc; /* n_seqs x m_bar ASCII */
    uint8_t type; int32_t flank_k, k1, k2; double perfect;
    /* match masks of the bit-parallel timing path (bbo_annotate_batch_fast): bit j of word w <-> pattern character 64 w + j */
    int W64;               /* 64-bit words of the flank */
    uint64_t* fpeq[2];     /* [strand][16 text codes][W64]; strand 1 = complement(flank) */
    uint64_t* bpeq[2];     /* [strand][n_seqs][16 text codes], padded barcodes (<= 64 characters) */
    void* bartab[2];       /* bbo_bartab (bb_oracle_simd.h): the same masks, 8 patterns per vector; NULL without AVX-512 */
} ogroup;

struct bbo_ctx { uint32_t n_groups; ogroup* g; bb_params p; bb_policy pol; };

static int prep_group(const bb_policy* P, const bb_group_desc* d, ogroup* g) {
    memset(g, 0, sizeof(*g));
    if (!d->seqs || !d->seq_lens || d->n_seqs == 0) return BB_E_INVALID;
    if (d->n_seqs == 1) return BB_E_ONE_QUERY;                     /* barcodes.rs:113-117 */
    uint32_t L = d->seq_lens[0], n = d->n_seqs;
    for (uint32_t s = 1; s < n; ++s)
        if (d->seq_lens[s] != L) return BB_E_UNEQUAL_LEN;          /* barcodes.rs:325-328 */
    if (L == 0) return BB_E_NO_BARCODE;
    for (uint32_t s = 0; s < n; ++s) {
        if (!d->seqs[s]) return BB_E_INVALID;
        for (uint32_t i = 0; i < L; ++i)
            if (bbo_iupac_code(d->seqs[s][i]) == 0xFF) return BB_E_NOT_IUPAC; /* barcodes.rs:45-47 */
    }
    /* longest common prefix / suffix, barcodes.rs:337-385 */
    uint32_t pre = L, suf = L;
    for (uint32_t s = 1; s < n; ++s) {
        uint32_t c = 0;
        while (c < L && d->seqs[0][c] == d->seqs[s][c]) ++c;
        if (c < pre) pre = c;
        c = 0;
        while (c < L && d->seqs[0][L - 1 - c] == d->seqs[s][L - 1 - c]) ++c;
        if (c < suf) suf = c;
    }
    if (pre + suf >= L) return BB_E_NO_BARCODE;                    /* barcodes.rs:124-128 */
    if (pre == 0 && suf == 0) return BB_E_NO_FLANK;                /* barcodes.rs:131-133 */
    uint32_t mask = L - pre - suf;
    g->n_seqs = n; g->seq_len = L; g->prefix_len = pre; g->suffix_len = suf; g->mask_len = mask;
    g->flank_len = L;
    g->flank = (uint8_t*)malloc(L);
    memcpy(g->flank, d->seqs[0], pre);
    memset(g->flank + pre, 'N', mask);                             /* barcodes.rs:145-154 */
    memcpy(g->flank + pre + mask, d->seqs[0] + L - suf, suf);
    g->bar_lo = pre; g->bar_hi = pre + mask - 1;                   /* barcodes.rs:192 */
    g->pad_lo = pre >= PADDING ? pre - PADDING : 0;                /* barcodes.rs:160-163 */
    g->pad_hi = pre + mask + PADDING;
    uint32_t end = g->pad_hi < L ? g->pad_hi : L;                  /* barcodes.rs:167 */
    g->m_bar = end - g->pad_lo;
    g->pat_fwd = (uint8_t*)malloc((size_t)n * g->m_bar);
    g->pat_rc = (uint8_t*)malloc((size_t)n * g->m_bar);
    for (uint32_t s = 0; s < n; ++s) {
        memcpy(g->pat_fwd + (size_t)s * g->m_bar, d->seqs[s] + g->pad_lo, g->m_bar);
        for (uint32_t i = 0; i < g->m_bar; ++i)                    /* barcodes.rs:85-88,394-396 */
            g->pat_rc[(size_t)s * g->m_bar + i] = rc_char(d->seqs[s][g->pad_lo + g->m_bar - 1 - i]);
    }
    g->type = d->type;
    g->flank_k = d->flank_k >= 0 ? d->flank_k : bbo_edit_cut_off((int)(pre + suf)); /* annotator.rs:219-226 */
    g->k1 = (int32_t)((float)g->m_bar * 0.4f);                     /* searcher.rs:458-460 */
    g->k2 = (int32_t)g->m_bar;                                     /* searcher.rs:276 */
    /* searcher.rs:229-239: all-Match CIGAR of length pad_hi - pad_lo (pad_hi NOT clamped) */
    uint32_t lbar = g->pad_hi - g->pad_lo;
    uint8_t* perfect = (uint8_t*)calloc(lbar, 1);
    g->perfect = lodhi_pol(P, perfect, (int)lbar);
    free(perfect);
    g->W64 = (int)((L + 63) / 64);
    for (int st = 0; st < 2; ++st) {
        g->fpeq[st] = (uint64_t*)calloc((size_t)16 * g->W64, sizeof(uint64_t));
        for (uint32_t j = 0; j < L; ++j) {
            uint8_t pc = text_code(g->flank[j]);
            if (st) pc = comp_code(pc);
            for (int code = 0; code < 16; ++code)
                if (pc & code) g->fpeq[st][(size_t)code * g->W64 + (j >> 6)] |= 1ull << (j & 63);
        }
        g->bpeq[st] = NULL;
        if (g->m_bar <= 64) {
            g->bpeq[st] = (uint64_t*)calloc((size_t)n * 16, sizeof(uint64_t));
            const uint8_t* pats = st ? g->pat_rc : g->pat_fwd;
            for (uint32_t s2 = 0; s2 < n; ++s2)
                for (uint32_t j = 0; j < g->m_bar; ++j)
                    for (int code = 0; code < 16; ++code)
                        if (text_code(pats[(size_t)s2 * g->m_bar + j]) & code) g->bpeq[st][(size_t)s2 * 16 + code] |= 1ull << j;
        }
    }
    return BB_OK;
}
static void free_bartab(void* t);   /* bb_oracle_simd.h */
static void group_build_bartab(ogroup* g);
static void free_group(ogroup* g) { free(g->flank); free(g->pat_fwd); free(g->pat_rc); for (int st = 0; st < 2; ++st) { free(g->fpeq[st]); free(g->bpeq[st]); free_bartab(g->bartab[st]); } }

int bbo_create(const bb_group_desc* groups, uint32_t n_groups, const bb_params* params, bbo_ctx** out) {
    return bbo_create_policy(groups, n_groups, params, cur_pol(), out);
}
int bbo_create_policy(const bb_group_desc* groups, uint32_t n_groups, const bb_params* params, const bb_policy* policy, bbo_ctx** out) {
    if (!groups || !params || !out || n_groups == 0) return BB_E_INVALID;
    if (policy && bb_policy_validate(policy)) return BB_E_INVALID;
    bbo_ctx* c = (bbo_ctx*)calloc(1, sizeof(*c));
    c->g = (ogroup*)calloc(n_groups, sizeof(ogroup));
    c->n_groups = n_groups; c->p = *params;
    if (policy) c->pol = *policy; else bb_policy_default(&c->pol);
    for (uint32_t i = 0; i < n_groups; ++i) {
        int rcode = prep_group(&c->pol, &groups[i], &c->g[i]);
        if (rcode != BB_OK) { for (uint32_t k = 0; k <= i; ++k) free_group(&c->g[k]); free(c->g); free(c); return rcode; }
        group_build_bartab(&c->g[i]);
    }
    *out = c;
    return BB_OK;
}
void bbo_destroy(bbo_ctx* c) {
    if (!c) return;
    for (uint32_t i = 0; i < c->n_groups; ++i) free_group(&c->g[i]);
    free(c->g); free(c);
}
int bbo_group_get_info(const bbo_ctx* c, uint32_t gi, bb_group_info* o) {
    if (!c || gi >= c->n_groups || !o) return BB_E_INVALID;
    const ogroup* g = &c->g[gi];
    o->flank_len = g->flank_len; o->prefix_len = g->prefix_len; o->suffix_len = g->suffix_len; o->mask_len = g->mask_len;
    o->bar_lo = g->bar_lo; o->bar_hi = g->bar_hi; o->pad_lo = g->pad_lo; o->pad_hi = g->pad_hi;
    o->pattern_len = g->m_bar; o->flank_k = g->flank_k; o->bar_k1 = g->k1; o->bar_k2 = g->k2; o->perfect_score = g->perfect;
    return BB_OK;
}
int bbo_group_get_flank(const bbo_ctx* c, uint32_t gi, uint8_t* out) {
    if (!c || gi >= c->n_groups || !out) return BB_E_INVALID;
    memcpy(out, c->g[gi].flank, c->g[gi].flank_len);
    return BB_OK;
}
int bbo_group_get_pattern(const bbo_ctx* c, uint32_t gi, uint32_t idx, int rc, uint8_t* out) {
    if (!c || gi >= c->n_groups || !out || idx >= c->g[gi].n_seqs) return BB_E_INVALID;
    const ogroup* g = &c->g[gi];
    memcpy(out, (rc ? g->pat_rc : g->pat_fwd) + (size_t)idx * g->m_bar, g->m_bar);
    return BB_OK;
}

/* ------------------------------------------------------------------------------------------ */
/* [H7] sassy v2 search_encoded_patterns restated for ONE pattern: forward strand only, no      */
/* overhang (regular_searcher = new_rc(), patterns were encoded per strand: barcodes.rs:76,     */
/* searcher.rs:333), every local minimum <= k reported in ascending end position; Barbell keeps */
/* the first strictly-lowest-cost one per pattern (searcher.rs:294-300).                        */
/* returns 1 and fills `best` if the pattern has a match <= k                                   */
/* ------------------------------------------------------------------------------------------ */
static int best_match_for_pattern(const bb_policy* P, const uint8_t* pcode, int m, const uint8_t* wcode, int wn, int k, bbo_match* best) {
    end_list ends = {0, 0, 0};
    scan_strand(P, pcode, m, wcode, wn, k, -1.f, &ends);
    int bi = -1;
    for (int t = 0; t < ends.n; ++t)  /* searcher.rs:294-300; policy [H7]: BB_TIE_LAST = the Vec in descending position order */
        if (bi < 0 || ends.v[t].cost < ends.v[bi].cost || (P->bar_tie == BB_TIE_LAST && ends.v[t].cost == ends.v[bi].cost)) bi = t;
    if (bi >= 0) {
        trace_match(P, pcode, m, wcode, wn, k, -1.f, ends.v[bi].e, ends.v[bi].cost, best);
        best->strand = BB_FWD; best->rc_text_len = wn;
    }
    free(ends.v);
    return bi >= 0;
}

/* ------------------------------------------------------------------------------------------ */
/* Bit-parallel variants of scan_strand and best_match_for_pattern (Myers 1999 / Hyyro 2003, 64-bit words): the SAME    */
/* results, computed a word of DP cells at a time.  They exist for ONE purpose: bench.py's cpu_baseline, so that the    */
/* reported CPU figure is not a scalar O(m n) loop against a reference that runs AVX2 sassy.  They are checked against  */
/* the scalar functions above (tests/test_oracle_fast.py) and are never what a parity test compares the GPU with.        */
/* ------------------------------------------------------------------------------------------ */
#define BBO_MAXW64 4
#define BB_FAST_MAXWIN 160
#include "bb_oracle_simd.h"
/* per-thread scratch of the vector forms (flags of a scan, move planes of a window) */
static __thread uint8_t* t_scan_scratch = NULL; static __thread size_t t_scan_cap = 0;
static __thread __m512i* t_planes = NULL; static __thread size_t t_planes_cap = 0;
static __thread __m512i* t_opsT = NULL; static __thread size_t t_opsT_cap = 0;   /* the lockstep walks' op matrix + one vector of lengths per vector of patterns */

static void free_bartab(void* t) { if (t) { free(((bbo_bartab*)t)->eq); free(t); } }
/* the vector tables of a group's barcodes, built when the context is made (bbo_create_policy) */
static void group_build_bartab(ogroup* g) {
    for (int st = 0; st < 2; ++st) {
        g->bartab[st] = NULL;
        if (!bbo_have_avx512() || !g->bpeq[st]) continue;
        bbo_bartab* t = (bbo_bartab*)calloc(1, sizeof(bbo_bartab));
        bbo_bartab_build(t, g->bpeq[st], g->n_seqs);
        g->bartab[st] = t;
    }
}
/* one strand of the flank scan: end positions by the policy's local-minimum rule, exactly scan_strand's */
static int scan_strand_fast(const bb_policy* P, const uint64_t* peq, int W, int m, const uint8_t* tcode, int n, int reverse, int k, float alpha,
                            end_list* out) {
    uint64_t pv[BBO_MAXW64], mv[BBO_MAXW64];
    if (W > BBO_MAXW64) return 0;
    for (int w = 0; w < W; ++w) { pv[w] = 0; mv[w] = 0; }
    for (int j = 1; j <= m; ++j) {
        const int dlt = alpha >= 0.f ? overhang_cost(P, alpha, j) - overhang_cost(P, alpha, j - 1) : 1;
        if (dlt < 0 || dlt > 1) return 0;                         /* not a 0/1 column: the caller takes the scalar scan */
        if (dlt) pv[(j - 1) >> 6] |= 1ull << ((j - 1) & 63);
    }
    int32_t score = alpha >= 0.f ? overhang_cost(P, alpha, m) : m;
    const int TW = (m - 1) >> 6, TB = (m - 1) & 63;
    lm_state st = {0, 0, 0, 0, P->lm_rule};
    lm_step(&st, 0, score, k, out);
    for (int i = 1; i <= n; ++i) {
        const uint64_t* e = peq + (size_t)tcode[reverse ? n - i : i - 1] * W;
        uint64_t carry = 0, pin = 0, min_ = 0;
        for (int w = 0; w < W; ++w) {
            const uint64_t eq = e[w], x = eq & pv[w];
            const unsigned __int128 sum = (unsigned __int128)x + pv[w] + carry;
            carry = (uint64_t)(sum >> 64);
            const uint64_t d0 = (((uint64_t)sum ^ pv[w]) | eq | mv[w]);
            const uint64_t ph = mv[w] | ~(d0 | pv[w]), mh = pv[w] & d0;
            if (w == TW) score += (int32_t)((ph >> TB) & 1u) - (int32_t)((mh >> TB) & 1u);
            const uint64_t phs = (ph << 1) | pin, mhs = (mh << 1) | min_;
            pin = ph >> 63; min_ = mh >> 63;
            pv[w] = mhs | ~(d0 | phs);
            mv[w] = phs & d0;
        }
        lm_step(&st, i, score, k, out);
    }
    int last = n;
    if (alpha >= 0.f) {
        int32_t d = score;                                        /* D[m-o][n]: walk the last column's vertical deltas upwards */
        for (int o = 1; o <= m; ++o) {
            const int b = m - o;
            d -= (int32_t)((pv[b >> 6] >> (b & 63)) & 1u) - (int32_t)((mv[b >> 6] >> (b & 63)) & 1u);
            lm_step(&st, n + o, d + overhang_cost(P, alpha, o), k, out);
        }
        last = n + m;
    }
    lm_finish(&st, last, k, out);
    return 1;
}
/* trace_match with the window's DP as 64-bit words and the preferred move of every cell as two bit planes (what best_match_for_pattern_fast does for
 * a barcode): at cell (j, i) Match applies iff the characters match (then the diagonal is free), Sub iff the diagonal costs 1, Ins iff the horizontal
 * step into the cell costs 1, Del iff the vertical one does — trace_match's four cost comparisons, read off the column step's vectors.  0: not
 * applicable (overhang costs that are no 0/1 deltas, full-matrix mode): the caller takes trace_match. */
#define BBO_TRACE_FAST_MAXW (64 * BBO_MAXW64 + 128)
static int trace_match_fast(const bb_policy* P, const uint64_t* peq, int W, int m, const uint8_t* tcode, int n, int k, float alpha,
                            int e, int cost, bbo_match* out) {
    if (g_full_trace || W > BBO_MAXW64) return 0;
    const int o = e > n ? e - n : 0;
    const int j0 = m - o, i0 = e > n ? n : e;
    int s0 = i0 - (m + k);
    if (s0 < 0) s0 = 0;
    const int w = i0 - s0;
    if (w > BBO_TRACE_FAST_MAXW) return 0;
    bbo_colstate cs;
    if (!bbo_win_init(P, W, m, alpha, s0 == 0, &cs)) return 0;
    static __thread uint64_t lo[BBO_TRACE_FAST_MAXW + 1][BBO_MAXW64], hi[BBO_TRACE_FAST_MAXW + 1][BBO_MAXW64];
    for (int c = 1; c <= w; ++c) {
        const uint64_t* eqv = peq + (size_t)tcode[s0 + c - 1] * W;
        uint64_t carry = 0, pin = 0, min_ = 0;
        for (int x = 0; x < W; ++x) {
            const uint64_t eq = eqv[x], pv = cs.pv[x], mv = cs.mv[x], xx = eq & pv;
            const unsigned __int128 sum = (unsigned __int128)xx + pv + carry;
            carry = (uint64_t)(sum >> 64);
            const uint64_t d0 = (((uint64_t)sum ^ pv) | eq | mv);
            const uint64_t ph = mv | ~(d0 | pv), mh = pv & d0;
            const uint64_t phs = (ph << 1) | pin, mhs = (mh << 1) | min_;
            pin = ph >> 63; min_ = mh >> 63;
            const uint64_t npv = mhs | ~(d0 | phs);
            cs.pv[x] = npv; cs.mv[x] = phs & d0;
            uint64_t v[4], sp[4] = {0, 0, 0, 0}, taken = 0;
            v[BBO_MATCH] = d0 & eq; v[BBO_SUB] = ~d0; v[BBO_INS] = ph; v[BBO_DEL] = npv;
            for (int q = 0; q < 4; ++q) { const int op = P->trace_prio[q]; const uint64_t y = v[op] & ~taken; taken |= y; sp[op] |= y; }
            lo[c][x] = sp[BBO_SUB] | sp[BBO_DEL]; hi[c][x] = sp[BBO_INS] | sp[BBO_DEL];
        }
    }
    uint8_t rev[64 * BBO_MAXW64 + BBO_TRACE_FAST_MAXW + 2];
    int nops = 0, j = j0, i = w;
    while (j > 0) {
        if (i == 0 && s0 == 0 && alpha >= 0.f) break;            /* left overhang: rest of pattern is outside */
        uint8_t op = BBO_DEL;
        if (i > 0) { const int b = j - 1; op = (uint8_t)(((lo[i][b >> 6] >> (b & 63)) & 1u) | (((hi[i][b >> 6] >> (b & 63)) & 1u) << 1)); }
        rev[nops++] = op;
        if (op != BBO_INS) --j;
        if (op != BBO_DEL) --i;
    }
    out->pattern_start = j; out->pattern_end = j0;
    out->text_start = s0 + i; out->text_end = i0;
    out->cost = cost; out->n_ops = nops;
    out->ops = (uint8_t*)malloc((size_t)(nops ? nops : 1));
    for (int t = 0; t < nops; ++t) out->ops[t] = rev[nops - 1 - t];
    return 1;
}

/* the flank search with bit-parallel scans; matches traced by trace_match like search_pol's */
static int search_fast(const bb_policy* P, const ogroup* g, const uint8_t* tc, int n, float alpha, bbo_match** out) {
    const int m = (int)g->flank_len, k = g->flank_k;
    uint8_t* pc = (uint8_t*)malloc((size_t)m); uint8_t* pcc = (uint8_t*)malloc((size_t)m);
    /* the reversed text, readable BBO_SIMD_PAD bytes either side like the caller's tc (the vector scan's lanes start before the text and run past it) */
    uint8_t* trv_alloc = (uint8_t*)malloc((size_t)n + 2 * BBO_SIMD_PAD);
    uint8_t* trv = trv_alloc + BBO_SIMD_PAD;
    memset(trv_alloc, 0, BBO_SIMD_PAD); memset(trv + n, 0, BBO_SIMD_PAD);
    for (int j = 0; j < m; ++j) { pc[j] = text_code(g->flank[j]); pcc[j] = comp_code(pc[j]); }
    for (int i = 0; i < n; ++i) trv[i] = tc[n - 1 - i];
    end_list ef = {0, 0, 0}, er = {0, 0, 0};
    if (!scan_strand_simd(P, g->fpeq[0], g->W64, m, tc, n, k, alpha, &ef, &t_scan_scratch, &t_scan_cap)) {
        ef.n = 0;
        if (!scan_strand_fast(P, g->fpeq[0], g->W64, m, tc, n, 0, k, alpha, &ef)) { ef.n = 0; scan_strand(P, pc, m, tc, n, k, alpha, &ef); }
    }
    if (!scan_strand_simd(P, g->fpeq[1], g->W64, m, trv, n, k, alpha, &er, &t_scan_scratch, &t_scan_cap)) {
        er.n = 0;
        if (!scan_strand_fast(P, g->fpeq[1], g->W64, m, tc, n, 1, k, alpha, &er)) { er.n = 0; scan_strand(P, pcc, m, trv, n, k, alpha, &er); }
    }
    const int total = ef.n + er.n;
    bbo_match* ms = (bbo_match*)calloc((size_t)(total ? total : 1), sizeof(bbo_match));
    for (int t = 0; t < ef.n; ++t) {
        if (!trace_match_fast(P, g->fpeq[0], g->W64, m, tc, n, k, alpha, ef.v[t].e, ef.v[t].cost, &ms[t]))
            trace_match(P, pc, m, tc, n, k, alpha, ef.v[t].e, ef.v[t].cost, &ms[t]);
        ms[t].strand = BB_FWD; ms[t].rc_text_len = n;
    }
    for (int t = 0; t < er.n; ++t) {
        bbo_match* mm = &ms[ef.n + (P->rc_order == BB_RC_FWD_ORDER ? er.n - 1 - t : t)];
        if (!trace_match_fast(P, g->fpeq[1], g->W64, m, trv, n, k, alpha, er.v[t].e, er.v[t].cost, mm))
            trace_match(P, pcc, m, trv, n, k, alpha, er.v[t].e, er.v[t].cost, mm);
        const int ts = mm->text_start, te = mm->text_end;
        mm->text_start = n - te; mm->text_end = n - ts;
        mm->strand = BB_RC; mm->rc_text_len = n;
        mm->rc_mirror_len = P->rc_path == BB_RCPATH_MIRROR ? m : 0;
    }
    free(ef.v); free(er.v); free(pc); free(pcc); free(trv_alloc);
    *out = ms;
    return total;
}
/* best_match_for_pattern on one 64-bit word (m <= 64, window <= BB_FAST_MAXWIN columns): the forward pass keeps the preferred
 * move of every cell as two bit planes (by default Match: d0 & eq; else Ins: ph; else Sub: ~d0; else Del — trace_match's
 * order, any policy order alike), the walk back reads them */
static int best_match_for_pattern_fast(const bb_policy* P, const uint64_t* peq16, int m, const uint8_t* wcode, int wn, int k, bbo_match* best,
                                       uint8_t* ops_store /* m + wn + 2 bytes of the caller's: no allocation per pattern */) {
    uint64_t lo[BB_FAST_MAXWIN + 1], hi[BB_FAST_MAXWIN + 1];
    uint64_t pv = m >= 64 ? ~0ull : ((1ull << m) - 1ull), mv = 0;
    int32_t score = m, prev = m, best_cost = 0x7FFFFFFF, best_pos = -1, cand = 0;
    int dec = 1;
    const int TB = m - 1;
    for (int c = 1; c <= wn; ++c) {
        const uint64_t eq = peq16[wcode[c - 1]], x = eq & pv;
        const uint64_t d0 = (((x + pv) ^ pv) | eq | mv);
        const uint64_t ph = mv | ~(d0 | pv), mh = pv & d0;
        score += (int32_t)((ph >> TB) & 1u) - (int32_t)((mh >> TB) & 1u);
        const uint64_t phs = ph << 1, mhs = mh << 1;
        pv = mhs | ~(d0 | phs); mv = phs & d0;
        {   /* policy [H3]: per cell the first applicable op of the order — Match: d0 & eq, Sub: ~d0, Ins: ph, Del: the new column's
             * vertical +1 — trace_match's cost compares as bit-vectors */
            uint64_t v[4], s[4] = {0, 0, 0, 0}, taken = 0;
            v[BBO_MATCH] = d0 & eq; v[BBO_SUB] = ~d0; v[BBO_INS] = ph; v[BBO_DEL] = pv;
            for (int q = 0; q < 4; ++q) { const int op = P->trace_prio[q]; const uint64_t x = v[op] & ~taken; taken |= x; s[op] |= x; }
            lo[c] = s[BBO_SUB] | s[BBO_DEL]; hi[c] = s[BBO_INS] | s[BBO_DEL];
        }
        if (score > prev) {                                                            /* lm_step + searcher.rs:294-300 in one */
            if (dec && prev <= k && (prev < best_cost || (P->bar_tie == BB_TIE_LAST && prev == best_cost))) { best_cost = prev; best_pos = P->lm_rule == BB_LM_PLATEAU_LEFT ? cand : c - 1; }
            dec = 0;
        } else if (score < prev) { dec = 1; cand = c; }
        else if (P->lm_rule == BB_LM_STRICT) dec = 0;
        prev = score;
    }
    if (dec && prev <= k && (prev < best_cost || (P->bar_tie == BB_TIE_LAST && prev == best_cost))) { best_cost = prev; best_pos = P->lm_rule == BB_LM_PLATEAU_LEFT ? cand : wn; }
    if (best_pos < 0) return 0;
    uint8_t rev[64 + BB_FAST_MAXWIN + 2];
    int nops = 0, j = m, i = best_pos;
    while (j > 0) {
        uint8_t op = BBO_DEL;
        if (i > 0) op = (uint8_t)(((lo[i] >> (j - 1)) & 1u) | (((hi[i] >> (j - 1)) & 1u) << 1));
        rev[nops++] = op;
        if (op != BBO_INS) --j;
        if (op != BBO_DEL) --i;
    }
    memset(best, 0, sizeof(*best));
    best->pattern_start = 0; best->pattern_end = m; best->text_start = i; best->text_end = best_pos;
    best->cost = best_cost; best->n_ops = nops; best->strand = BB_FWD; best->rc_text_len = wn;
    best->ops = ops_store;
    for (int t = 0; t < nops; ++t) best->ops[t] = rev[nops - 1 - t];
    return 1;
}

typedef struct { bb_row* v; int n, cap; } row_list;
static void row_push(row_list* l, const bb_row* r) {
    if (l->n == l->cap) { l->cap = l->cap ? 2 * l->cap : 8; l->v = (bb_row*)realloc(l->v, sizeof(bb_row) * (size_t)l->cap); }
    l->v[l->n++] = *r;
}

/* searcher.rs:241-265 */
static void push_flank_only(row_list* rows, uint32_t read_idx, uint32_t read_len, uint32_t gi, const ogroup* g, const bbo_match* fm) {
    bb_row r; memset(&r, 0, sizeof(r));
    r.read_idx = read_idx; r.read_len = read_len;
    r.rel_dist_to_end = bbo_rel_dist_to_end(fm->text_start, read_len);
    r.read_start_bar = (uint32_t)fm->text_start; r.read_end_bar = (uint32_t)fm->text_end;
    r.read_start_flank = (uint32_t)fm->text_start; r.read_end_flank = (uint32_t)fm->text_end;
    r.bar_start = 0; r.bar_end = 0;
    r.match_type = g->type == BB_FTAG ? BB_FFLANK : BB_RFLANK;
    r.flank_cost = (int16_t)fm->cost; r.barcode_cost = (int16_t)g->m_bar; /* barcodes[0].seq.len() */
    r.barcode_idx = -1; r.group_idx = (uint8_t)gi; r.strand = (uint8_t)fm->strand;
    row_push(rows, &r);
}

/* Diagnostics for tools/policy_feasible.py: the reference's own invariants on this path, COUNTED instead of aborting — a policy setting
 * under which real Barbell would panic (searcher.rs:388 expect, the slice at :456) or put the barcode window off the barcode of its own
 * documented examples cannot be what the real crates do.  Off (NULL) in every other entry point. */
static __thread bbo_diag* t_diag = NULL;
static __thread const int32_t* t_truth = NULL;   /* BBO_TRUTH_PER_READ x BBO_TRUTH_FIELDS int32 of the read in hand */
/* the planted construct (same group and strand) a flank match lies on: at least half of the construct's span is inside the match */
static const int32_t* diag_on_target(uint32_t gi, const bbo_match* fm) {
    if (!t_truth) return NULL;
    for (int t = 0; t < BBO_TRUTH_PER_READ; ++t) {
        const int32_t* tr = t_truth + t * BBO_TRUTH_FIELDS;   /* group, strand, construct_lo, construct_hi, bar_lo, bar_hi, barcode idx */
        if (tr[0] != (int32_t)gi || tr[1] != fm->strand) continue;
        const int lo = fm->text_start > tr[2] ? fm->text_start : tr[2], hi = fm->text_end < tr[3] ? fm->text_end : tr[3];
        if (2 * (hi - lo) >= tr[3] - tr[2]) return tr;
    }
    return NULL;
}

/* Demuxer::demux (searcher.rs:430-490) for one read; rows appended to `rows` (already collapsed) */
static void demux_read(const bbo_ctx* c, uint32_t read_idx, const uint8_t* read, uint32_t n, row_list* rows, int fast) {
    int first_row = rows->n;
    uint8_t* rcode_alloc = (uint8_t*)malloc((size_t)n + 2 * BBO_SIMD_PAD);   /* (padded: scan_strand_simd) */
    uint8_t* rcode = rcode_alloc + BBO_SIMD_PAD;
    memset(rcode_alloc, 0, BBO_SIMD_PAD); memset(rcode + n, 0, BBO_SIMD_PAD);
    { const uint8_t* T = text_code_table(); for (uint32_t i = 0; i < n; ++i) rcode[i] = T[read[i]]; }
    for (uint32_t gi = 0; gi < c->n_groups; ++gi) {                                   /* :433 */
        const ogroup* g = &c->g[gi];
        bbo_match* fms = NULL;
        int nfm = fast && g->W64 <= BBO_MAXW64 ? search_fast(&c->pol, g, rcode, (int)n, c->p.alpha, &fms)
                                                 : search_pol(&c->pol, g->flank, (int)g->flank_len, read, (int)n, g->flank_k, c->p.alpha, 1, &fms); /* :438 */
        if (t_diag) t_diag->flank_matches += (uint64_t)nfm;
        for (int f = 0; f < nfm; ++f) {                                               /* :440 */
            const bbo_match* fm = &fms[f];
            int lo, hi;
            const int32_t* on = t_diag ? diag_on_target(gi, fm) : NULL;
            if (on) t_diag->on_target++;
            if (!bbo_get_matching_region(fm, (int)g->bar_lo, (int)g->bar_hi, &lo, &hi)) { if (t_diag) t_diag->region_none++; continue; } /* :445-449 */
            uint32_t ws = lo >= PADDING ? (uint32_t)(lo - PADDING) : 0;                /* :453 */
            uint32_t we = (uint32_t)(hi + PADDING) < n ? (uint32_t)(hi + PADDING) : n; /* :454 */
            if (we < ws) { we = ws; if (t_diag) t_diag->slice_panic++; } /* the reference would panic on the slice; cannot happen for lo <= n */
            if (on) {
                t_diag->window_overlaps += (int32_t)ws < on[5] && (int32_t)we > on[4];
                t_diag->window_covers += (int32_t)ws <= on[4] && (int32_t)we >= on[5];
            }
            const uint8_t* wcode = rcode + ws; int wn = (int)(we - ws);
            const uint8_t* pats = fm->strand == BB_FWD ? g->pat_fwd : g->pat_rc;       /* barcodes.rs:97-102 */
            int m = (int)g->m_bar;
            /* per-thread buffers, grown once (a hit of a 96-barcode kit used to cost four allocations and 7 KB of zeroing) */
            static __thread bbo_match* t_best = NULL; static __thread uint8_t* t_has = NULL; static __thread double* t_sc = NULL; static __thread uint32_t t_cand_cap = 0;
            static __thread uint8_t* t_pcode = NULL; static __thread int t_pcode_cap = 0;
            if (t_cand_cap < g->n_seqs) {
                free(t_best); free(t_has); free(t_sc);
                t_cand_cap = g->n_seqs + 8;
                t_best = (bbo_match*)calloc(t_cand_cap, sizeof(bbo_match)); t_has = (uint8_t*)malloc(t_cand_cap); t_sc = (double*)malloc(sizeof(double) * t_cand_cap);
            }
            if (t_pcode_cap < m) { free(t_pcode); t_pcode_cap = m + 16; t_pcode = (uint8_t*)malloc((size_t)t_pcode_cap); }
            bbo_match* best = t_best; uint8_t* has = t_has; uint8_t* pcode = t_pcode;
            memset(has, 0, g->n_seqs);
            const int fast_bar = fast && g->bpeq[fm->strand] && wn <= BB_FAST_MAXWIN;
            const size_t ops_stride = (size_t)(m + wn + 2);
            /* the candidates' op strings, one block (the vector form keeps its walks as a matrix instead: bt below) */
            uint8_t* ops_arena = fast_bar && !g->bartab[fm->strand] ? (uint8_t*)malloc(ops_stride * g->n_seqs) : NULL;
            int k = g->k1, matched = 0;
            const bbo_bartab* bt = fast_bar ? (const bbo_bartab*)g->bartab[fm->strand] : NULL;
            const size_t ops_cap = 64 + BB_FAST_MAXWIN + 2;
            if (bt) {
                const size_t need = (size_t)bt->n_vec * 2 * (BB_FAST_MAXWIN + 1);
                if (t_planes_cap < need) { free(t_planes); t_planes = (__m512i*)aligned_alloc(64, need * sizeof(__m512i)); t_planes_cap = need; }
                const size_t need2 = (size_t)bt->n_vec * (ops_cap + 1);
                if (t_opsT_cap < need2) { free(t_opsT); t_opsT = (__m512i*)aligned_alloc(64, need2 * sizeof(__m512i)); t_opsT_cap = need2; }
            }
            __m512i* const nopsT = bt ? t_opsT + (size_t)bt->n_vec * ops_cap : NULL;
            for (int pass = 0; pass < 2; ++pass) {                                    /* :282-328 */
                matched = 0;
                if (bt) {   /* every pattern of the group at once, 8 per vector (bb_oracle_simd.h) */
                    matched = best_matches_simd(&c->pol, bt, g->n_seqs, m, wcode, wn, k, best, has, ops_arena, ops_stride, t_planes, t_opsT, ops_cap, nopsT);
                    if (matched <= 1 && g->k1 < g->k2 && pass == 0) { k = g->k2; continue; }
                    break;
                }
                for (uint32_t p = 0; p < g->n_seqs; ++p) {
                    for (int j = 0; j < m; ++j) pcode[j] = text_code(pats[(size_t)p * m + j]);
                    if (has[p]) { if (!fast_bar) free(best[p].ops); best[p].ops = NULL; has[p] = 0; }
                    if (fast_bar)
                        has[p] = (uint8_t)best_match_for_pattern_fast(&c->pol, g->bpeq[fm->strand] + (size_t)p * 16, m, wcode, wn, k, &best[p],
                                                                      ops_arena + ops_stride * p);
                    else
                        has[p] = (uint8_t)best_match_for_pattern(&c->pol, pcode, m, wcode, wn, k, &best[p]);
                    matched += has[p];
                }
                if (matched <= 1 && g->k1 < g->k2 && pass == 0) k = g->k2; else break; /* :303-306 */
            }
            if (matched == 0) {                                                       /* :353-362 */
                push_flank_only(rows, read_idx, n, gi, g, fm);
            } else {
                /* :364-377 score every candidate, stable sort descending by s_norm: top = first
                 * maximum in ascending idx order, second = best of the rest */
                int top = -1, second = -1; double top_s = 0, second_s = 0;
                double* sc = t_sc;
                if (bt && c->pol.lodhi_p <= 5) {   /* the scores of a vector's eight candidates at a time, the same f64 operations each (lodhi_polT) */
                    for (int v = 0; v < bt->n_vec; ++v) {
                        double sv[8];
                        lodhi_polT(&c->pol, t_opsT + (size_t)v * ops_cap, &nopsT[v], sv);
                        for (int l = 0; l < 8 && (uint32_t)(8 * v + l) < g->n_seqs; ++l)
                            if (has[8 * v + l]) sc[8 * v + l] = g->perfect > 0.0 ? sv[l] / g->perfect : 0.0;
                    }
                }
                for (uint32_t p = 0; p < g->n_seqs; ++p) {
                    if (!has[p]) continue;
                    if (!bt || c->pol.lodhi_p > 5) {
                        uint8_t tmp_ops[64 + BB_FAST_MAXWIN + 2];
                        if (bt) bbo_ops_from_T(t_opsT, ops_cap, p, best[p].n_ops, tmp_ops);
                        double s = lodhi_pol(&c->pol, bt ? tmp_ops : best[p].ops, best[p].n_ops);
                        sc[p] = g->perfect > 0.0 ? s / g->perfect : 0.0;               /* :368-372 */
                    }
                    if (top < 0 || sc[p] > top_s) { top = (int)p; top_s = sc[p]; }
                }
                for (uint32_t p = 0; p < g->n_seqs; ++p) {
                    if (!has[p] || (int)p == top) continue;
                    if (second < 0 || sc[p] > second_s) { second = (int)p; second_s = sc[p]; }
                }
                uint8_t top_ops[64 + BB_FAST_MAXWIN + 2];
                if (bt) { bbo_ops_from_T(t_opsT, ops_cap, (uint32_t)top, best[top].n_ops, top_ops); best[top].ops = top_ops; }
                int rel_lo = (int)(g->bar_lo - g->pad_lo), rel_hi = (int)(g->bar_hi - g->pad_lo); /* :379-382 */
                int pl, ph, tl, th, bc;
                int panicked = 0;
                if (!bbo_map_pat_to_text_with_cost(&best[top], rel_lo, rel_hi, &pl, &ph, &tl, &th, &bc)) {
                    if (!t_diag) { fprintf(stderr, "bb_oracle: No barcode match region found; unusual\n"); abort(); } /* :388 */
                    t_diag->subpath_none++; panicked = 1;
                }
                int valid = !panicked && top_s >= c->p.min_score;                      /* :391-396 */
                if (second >= 0) valid = valid && (top_s - second_s) >= c->p.min_score_diff;
                if (valid && on) { t_diag->tag_rows_on_target++; t_diag->tag_rows_correct += top == on[6]; }
                if (valid) {                                                           /* :398-416 */
                    bb_row r; memset(&r, 0, sizeof(r));
                    r.read_idx = read_idx; r.read_len = n;
                    r.rel_dist_to_end = bbo_rel_dist_to_end(fm->text_start, n);
                    r.read_start_bar = ws + (uint32_t)tl; r.read_end_bar = ws + (uint32_t)th;
                    r.read_start_flank = (uint32_t)fm->text_start; r.read_end_flank = (uint32_t)fm->text_end;
                    r.bar_start = ws + (uint32_t)pl; r.bar_end = ws + (uint32_t)ph;
                    r.match_type = g->type;
                    r.flank_cost = (int16_t)fm->cost; r.barcode_cost = (int16_t)bc;
                    r.barcode_idx = (int16_t)top; r.group_idx = (uint8_t)gi; r.strand = (uint8_t)fm->strand;
                    row_push(rows, &r);
                } else {
                    push_flank_only(rows, read_idx, n, gi, g, fm);                     /* :417-425 */
                }
            }
            if (!fast_bar) for (uint32_t p = 0; p < g->n_seqs; ++p) if (has[p]) free(best[p].ops);
            free(ops_arena);
        }
        bbo_free_matches(fms, nfm);
    }
    free(rcode_alloc);
    rows->n = first_row + bbo_collapse(rows->v + first_row, rows->n - first_row, 0.8f); /* :489 */
}

static int annotate_batch_impl(bbo_ctx* c, const uint8_t* bases, const uint64_t* offsets, uint32_t n_reads,
                               bb_row* rows, uint64_t rows_cap, uint64_t* n_rows, int n_threads, int fast);
int bbo_annotate_batch(bbo_ctx* c, const uint8_t* bases, const uint64_t* offsets, uint32_t n_reads,
                       bb_row* rows, uint64_t rows_cap, uint64_t* n_rows, int n_threads) {
    return annotate_batch_impl(c, bases, offsets, n_reads, rows, rows_cap, n_rows, n_threads, 0);
}
/* 1: bbo_annotate_batch_fast runs the AVX-512 forms of the flank scan and the barcode pass (bb_oracle_simd.h) on this CPU; 0: 64-bit words */
int bbo_fast_is_simd(void) { return bbo_have_avx512(); }
int bbo_annotate_batch_fast(bbo_ctx* c, const uint8_t* bases, const uint64_t* offsets, uint32_t n_reads,
                            bb_row* rows, uint64_t rows_cap, uint64_t* n_rows, int n_threads) {
    return annotate_batch_impl(c, bases, offsets, n_reads, rows, rows_cap, n_rows, n_threads, 1);
}
static int annotate_batch_impl(bbo_ctx* c, const uint8_t* bases, const uint64_t* offsets, uint32_t n_reads,
                               bb_row* rows, uint64_t rows_cap, uint64_t* n_rows, int n_threads, int fast) {
    if (!c || (!bases && n_reads) || !offsets || !n_rows) return BB_E_INVALID;
    row_list* per = (row_list*)calloc(n_reads ? n_reads : 1, sizeof(row_list));
#ifdef _OPENMP
    if (n_threads < 1) n_threads = 1;
    /* One worker per CPU, pinned for the region (round 5): left to the scheduler the workers of this loop migrate, and with them the
     * per-read working set — measured on the 8-CPU build box 3.8 k reads/s on 8 floating threads against 30 k pinned (4.4 k on one thread),
     * i.e. the "CPU baseline" of rounds 1-4 was an eighth of what the same code does.  The calling thread's mask is restored afterwards. */
    {   /* the per-read scratch (a few KB to tens of KB, malloc'd and freed per read and per hit) makes every thread's arena grow and trim
         * all the time: mprotect / madvise under the process's mapping lock — the loop stopped scaling at 16 threads on the 256-thread
         * GPU box (112 k reads/s at 16, 51 k at 256).  Arenas that keep what they have freed do not go to the kernel. */
        static int tuned = 0;
        if (!tuned) { tuned = 1; mallopt(M_TRIM_THRESHOLD, 1 << 30); mallopt(M_TOP_PAD, 16 << 20); mallopt(M_MMAP_THRESHOLD, 1 << 30); }
    }
    cpu_set_t allowed, caller;
    const int have_mask = n_threads > 1 && g_pin_threads && sched_getaffinity(0, sizeof(allowed), &allowed) == 0;
    int cpus[1024], n_cpus = 0;
    if (have_mask) { caller = allowed; for (int q = 0; q < CPU_SETSIZE && n_cpus < 1024; ++q) if (CPU_ISSET(q, &allowed)) cpus[n_cpus++] = q; }
#pragma omp parallel num_threads(n_threads)
    {
        if (have_mask && n_cpus > 0) {
            cpu_set_t one; CPU_ZERO(&one); CPU_SET(cpus[omp_get_thread_num() % n_cpus], &one);
            (void)sched_setaffinity(0, sizeof(one), &one);
        }
#pragma omp for schedule(dynamic, 16)
        for (long i = 0; i < (long)n_reads; ++i)
            demux_read(c, (uint32_t)i, bases + offsets[i], (uint32_t)(offsets[i + 1] - offsets[i]), &per[i], fast);
        if (have_mask && n_cpus > 0) (void)sched_setaffinity(0, sizeof(allowed), &allowed);   /* workers and caller alike: free to move again */
    }
    if (have_mask) (void)sched_setaffinity(0, sizeof(caller), &caller);
#else
    for (long i = 0; i < (long)n_reads; ++i)
        demux_read(c, (uint32_t)i, bases + offsets[i], (uint32_t)(offsets[i + 1] - offsets[i]), &per[i], fast);
#endif
    (void)n_threads;
    uint64_t total = 0;
    for (uint32_t i = 0; i < n_reads; ++i) total += (uint64_t)per[i].n;
    *n_rows = total;
    int rcode = BB_OK;
    if (total > rows_cap || (!rows && total)) rcode = BB_E_CAPACITY;
    else {
        uint64_t o = 0;
        for (uint32_t i = 0; i < n_reads; ++i) { if (per[i].n) memcpy(rows + o, per[i].v, sizeof(bb_row) * (size_t)per[i].n); o += (uint64_t)per[i].n; }
    }
    for (uint32_t i = 0; i < n_reads; ++i) free(per[i].v);
    free(per);
    return rcode;
}

/* the same per-read procedure with the invariants counted (tools/policy_feasible.py); rows are discarded */
int bbo_annotate_diag(bbo_ctx* c, const uint8_t* bases, const uint64_t* offsets, uint32_t n_reads, const int32_t* truth, int n_threads, int fast,
                      bbo_diag* out) {
    if (!c || (!bases && n_reads) || !offsets || !out) return BB_E_INVALID;
    memset(out, 0, sizeof(*out));
#ifdef _OPENMP
    if (n_threads < 1) n_threads = 1;
#pragma omp parallel num_threads(n_threads)
#endif
    {
        bbo_diag mine; memset(&mine, 0, sizeof(mine));
        t_diag = &mine;
#ifdef _OPENMP
#pragma omp for schedule(dynamic, 16)
#endif
        for (long i = 0; i < (long)n_reads; ++i) {
            row_list rl; memset(&rl, 0, sizeof(rl));
            t_truth = truth ? truth + (size_t)i * BBO_TRUTH_PER_READ * BBO_TRUTH_FIELDS : NULL;
            demux_read(c, (uint32_t)i, bases + offsets[i], (uint32_t)(offsets[i + 1] - offsets[i]), &rl, fast);
            mine.rows += (uint64_t)rl.n;
            free(rl.v);
        }
        t_diag = NULL; t_truth = NULL;
#ifdef _OPENMP
#pragma omp critical
#endif
        {
            const uint64_t* a = (const uint64_t*)&mine; uint64_t* o = (uint64_t*)out;
            for (size_t k = 0; k < sizeof(bbo_diag) / sizeof(uint64_t); ++k) o[k] += a[k];
        }
    }
    (void)n_threads;
    return BB_OK;
}

/* ------------------------------------------------------------------------------------------ */
/* filter step: pattern.rs:96-240 + filter.rs:183-214 restated on bb_row       */
/* ------------------------------------------------------------------------------------------ */
static uint32_t row_slot(const bbo_ctx* c, const bb_row* r) {
    uint32_t off = 0;
    for (uint32_t g = 0; g < r->group_idx; ++g) off += c->g[g].n_seqs + 1;
    return off + (r->barcode_idx >= 0 ? (uint32_t)r->barcode_idx : c->g[r->group_idx].n_seqs);
}
/* match_pattern (pattern.rs:205-240): element e must match row e; returns 1 on match */
static int match_pattern_rows(const bbo_ctx* c, const bb_row* rows, uint32_t n, const bb_pattern* p, const uint32_t* label_ids) {
    if (n < p->n_elems) return 0;                                            /* :212-214 */
    int64_t prev_end = 0; int have_prev = 0;
    int32_t ph_key[64]; uint32_t ph_label[64]; int n_ph = 0;
    for (uint32_t e = 0; e < p->n_elems; ++e) {
        const bb_pattern_elem* el = &p->elems[e];
        const bb_row* m = &rows[e];                                          /* current_match_idx == e */
        const uint32_t slot = row_slot(c, m);
        /* check_match_type_and_label :96-126 */
        if (m->match_type != el->match_type) return 0;
        if ((m->match_type == BB_FTAG || m->match_type == BB_RTAG) && el->label_ok && !el->label_ok[slot]) return 0;
        /* check_placeholder :128-145 */
        if (el->placeholder >= 0) {
            int found = -1;
            for (int q = 0; q < n_ph; ++q) if (ph_key[q] == el->placeholder) found = q;
            if (found >= 0) { if (ph_label[found] != label_ids[slot]) return 0; }
            else if (n_ph < 64) { ph_key[n_ph] = el->placeholder; ph_label[n_ph] = label_ids[slot]; ++n_ph; }
        }
        /* check_orientation :147-149 */
        if (el->orientation >= 0 && el->orientation != (int8_t)m->strand) return 0;
        /* check_relative_position :151-190 */
        const int64_t m_start = m->read_start_bar, m_end = m->read_end_bar, seq_len = m->read_len;
        if (el->relative_to == BB_REL_LEFT) {
            if (m_start < el->range_lo || m_start > el->range_hi) return 0;
        } else if (el->relative_to == BB_REL_RIGHT) {
            if (m_end < seq_len - el->range_hi || m_end > seq_len - el->range_lo) return 0;
        } else if (el->relative_to == BB_REL_PREV_LEFT) {
            if (have_prev && (m_start < prev_end + el->range_lo || m_start > prev_end + el->range_hi)) return 0;
        }
        prev_end = m_end; have_prev = 1;                                      /* :229 */
    }
    return 1;
}
int bbo_filter_rows(const bbo_ctx* c, const bb_pattern* patterns, uint32_t n_patterns, const uint32_t* label_ids,
                    const bb_row* rows, uint64_t n_rows, bb_row_verdict* out) {
    if (!c || (!patterns && n_patterns) || !label_ids || (!rows && n_rows) || (!out && n_rows)) return BB_E_INVALID;
    uint64_t i = 0;
    while (i < n_rows) {
        uint64_t j = i;
        while (j < n_rows && rows[j].read_idx == rows[i].read_idx) ++j;      /* group of one read (filter.rs:54-85) */
        const uint32_t n = (uint32_t)(j - i);
        uint32_t max_matches = 0; const bb_pattern* best = NULL;              /* check_filter_pass filter.rs:183-214 */
        for (uint32_t p = 0; p < n_patterns; ++p)
            if (match_pattern_rows(c, rows + i, n, &patterns[p], label_ids) && patterns[p].n_elems > max_matches) {
                max_matches = patterns[p].n_elems; best = &patterns[p];
            }
        for (uint32_t r = 0; r < n; ++r) {
            bb_row_verdict* v = &out[i + r];
            memset(v, 0, sizeof(*v));
            v->pass = max_matches == n;
            v->match_idx = (uint16_t)r;
            if (best && r < best->n_elems) {
                v->n_cuts = best->elems[r].n_cuts;
                for (uint32_t q = 0; q < v->n_cuts && q < BB_MAX_CUTS; ++q) v->cuts[q] = best->elems[r].cuts[q];
            }
        }
        i = j;
    }
    return BB_OK;
}

/* ------------------------------------------------------------------------------------------ */
/* trim/split step: trim.rs:127-300 + the record text of trim.rs:447-460       */
/* ------------------------------------------------------------------------------------------ */
typedef struct { uint32_t gid, start, end; uint8_t after; uint32_t row; } cut_entry;   /* (start_flank, end_flank, cut, anno) */
typedef struct { uint32_t gid, first, n; } cut_group;                                   /* entries of one group id, in row order */
typedef struct { uint32_t start, end; int32_t a[2]; int n_a; } complete_slice;

/* H10: the reference collects groups in a HashMap and stable-sorts them by the start of their first
 * entry (trim.rs:145-152); ties keep the map's (random) iteration order.  Here ties keep the order
 * in which the group ids first appear in the rows. */
static int preprocess_cuts(const bb_row* rows, const bb_row_verdict* v, uint32_t n, uint32_t seq_len, complete_slice** out) {
    uint32_t n_e = 0;
    for (uint32_t r = 0; r < n; ++r) n_e += v[r].n_cuts;
    cut_entry* e = (cut_entry*)malloc(sizeof(cut_entry) * (n_e + 1));
    cut_group* g = (cut_group*)malloc(sizeof(cut_group) * (n_e + 1));
    uint32_t* member = (uint32_t*)malloc(sizeof(uint32_t) * (n_e + 1));   /* entries ordered by group, row order inside */
    uint32_t k = 0, n_g = 0;
    for (uint32_t r = 0; r < n; ++r)
        for (uint32_t q = 0; q < v[r].n_cuts; ++q) {
            e[k].gid = v[r].cuts[q].group_id; e[k].start = rows[r].read_start_flank; e[k].end = rows[r].read_end_flank;
            e[k].after = v[r].cuts[q].direction == BB_CUT_AFTER; e[k].row = r; ++k;
        }
    for (uint32_t i = 0; i < n_e; ++i) {                                   /* groups in first-appearance order */
        uint32_t j = 0;
        while (j < n_g && g[j].gid != e[i].gid) ++j;
        if (j == n_g) { g[n_g].gid = e[i].gid; g[n_g].n = 0; ++n_g; }
        g[j].n++;
    }
    uint32_t pos = 0;
    for (uint32_t j = 0; j < n_g; ++j) {
        g[j].first = pos;
        for (uint32_t i = 0; i < n_e; ++i) if (e[i].gid == g[j].gid) member[pos++] = i;
    }
    for (uint32_t i = 1; i < n_g; ++i) {                                    /* stable insertion sort by first entry's start */
        cut_group t = g[i]; uint32_t j = i;
        while (j > 0 && e[member[g[j - 1].first]].start > e[member[t.first]].start) { g[j] = g[j - 1]; --j; }
        g[j] = t;
    }
    complete_slice* s = (complete_slice*)malloc(sizeof(complete_slice) * (n_g + 1));
    uint32_t n_s = 0;
    for (uint32_t i = 0; i < n_g; ++i) {
        const cut_entry* m0 = &e[member[g[i].first]];
        if (g[i].n == 2) {                                                  /* trim.rs:156-181 */
            const cut_entry* m1 = &e[member[g[i].first + 1]];
            s[n_s].start = m0->after ? m0->end : m0->start;
            s[n_s].end = m1->after ? m1->end : m1->start;
            s[n_s].a[0] = (int32_t)m0->row; s[n_s].a[1] = (int32_t)m1->row; s[n_s].n_a = 2; ++n_s;
        } else if (g[i].n == 1) {
            if (!m0->after) {                                               /* Before: look left, trim.rs:186-215 */
                uint32_t st = 0; int32_t left = -1;
                if (i > 0) {
                    const cut_group* pg = &g[i - 1]; uint32_t best = 0;
                    for (uint32_t q = 0; q < pg->n; ++q)                    /* max_by_key: last maximum */
                        if (e[member[pg->first + q]].end >= e[member[pg->first + best]].end) best = q;
                    st = e[member[pg->first + best]].end; left = (int32_t)e[member[pg->first + best]].row;
                }
                s[n_s].start = st; s[n_s].end = m0->start; s[n_s].n_a = 0;
                if (left >= 0) s[n_s].a[s[n_s].n_a++] = left;
                s[n_s].a[s[n_s].n_a++] = (int32_t)m0->row; ++n_s;
            } else {                                                        /* After: look right, trim.rs:217-248 */
                uint32_t en = seq_len; int32_t right = -1;
                if (i + 1 < n_g) {
                    const cut_group* ng = &g[i + 1]; uint32_t best = 0;
                    for (uint32_t q = 0; q < ng->n; ++q)                    /* min_by_key: first minimum */
                        if (e[member[ng->first + q]].start < e[member[ng->first + best]].start) best = q;
                    en = e[member[ng->first + best]].start; right = (int32_t)e[member[ng->first + best]].row;
                }
                s[n_s].start = m0->end; s[n_s].end = en; s[n_s].n_a = 0;
                s[n_s].a[s[n_s].n_a++] = (int32_t)m0->row;
                if (right >= 0) s[n_s].a[s[n_s].n_a++] = right;
                ++n_s;
            }
        }                                                                   /* other sizes: no slice */
    }
    free(e); free(g); free(member);
    *out = s;
    return (int)n_s;
}

/* LabelConfig::create_label (trim.rs:58-105) as a key: parts = label_id*2 + strand bit */
static uint32_t label_key_of(const bbo_ctx* c, const bb_trim_config* cfg, const uint8_t* is_flank, const uint32_t* part_rank,
                             const uint32_t* label_ids, const bb_row* rows, const complete_slice* s) {
    if (!cfg->add_labels) return 0;
    uint32_t parts[2]; int np = 0;
    for (int i = 0; i < s->n_a; ++i) {
        const bb_row* m = &rows[s->a[i]];
        const uint32_t id = label_ids[row_slot(c, m)];
        if (!cfg->add_flank && is_flank[id]) continue;
        parts[np++] = id * 2u + (cfg->add_orientation ? (m->strand == BB_RC) : 0u);
    }
    if (np == 0) return 0;
    if (cfg->sort_labels) {
        if (np == 2 && part_rank[parts[1]] < part_rank[parts[0]]) { uint32_t t = parts[0]; parts[0] = parts[1]; parts[1] = t; }
    } else if (cfg->only_side != BB_SIDE_NONE) {
        parts[0] = cfg->only_side == BB_SIDE_LEFT ? parts[0] : parts[np - 1];
        np = 1;
    }
    return (parts[0] + 1u) << 16 | (np == 2 ? parts[1] + 1u : 0u);
}

static uint8_t comp_base(uint8_t ch) {                                     /* the RC table of trim.rs:486-530 */
    static const char* pairs = "ATCGRYKMBVDH";                            /* A<->T C<->G R<->Y K<->M B<->V D<->H; S W N X fixed */
    for (int i = 0; pairs[i]; ++i) {
        if (ch == (uint8_t)pairs[i]) return (uint8_t)pairs[i ^ 1];
        if (ch == (uint8_t)(pairs[i] + 32)) return (uint8_t)(pairs[i ^ 1] + 32);
    }
    return ch;
}
static int n_digits(uint32_t v) { int d = 1; while (v >= 10) { v /= 10; ++d; } return d; }

typedef struct { bb_slice s; uint64_t order; } slice_rec;
static int cmp_slice_rec(const void* a, const void* b) {
    const slice_rec* x = (const slice_rec*)a; const slice_rec* y = (const slice_rec*)b;
    if (x->s.label_key != y->s.label_key) return x->s.label_key < y->s.label_key ? -1 : 1;
    return x->order < y->order ? -1 : (x->order > y->order);
}

int bbo_trim_batch(const bbo_ctx* c, const bb_trim_config* cfg, const uint8_t* label_is_flank, const uint32_t* part_rank,
                   const uint32_t* label_ids, const bb_row* rows, const bb_row_verdict* verdicts, uint64_t n_rows,
                   const uint8_t* bases, const uint8_t* quals, const uint64_t* offsets, const bb_headers* h, uint32_t n_reads,
                   uint8_t* text, uint64_t text_cap, uint64_t* text_len, bb_slice* slices, uint64_t slices_cap, uint64_t* n_slices,
                   bb_label_span* spans, uint32_t spans_cap, uint32_t* n_spans, uint8_t* read_status) {
    if (!c || !cfg || !label_ids || !text_len || !n_slices || !n_spans || !read_status || !h) return BB_E_INVALID;
    if (cfg->sort_labels && cfg->only_side != BB_SIDE_NONE) return BB_E_INVALID;             /* trim.rs:330-334 */
    memset(read_status, BB_TRIM_NONE, n_reads);
    slice_rec* recs = NULL; uint64_t n_rec = 0, cap_rec = 0;
    uint64_t i = 0;
    while (i < n_rows) {
        uint64_t j = i;
        while (j < n_rows && rows[j].read_idx == rows[i].read_idx) ++j;
        const uint32_t read = rows[i].read_idx, n = (uint32_t)(j - i);
        if (read >= n_reads) { free(recs); return BB_E_INVALID; }
        if (verdicts[i].pass) {                                             /* filtered.tsv holds passing reads only */
            const uint32_t seq_len = (uint32_t)(offsets[read + 1] - offsets[read]);
            complete_slice* sl = NULL;
            const int n_s = preprocess_cuts(rows + i, verdicts + i, n, seq_len, &sl);
            int written = 0;
            for (int q = 0; q < n_s; ++q) {                                 /* process_read_and_anno trim.rs:270-297 */
                if (sl[q].start >= sl[q].end) continue;
                if (n_rec == cap_rec) { cap_rec = cap_rec ? cap_rec * 2 : 1024; recs = (slice_rec*)realloc(recs, cap_rec * sizeof(slice_rec)); }
                bb_slice* o = &recs[n_rec].s;
                memset(o, 0, sizeof(*o));
                o->read_idx = read; o->start = sl[q].start; o->end = sl[q].end; o->suffix = (uint16_t)q;
                o->label_key = label_key_of(c, cfg, label_is_flank, part_rank, label_ids, rows + i, &sl[q]);
                if (cfg->flip)
                    for (int a = 0; a < sl[q].n_a; ++a) {                   /* should_flip trim.rs:310-315 */
                        const bb_row* m = &rows[i + (uint32_t)sl[q].a[a]];
                        if (m->match_type == BB_FTAG && m->strand == BB_RC) o->flip = 1;
                    }
                const uint64_t hl = h->hdr_offsets[read + 1] - h->hdr_offsets[read];
                const uint32_t L = cfg->skip_trim ? seq_len : o->end - o->start;
                const uint32_t desc_len = (uint32_t)(hl - h->desc_start[read]);
                o->rec_len = 1 + h->id_len[read] + (q ? 1 + (uint32_t)n_digits((uint32_t)q) : 0) +
                             ((cfg->write_full_header && desc_len) ? 1 + desc_len : 0) + 1 + L + 3 + L + 1;
                recs[n_rec].order = n_rec; ++n_rec; ++written;
            }
            free(sl);
            read_status[read] = written ? BB_TRIM_TRIMMED : BB_TRIM_FAILED;
        }
        i = j;
    }
    if (n_rec) qsort(recs, n_rec, sizeof(slice_rec), cmp_slice_rec);       /* group by label, read order inside */
    uint64_t total = 0; uint32_t ns = 0;
    for (uint64_t r = 0; r < n_rec; ++r) {
        recs[r].s.out_off = total; total += recs[r].s.rec_len;
        if (r == 0 || recs[r].s.label_key != recs[r - 1].s.label_key) ++ns;
    }
    *text_len = total; *n_slices = n_rec; *n_spans = ns;
    if (total > text_cap || n_rec > slices_cap || ns > spans_cap) { free(recs); return BB_E_CAPACITY; }
    ns = 0;
    for (uint64_t r = 0; r < n_rec; ++r) {
        const bb_slice* o = &recs[r].s;
        slices[r] = *o;
        if (r == 0 || o->label_key != recs[r - 1].s.label_key) {
            spans[ns].label_key = o->label_key; spans[ns].n_records = 0; spans[ns].first = r; spans[ns].off = o->out_off; spans[ns].len = 0; ++ns;
        }
        spans[ns - 1].n_records++; spans[ns - 1].len += o->rec_len;
        /* the record, trim.rs:447-460 */
        uint8_t* w = text + o->out_off;
        const uint32_t read = o->read_idx;
        const uint8_t* hd = h->hdr + h->hdr_offsets[read];
        const uint64_t hl = h->hdr_offsets[read + 1] - h->hdr_offsets[read];
        *w++ = '@';
        memcpy(w, hd, h->id_len[read]); w += h->id_len[read];
        if (o->suffix) w += sprintf((char*)w, "_%u", (unsigned)o->suffix);
        if (cfg->write_full_header && hl > h->desc_start[read]) {
            *w++ = ' ';
            memcpy(w, hd + h->desc_start[read], hl - h->desc_start[read]); w += hl - h->desc_start[read];
        }
        *w++ = '\n';
        const uint64_t b0 = offsets[read];
        const uint32_t seq_len = (uint32_t)(offsets[read + 1] - b0);
        const uint32_t s0 = cfg->skip_trim ? 0 : o->start, s1 = cfg->skip_trim ? seq_len : o->end, L = s1 - s0;
        for (uint32_t k = 0; k < L; ++k) w[k] = o->flip ? comp_base(bases[b0 + s1 - 1 - k]) : bases[b0 + s0 + k];
        w += L;
        *w++ = '\n'; *w++ = '+'; *w++ = '\n';
        for (uint32_t k = 0; k < L; ++k) w[k] = o->flip ? quals[b0 + s1 - 1 - k] : quals[b0 + s0 + k];
        w += L;
        *w++ = '\n';
        if ((uint64_t)(w - (text + o->out_off)) != o->rec_len) { free(recs); return BB_E_INVALID; }
    }
    free(recs);
    return BB_OK;
}

/* ------------------------------------------------------------------------------------------ */
/* inspect step: get_group_structure inspect.rs:9-117 per row                   */
/* ------------------------------------------------------------------------------------------ */
static uint32_t bucket_position(uint32_t pos, uint32_t bs) { return ((pos ? pos - 1 : 0) / bs) * bs; }   /* inspect.rs:9-13 */
static uint32_t sat_sub(uint32_t a, uint32_t b) { return a > b ? a - b : 0; }
int bbo_inspect_rows(const bb_row* rows, const bb_row_verdict* verdicts, uint64_t n_rows, uint32_t bs, bb_inspect_elem* out) {
    if ((!rows && n_rows) || (!out && n_rows) || bs == 0) return BB_E_INVALID;
    for (uint64_t i = 0; i < n_rows; ++i) {
        const bb_row* a = &rows[i];
        const int first = i == 0 || rows[i - 1].read_idx != a->read_idx;
        bb_inspect_elem* e = &out[i];
        memset(e, 0, sizeof(*e));
        const uint32_t start = a->read_start_bar, end = a->read_end_bar, len = a->read_len;
        if (!first) {                                                       /* :41-58 */
            const uint32_t d_prev = sat_sub(start, rows[i - 1].read_end_bar), d_right = sat_sub(len, end);
            if (d_prev <= d_right) { e->tag = BB_REL_PREV_LEFT; e->lo = bucket_position(d_prev, bs); e->hi = e->lo + bs; }
            else { e->tag = BB_REL_RIGHT; e->lo = bucket_position(sat_sub(len, end), bs); e->hi = bucket_position(sat_sub(len, start), bs) + bs; }
        } else if (a->rel_dist_to_end > 0) {                               /* :59-63 */
            e->tag = BB_REL_LEFT; e->lo = bucket_position(start, bs); e->hi = e->lo + bs;
        } else {                                                            /* :64-70 */
            e->tag = BB_REL_RIGHT; e->lo = bucket_position(sat_sub(len, end), bs); e->hi = bucket_position(sat_sub(len, start), bs) + bs;
        }
        e->has_cut = verdicts && verdicts[i].n_cuts > 0;                    /* :72-85 */
        e->match_type = a->match_type; e->strand = a->strand; e->first = (uint32_t)first;
    }
    return BB_OK;
}

/* ------------------------------------------------------------------------------------------ */
/* FASTQ ingest: 4-line records + split_fastq_header (io.rs:6-17), scalar       */
/* ------------------------------------------------------------------------------------------ */
/* byte length of the char::is_whitespace character (Unicode White_Space, UTF-8) starting at p, 0 if none (io.rs:6-17) */
static uint32_t ws_len_c(const uint8_t* p, uint64_t n) {
    uint8_t c = p[0];
    if (c == ' ' || (c >= 9 && c <= 13)) return 1;
    if (c == 0xC2) return (n >= 2 && (p[1] == 0x85 || p[1] == 0xA0)) ? 2 : 0;
    if (n < 3) return 0;
    if (c == 0xE1) return (p[1] == 0x9A && p[2] == 0x80) ? 3 : 0;
    if (c == 0xE2) {
        if (p[1] == 0x80) return ((p[2] >= 0x80 && p[2] <= 0x8A) || p[2] == 0xA8 || p[2] == 0xA9 || p[2] == 0xAF) ? 3 : 0;
        return (p[1] == 0x81 && p[2] == 0x9F) ? 3 : 0;
    }
    if (c == 0xE3) return (p[1] == 0x80 && p[2] == 0x80) ? 3 : 0;
    return 0;
}
int bbo_fastq_parse(const uint8_t* text, uint64_t len, int final_block, bb_fastq_info* info, uint64_t* offsets, uint8_t* bases,
                    uint8_t* quals, uint8_t* hdr, uint64_t* hdr_offsets, uint32_t* id_len, uint32_t* desc_start) {
    if (!info || (!text && len)) return BB_E_INVALID;
    const int lpr = (final_block & BB_FASTQ_TWO_LINE) ? 2 : 4;    /* lines per record: the compact form has header + sequence only */
    final_block &= BB_FASTQ_FINAL;
    memset(info, 0, sizeof(*info));
    info->bad_record = -1;
    const int fill = offsets && bases && (quals || lpr == 2) && hdr && hdr_offsets && id_len && desc_start;
    uint64_t pos = 0, n = 0, nb = 0, nh = 0;
    if (fill) { offsets[0] = 0; hdr_offsets[0] = 0; }
    for (;;) {
        uint64_t ls[4], le[4], p = pos;                       /* the next four lines */
        int got = 0;
        while (got < lpr && p < len) {
            uint64_t e = p;
            while (e < len && text[e] != '\n') ++e;
            if (e == len && !final_block) break;               /* incomplete line: leave it to the next block */
            ls[got] = p; le[got] = e; ++got;
            p = e < len ? e + 1 : len;
        }
        if (got < lpr) {
            if (final_block) {
                int blank = 1;
                for (uint64_t q = pos; q < len; ++q) if (text[q] != '\n' && text[q] != '\r') blank = 0;
                if (!blank) { info->n_records = n; info->consumed = pos; info->bad_record = (int64_t)n; return BB_E_FASTQ; }
                pos = len;
            }
            break;
        }
        for (int i = 0; i < lpr; ++i) if (le[i] > ls[i] && text[le[i] - 1] == '\r') --le[i];
        if (final_block) {   /* blank lines after the last record: nothing from here to the end of the stream but line ends (in the two-line form
                                two of them would otherwise be taken for a record) */
            int blank = 1;
            for (uint64_t q = pos; q < len && blank; ++q) if (text[q] != '\n' && text[q] != '\r') blank = 0;
            if (blank) { pos = len; break; }
        }
        if (!(le[0] > ls[0] && text[ls[0]] == '@' && (lpr == 2 || (le[2] > ls[2] && text[ls[2]] == '+' && le[1] - ls[1] == le[3] - ls[3])))) {
            if (info->bad_record < 0) info->bad_record = (int64_t)n;
        }
        const uint64_t L = le[1] - ls[1], HL = le[0] > ls[0] ? le[0] - ls[0] - 1 : 0;
        if (fill && info->bad_record < 0) {
            memcpy(bases + nb, text + ls[1], L);
            if (lpr == 4) memcpy(quals + nb, text + ls[3], L);
            memcpy(hdr + nh, text + ls[0] + 1, HL);
            uint32_t idl = (uint32_t)HL, ds = (uint32_t)HL;
            for (uint32_t q = 0; q < HL; ++q) if (ws_len_c(text + ls[0] + 1 + q, HL - q)) { idl = q; break; }
            if (idl < HL) { ds = idl; for (uint32_t w; ds < HL && (w = ws_len_c(text + ls[0] + 1 + ds, HL - ds)) != 0;) ds += w; }
            id_len[n] = idl; desc_start[n] = ds;
            offsets[n + 1] = nb + L; hdr_offsets[n + 1] = nh + HL;
        }
        nb += L; nh += HL; ++n;
        pos = p;
    }
    info->n_records = n; info->consumed = pos; info->n_bases = nb; info->n_hdr = nh;
    return info->bad_record >= 0 ? BB_E_FASTQ : BB_OK;
}
