/*
 * barbell_amd_policy.h — the switchable assumptions ("hazards" H1..H8 of SURVEY.md §8c) about the two crates whose
 * sources are absent from the reference tree: sassy 0.2.1 (Cargo.toml:20; call sites searcher.rs:209-211,282-288,438)
 * and cigar-lodhi-rs 0.1.0 (Cargo.toml:36; call sites searcher.rs:209,238,367).
 *
 * Barbell's own code pins what it does with the crates' results; it does not pin which end of a cost plateau sassy
 * reports, how its traceback breaks ties, how the overhang cost is rounded, in which order reverse-complement matches
 * are returned, or which formula `Lodhi::compute` evaluates.  Each of those is ONE field here, honoured by the HIP
 * kernels (barbell_amd/csrc) and by the CPU checker (oracle/) alike, so that the first contact with the real crates
 * (tools/ref_diff.py --fit on a box with `barbell`, or tests/golden/ref_kat.jsonl from tools/ref_golden/kat.rs) closes
 * parity by choosing a policy, not by rewriting kernels.  The default is what rounds 1-2 implemented.
 *
 * Text form (BARBELL_AMD_POLICY, `barbell-amd --policy`, tools/ref_diff.py): comma separated key=value, any subset:
 *     lm=right|left|strict   rc=scan|fwd   trace=MISD (a permutation of M S I D)   ovh=floor|ceil|near[:f64]
 *     tie=first|last         lodhi=<p>:<lambda>:<eM><eS><eI><eD>      rcpath=fwd|mirror
 *     e.g. "lm=left,trace=MSID,lodhi=3:0.5:2211,rcpath=mirror"
 */
#ifndef BARBELL_AMD_POLICY_H
#define BARBELL_AMD_POLICY_H

#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#ifdef __cplusplus
extern "C" {
#endif

/* unit alignment ops, pa-types naming with Pos(i = pattern, j = text) */
#define BB_OP_MATCH 0  /* +(1,1), cost 0 */
#define BB_OP_SUB   1  /* +(1,1), cost 1 */
#define BB_OP_INS   2  /* +(0,1): a text character that is not in the pattern */
#define BB_OP_DEL   3  /* +(1,0): a pattern character that is not in the text */

/* [H1] which end positions with cost <= k `search` reports (cost sequence C[0..], one value per end position) */
#define BB_LM_PLATEAU_RIGHT 0  /* a position is reported when the cost stops falling there; a plateau at its right end */
#define BB_LM_PLATEAU_LEFT  1  /* the same minima, a plateau at its left end                                          */
#define BB_LM_STRICT        2  /* only strict local minima (C[i-1] > C[i] < C[i+1], ends of the sequence count as higher) */
/* [H2] order of the reverse-complement matches in the returned Vec (forward matches always come first:
 * cigar_parse.rs:163-176 pins that) */
#define BB_RC_SCAN_ORDER 0     /* as the rc scan finds them: ascending end position in the REVERSED text */
#define BB_RC_FWD_ORDER  1     /* ascending position in the forward text                                */
/* [H4] cost of o pattern characters hanging over a text end: round(alpha * o) */
#define BB_OVH_FLOOR 0
#define BB_OVH_CEIL  1
#define BB_OVH_NEAR  2         /* to nearest, ties to even */
#define BB_OVH_F64   4         /* flag: the product is formed in f64 instead of f32 */
/* [H5] pattern indices of the path cells (`Match::to_path()`, consumed by get_matching_region, cigar_parse.rs:71-82, with
 * bar_region = (prefix_len, prefix_len + mask_len - 1), barcodes.rs:192) of a Strand::Rc flank match */
#define BB_RCPATH_FWD    0     /* those of the forward flank: rc searched as complement(flank) in the reversed text           */
#define BB_RCPATH_MIRROR 1     /* mirrored, i -> m - 1 - i: rc searched as reverse_complement(flank) in the forward text, the
                                  region then selects rows m-1-bar_hi .. m-1-bar_lo of the flank (NB96: (8,31) for (14,37))   */
/* [H7] which of several equally cheap matches of one barcode pattern collect_candidates keeps (searcher.rs:294-300
 * keeps the first STRICTLY lowest of the Vec sassy returns; `last` models a Vec in descending position order) */
#define BB_TIE_FIRST 0
#define BB_TIE_LAST  1

typedef struct {
    uint8_t lm_rule;        /* [H1] BB_LM_*                                                              */
    uint8_t rc_order;       /* [H2] BB_RC_*                                                              */
    uint8_t trace_prio[4];  /* [H3] traceback: the ops in order of preference; default Match, Ins, Sub, Del */
    uint8_t ovh_round;      /* [H4] BB_OVH_* (| BB_OVH_F64)                                              */
    uint8_t bar_tie;        /* [H7] BB_TIE_*                                                             */
    uint8_t lodhi_p;        /* [H8] subsequence length, 1..4 (Lodhi::new(3, ..): 3)                      */
    uint8_t lodhi_exp[4];   /* [H8] decay exponent of one alignment column per op (M, S, I, D), 0..3: a triple of Match
                                    columns c1 < c2 < c3 weighs lambda^(sum of the exponents of the columns c1..c3).
                                    1,1,1,1 = span counted in alignment columns; 2,2,1,1 = span in the pattern plus span
                                    in the text (the two-string kernel of Lodhi et al. restricted to the alignment)  */
    uint8_t rc_path;        /* [H5] BB_RCPATH_*                                                          */
    uint8_t _pad[2];
    double  lodhi_lambda;   /* [H8] Lodhi::new(.., 0.5)                                                  */
} bb_policy;                /* 24 bytes */

static inline void bb_policy_default(bb_policy* p) {
    memset(p, 0, sizeof(*p));
    p->lm_rule = BB_LM_PLATEAU_RIGHT; p->rc_order = BB_RC_SCAN_ORDER;
    p->trace_prio[0] = BB_OP_MATCH; p->trace_prio[1] = BB_OP_INS; p->trace_prio[2] = BB_OP_SUB; p->trace_prio[3] = BB_OP_DEL;
    p->ovh_round = BB_OVH_FLOOR; p->bar_tie = BB_TIE_FIRST;
    p->lodhi_p = 3; p->lodhi_exp[0] = p->lodhi_exp[1] = p->lodhi_exp[2] = p->lodhi_exp[3] = 1; p->lodhi_lambda = 0.5;
}
/* 0 = usable */
static inline int bb_policy_validate(const bb_policy* p) {
    unsigned seen = 0;
    if (p->lm_rule > BB_LM_STRICT || p->rc_order > BB_RC_FWD_ORDER || p->bar_tie > BB_TIE_LAST || p->rc_path > BB_RCPATH_MIRROR) return -1;
    if ((p->ovh_round & 3) > BB_OVH_NEAR || (p->ovh_round & ~7u)) return -1;
    for (int i = 0; i < 4; ++i) { if (p->trace_prio[i] > 3) return -1; seen |= 1u << p->trace_prio[i]; }
    if (seen != 15u) return -1;
    if (p->lodhi_p < 1 || p->lodhi_p > 4) return -1;
    for (int i = 0; i < 4; ++i) if (p->lodhi_exp[i] > 3) return -1;
    if (!(p->lodhi_lambda > 0.0 && p->lodhi_lambda <= 1.0)) return -1;
    return 0;
}
static inline int bb_policy_trace_is_default(const bb_policy* p) {
    return p->trace_prio[0] == BB_OP_MATCH && p->trace_prio[1] == BB_OP_INS && p->trace_prio[2] == BB_OP_SUB && p->trace_prio[3] == BB_OP_DEL;
}
static inline int bb_policy_lodhi_is_default(const bb_policy* p) {
    return p->lodhi_p == 3 && p->lodhi_lambda == 0.5 && p->lodhi_exp[0] == 1 && p->lodhi_exp[1] == 1 && p->lodhi_exp[2] == 1 && p->lodhi_exp[3] == 1;
}
static inline int bb_policy_is_default(const bb_policy* p) {
    return p->lm_rule == 0 && p->rc_order == 0 && p->ovh_round == 0 && p->bar_tie == 0 && p->rc_path == 0 && bb_policy_trace_is_default(p) && bb_policy_lodhi_is_default(p);
}

/* text form -> struct (fields not named keep their value in *p); 0 ok, -1 malformed */
static inline int bb_policy_parse(const char* s, bb_policy* p) {
    char buf[256];
    if (!s) return 0;
    if (strlen(s) >= sizeof(buf)) return -1;
    strcpy(buf, s);
    /* split at ',' and ' ' without strtok's hidden state: contexts may be created from several threads at once */
    for (char* tok = buf; *tok;) {
        while (*tok == ',' || *tok == ' ') ++tok;
        if (!*tok) break;
        char* end = tok;
        while (*end && *end != ',' && *end != ' ') ++end;
        char* next = *end ? end + 1 : end;
        *end = 0;
        char* eq = strchr(tok, '=');
        if (!eq) return -1;
        *eq = 0;
        const char* v = eq + 1;
        if (!strcmp(tok, "lm")) {
            if (!strcmp(v, "right")) p->lm_rule = BB_LM_PLATEAU_RIGHT;
            else if (!strcmp(v, "left")) p->lm_rule = BB_LM_PLATEAU_LEFT;
            else if (!strcmp(v, "strict")) p->lm_rule = BB_LM_STRICT;
            else return -1;
        } else if (!strcmp(tok, "rc")) {
            if (!strcmp(v, "scan")) p->rc_order = BB_RC_SCAN_ORDER;
            else if (!strcmp(v, "fwd")) p->rc_order = BB_RC_FWD_ORDER;
            else return -1;
        } else if (!strcmp(tok, "trace")) {
            if (strlen(v) != 4) return -1;
            for (int i = 0; i < 4; ++i) {
                const char* at = strchr("MSID", v[i]);
                if (!at || !v[i]) return -1;
                p->trace_prio[i] = (uint8_t)(at - "MSID");
            }
        } else if (!strcmp(tok, "ovh")) {
            uint8_t r;
            if (!strncmp(v, "floor", 5)) { r = BB_OVH_FLOOR; v += 5; }
            else if (!strncmp(v, "ceil", 4)) { r = BB_OVH_CEIL; v += 4; }
            else if (!strncmp(v, "near", 4)) { r = BB_OVH_NEAR; v += 4; }
            else return -1;
            if (!strcmp(v, ":f64")) r |= BB_OVH_F64;
            else if (*v && strcmp(v, ":f32")) return -1;
            p->ovh_round = r;
        } else if (!strcmp(tok, "tie")) {
            if (!strcmp(v, "first")) p->bar_tie = BB_TIE_FIRST;
            else if (!strcmp(v, "last")) p->bar_tie = BB_TIE_LAST;
            else return -1;
        } else if (!strcmp(tok, "lodhi")) {
            int pp = 0; double lam = 0.0; char e[8] = "";
            if (sscanf(v, "%d:%lf:%4[0-9]", &pp, &lam, e) != 3 || strlen(e) != 4) return -1;
            if (pp < 1 || pp > 4) return -1;  /* before narrowing: 259 must not pass as 3 */
            p->lodhi_p = (uint8_t)pp; p->lodhi_lambda = lam;
            for (int i = 0; i < 4; ++i) p->lodhi_exp[i] = (uint8_t)(e[i] - '0');
        } else if (!strcmp(tok, "rcpath")) {
            if (!strcmp(v, "fwd")) p->rc_path = BB_RCPATH_FWD;
            else if (!strcmp(v, "mirror")) p->rc_path = BB_RCPATH_MIRROR;
            else return -1;
        } else return -1;
        tok = next;
    }
    return bb_policy_validate(p);
}
static inline void bb_policy_format(const bb_policy* p, char* out, size_t n) {
    static const char* const lm[] = {"right", "left", "strict"};
    static const char* const ov[] = {"floor", "ceil", "near", "?"};
    snprintf(out, n, "lm=%s,rc=%s,trace=%c%c%c%c,ovh=%s%s,tie=%s,lodhi=%d:%.17g:%d%d%d%d,rcpath=%s", lm[p->lm_rule % 3], p->rc_order ? "fwd" : "scan",
             "MSID"[p->trace_prio[0] & 3], "MSID"[p->trace_prio[1] & 3], "MSID"[p->trace_prio[2] & 3], "MSID"[p->trace_prio[3] & 3],
             ov[p->ovh_round & 3], (p->ovh_round & BB_OVH_F64) ? ":f64" : "", p->bar_tie ? "last" : "first", (int)p->lodhi_p, p->lodhi_lambda,
             (int)p->lodhi_exp[0], (int)p->lodhi_exp[1], (int)p->lodhi_exp[2], (int)p->lodhi_exp[3], p->rc_path ? "mirror" : "fwd");
}

#ifdef __cplusplus
}
#endif
#endif /* BARBELL_AMD_POLICY_H */
