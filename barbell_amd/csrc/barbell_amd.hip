// barbell_amd.hip — C-ABI implementation (include/barbell_amd.h): query preparation on the host
// (BarcodeGroup::new, barcodes.rs:105-197), device tables, and the per-batch kernel pipeline.
// No CPU fallback: every entry point that computes needs the GPU and fails loudly without it.
#include <hip/hip_runtime.h>

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <condition_variable>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/barbell_amd.h"
#include "../../include/barbell_amd_filter.h"
#include "../../include/barbell_amd_inspect.h"
#include "../../include/barbell_amd_synth.h"
#include "bb_common.h"
#include "bb_ctx.h"
#include "bb_launch.h"
#include "bb_lenstat.h"
#include "bb_k_rows.h"
#include "bb_k_misc.h"
#include "bb_synth.h"

namespace {

// barcodes.rs:394-441
uint8_t rc_char(uint8_t c) {
    static const char from[] = "ACTGactgRYSWKMBDHVNXryswkmbdhvnx";
    static const char to[] = "TGACtgacYRSWMKVHDBNXyrswmkvhdbnx";
    for (int i = 0; from[i]; ++i)
        if ((uint8_t)from[i] == c) return (uint8_t)to[i];
    return c;
}
// policy [H4]: cost of `len` pattern characters hanging over a text end = round(alpha * len); mode and width by policy
int overhang_cost(const bb_policy& P, float alpha, int len) {
    if (P.ovh_round & BB_OVH_F64) {
        const double v = (double)len * (double)alpha;
        switch (P.ovh_round & 3) { case BB_OVH_CEIL: return (int)ceil(v); case BB_OVH_NEAR: return (int)nearbyint(v); default: return (int)floor(v); }
    }
    const float v = (float)len * alpha;
    switch (P.ovh_round & 3) { case BB_OVH_CEIL: return (int)ceilf(v); case BB_OVH_NEAR: return (int)nearbyintf(v); default: return (int)floorf(v); }
}
// edit_model.rs:2-11
int edit_cut_off(int l) {
    double a = (double)l;
    double v = ceil(0.5100 * a - 1.7312 * sqrt(a));
    return v > 0.0 ? (int)v : 0;
}
// score of an all-Match CIGAR of length n (searcher.rs:229-239) under the policy's Lodhi recurrences (policy [H8];
// Lodhi::new(3, 0.5) by default; same op order as k_barcode)
double lodhi_all_match(const bb_policy& P, int n) {
    double d = 1.0;
    for (int e = 0; e < P.lodhi_exp[BB_OP_MATCH]; ++e) d = e == 0 ? P.lodhi_lambda : d * P.lodhi_lambda;
    const int lp = P.lodhi_p;
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, sc = 0.0;
    for (int c = 0; c < n; ++c) {
        sc = sc + d * (lp >= 4 ? a2 : lp == 3 ? a1 : lp == 2 ? a0 : 1.0);
        if (lp >= 4) a2 = d * (a2 + a1);
        if (lp >= 3) a1 = d * (a1 + a0);
        if (lp >= 2) a0 = d * (a0 + 1.0);
    }
    return sc;
}

int prep_group(const bb_group_desc& d, const bb_policy& pol, HostGroup& g) {
    if (!d.seqs || !d.seq_lens || d.n_seqs == 0) return BB_E_INVALID;
    if (d.n_seqs == 1) return BB_E_ONE_QUERY;
    const uint32_t L = d.seq_lens[0], n = d.n_seqs;
    for (uint32_t s = 1; s < n; ++s)
        if (d.seq_lens[s] != L) return BB_E_UNEQUAL_LEN;
    if (L == 0) return BB_E_NO_BARCODE;
    for (uint32_t s = 0; s < n; ++s) {
        if (!d.seqs[s]) return BB_E_INVALID;
        for (uint32_t i = 0; i < L; ++i)
            if (bb_iupac(d.seqs[s][i]) == 0xFF) return BB_E_NOT_IUPAC;
    }
    uint32_t pre = L, suf = L;  // barcodes.rs:337-385
    for (uint32_t s = 1; s < n; ++s) {
        uint32_t c = 0;
        while (c < L && d.seqs[0][c] == d.seqs[s][c]) ++c;
        pre = c < pre ? c : pre;
        c = 0;
        while (c < L && d.seqs[0][L - 1 - c] == d.seqs[s][L - 1 - c]) ++c;
        suf = c < suf ? c : suf;
    }
    if (pre + suf >= L) return BB_E_NO_BARCODE;
    if (pre == 0 && suf == 0) return BB_E_NO_FLANK;
    const uint32_t mask = L - pre - suf;
    g.seqs.resize(n);
    for (uint32_t s = 0; s < n; ++s) g.seqs[s].assign((const char*)d.seqs[s], L);
    g.flank = g.seqs[0].substr(0, pre) + std::string(mask, 'N') + g.seqs[0].substr(L - suf);
    bb_group_info& I = g.info;
    I.flank_len = L; I.prefix_len = pre; I.suffix_len = suf; I.mask_len = mask;
    I.bar_lo = pre; I.bar_hi = pre + mask - 1;
    I.pad_lo = pre >= BB_PADDING ? pre - BB_PADDING : 0;
    I.pad_hi = pre + mask + BB_PADDING;
    const uint32_t end = I.pad_hi < L ? I.pad_hi : L;
    I.pattern_len = end - I.pad_lo;
    g.pat[0].resize(n); g.pat[1].resize(n);
    for (uint32_t s = 0; s < n; ++s) {
        g.pat[0][s] = g.seqs[s].substr(I.pad_lo, I.pattern_len);
        std::string r(I.pattern_len, 'N');
        for (uint32_t i = 0; i < I.pattern_len; ++i) r[i] = (char)rc_char((uint8_t)g.pat[0][s][I.pattern_len - 1 - i]);
        g.pat[1][s] = r;
    }
    g.type = d.type;
    I.flank_k = d.flank_k >= 0 ? d.flank_k : edit_cut_off((int)(pre + suf));
    I.bar_k1 = (int32_t)((float)I.pattern_len * 0.4f);
    I.bar_k2 = (int32_t)I.pattern_len;
    I.perfect_score = lodhi_all_match(pol, (int)(I.pad_hi - I.pad_lo));
    return BB_OK;
}

struct Blob {
    std::vector<uint8_t> b;
    uint32_t alloc(size_t bytes) {
        size_t o = (b.size() + 15) & ~(size_t)15;
        b.resize(o + bytes, 0);
        return (uint32_t)o;
    }
};

void build_synth_tables(const std::vector<std::vector<std::string>>& seqs, bb_synth_params& P, std::vector<uint8_t>& table) {
    P.n_groups = (uint32_t)seqs.size();
    table.clear();
    for (size_t g = 0; g < seqs.size() && g < BB_MAX_GROUPS; ++g) {
        P.g[g].n_seqs = (uint32_t)seqs[g].size();
        P.g[g].seq_len = (uint32_t)seqs[g][0].size();
        P.g[g].off = (uint32_t)table.size();
        for (auto& s : seqs[g]) table.insert(table.end(), s.begin(), s.end());
    }
}

int ensure_ctl(bb_ctx* c, uint64_t flag_words);
// What a filter window (rows u .. u+R-1 of a flank whose forward Peq table is t) costs on L pseudo-random bases, run with the filter's own
// recurrence: every flagged column, plus per_run for the lead-in, piece granularity and margins of every flagged run (given up beyond `limit`).
static long filt_window_cost(const uint32_t* t, int S, int W, int k, int R, int u, int L, long per_run, long limit) {
    const uint32_t maskR = R >= 32 ? 0xFFFFFFFFu : (1u << R) - 1u;
    uint32_t eqs[4];
    for (int b = 0; b < 4; ++b) {
        const uint32_t* e = t + (size_t)"ACGT"[b] * S;
        uint64_t two = e[u >> 5];
        if ((u >> 5) + 1 < W) two |= (uint64_t)e[(u >> 5) + 1] << 32;
        eqs[b] = (uint32_t)(two >> (u & 31)) & maskR;
    }
    uint32_t pv = maskR, mv = 0, x = 0x9E3779B9u;
    int sc = R;
    long cost = 0;
    bool in = false;
    for (int i = 0; i < L && cost < limit; ++i) {
        x = x * 1664525u + 1013904223u;
        const uint32_t eq = eqs[x >> 30];
        const uint32_t xx = eq & pv, d0 = (((xx + pv) ^ pv) | eq | mv) & maskR;
        const uint32_t ph = (mv | ~(d0 | pv)) & maskR, mh = pv & d0;
        sc += (int)((ph >> (R - 1)) & 1u) - (int)((mh >> (R - 1)) & 1u);
        const uint32_t phs = ph << 1, mhs = mh << 1;
        pv = (mhs | ~(d0 | phs)) & maskR; mv = phs & d0 & maskR;
        const bool f = sc <= k;
        cost += f ? (in ? 1 : per_run) : 0;
        in = f;
    }
    return cost;
}
// bb_group_dev::filt_mode of a window at row u (what k_flank_verify scans whatever the flags say, and how the filter's forward block starts)
static uint32_t filt_mode_of(const bb_policy& pol, float alpha, int m, int k, int R, int u, bool wide) {
    int o_max = 0;  // most rows that can hang over a read end within the budget (edit_model: floor(alpha * o))
    for (int o = 1; o <= m; ++o)
        if (overhang_cost(pol, alpha, o) <= k) o_max = o;
    uint32_t mode = wide ? BB_FILT_WIDE : 0u;
    if (u == 0) mode |= BB_FILT_TRUE_INIT;
    if (!(u == 0 || u >= o_max)) mode |= BB_FILT_FWD_BEGIN_ALWAYS;
    if (u == 0 && o_max < R) mode |= BB_FILT_RC_BEGIN_HINT;
    else if (u < o_max) mode |= BB_FILT_RC_BEGIN_ALWAYS;
    if (o_max > m - u - R) mode |= BB_FILT_END_ALWAYS;
    if (getenv("BARBELL_AMD_FILTER_ENDS")) mode = (mode & (BB_FILT_TRUE_INIT | BB_FILT_WIDE)) | BB_FILT_FWD_BEGIN_ALWAYS | BB_FILT_RC_BEGIN_ALWAYS | BB_FILT_END_ALWAYS;  // test knob: both ends of every read
    return mode;
}

int upload_tables(bb_ctx* c) {
    Blob blob;
    c->gdev.resize(c->groups.size());
    memset(c->filt_twin, -1, sizeof(c->filt_twin)); memset(c->last_twin, -1, sizeof(c->last_twin));
    uint32_t count_off = 0;
    const float alpha = c->params.alpha;
    const bb_policy& pol = c->policy;
    for (size_t gi = 0; gi < c->groups.size(); ++gi) {
        const HostGroup& g = c->groups[gi];
        bb_group_dev& D = c->gdev[gi];
        memset(&D, 0, sizeof(D));
        const int m = (int)g.info.flank_len, W = (m + 31) / 32, S = bb_peq_stride_words(W);
        const int mb = (int)g.info.pattern_len, WB = (mb + 31) / 32, N = (int)g.seqs.size();
        D.m = m; D.W = W; D.flank_k = g.info.flank_k;
        D.bar_lo = (int)g.info.bar_lo; D.bar_hi = (int)g.info.bar_hi;
        D.m_bar = mb; D.WB = WB; D.n_seqs = N; D.k1 = g.info.bar_k1; D.k2 = g.info.bar_k2;
        D.rel_lo = (int)(g.info.bar_lo - g.info.pad_lo); D.rel_hi = (int)(g.info.bar_hi - g.info.pad_lo);
        D.type = g.type; D.score0 = overhang_cost(pol, alpha, m); D.count_off = (int)count_off; D.perfect = g.info.perfect_score;
        D.pol_lm = pol.lm_rule; D.pol_rc_fwd = pol.rc_order == BB_RC_FWD_ORDER; D.pol_tie_last = pol.bar_tie == BB_TIE_LAST;
        D.pol_prio = pol.trace_prio[0] | (pol.trace_prio[1] << 2) | (pol.trace_prio[2] << 4) | (pol.trace_prio[3] << 6);
        D.pol_lodhi_exp = pol.lodhi_exp[0] | (pol.lodhi_exp[1] << 8) | (pol.lodhi_exp[2] << 16) | (pol.lodhi_exp[3] << 24);
        D.pol_lodhi_p = pol.lodhi_p; D.pol_lambda = pol.lodhi_lambda; D.pol_rc_mirror = pol.rc_path == BB_RCPATH_MIRROR;
        count_off += (uint32_t)N + 1;
        for (int s = 0; s < 2; ++s) {
            D.off_peq_flank[s] = blob.alloc((size_t)256 * S * 4);
            uint32_t* t = reinterpret_cast<uint32_t*>(blob.b.data() + D.off_peq_flank[s]);
            for (int ch = 0; ch < 256; ++ch) {
                const uint8_t tc = bb_text_code((uint8_t)ch);
                for (int j = 0; j < m; ++j) {
                    uint8_t pc = bb_text_code((uint8_t)g.flank[j]);
                    if (s) pc = bb_comp_code(pc);  // rc strand: complement(pattern) vs reversed text
                    if (pc & tc) t[ch * S + (j >> 5)] |= 1u << (j & 31);
                }
            }
        }
        // Filtered scan (bb_kernels.h, k_flank_filter): worth it when the window's score is rarely <= k in unrelated text.
        // Every window position is tried on 32 K pseudo-random bases with the filter's own recurrence (runs of flagged
        // columns per text column); the quietest one is taken.
        D.filt_rows = 0; D.filt_off = 0; D.filt_mode = 0;
        {
            int o_max = 0;
            for (int o = 1; o <= m; ++o)
                if (overhang_cost(pol, alpha, o) <= D.flank_k) o_max = o;
            D.ovh_steps = std::min(m, o_max + 1);
        }
        {
            const int k = D.flank_k;
            if (c->scan_filter != 0 && m >= 2 && k < m) {
                const uint32_t* t = reinterpret_cast<const uint32_t*>(blob.b.data() + D.off_peq_flank[0]);
                const int L = 1 << 15;
                // cost of a window position in full-height columns per text column: every flagged column, plus the lead-in,
                // piece granularity and margins of every flagged run
                const long per_run = m + 3 * k + 24;
                auto best_window = [&](int R, int& best_u) -> double {
                    long best_cost = (long)L * per_run;
                    best_u = 0;
                    for (int u = 0; u + R <= m; ++u) {
                        const long cost = filt_window_cost(t, S, W, k, R, u, L, per_run, best_cost);
                        if (cost < best_cost) { best_cost = cost; best_u = u; }
                    }
                    return (double)best_cost / L;
                };
                // Everything in units of one full two-word scan of both strands (46 instructions per column): the packed filter
                // (15 rows per strand in one word) 0.43, the wide one (31 rows, a word per strand) 0.80, the full scan 0.5 per
                // word, verification = its columns at the full scan's price, with the waves' imbalance on top.
                const double full = W == 1 ? 0.6 : 0.5 * W;
                const int R15 = std::min(15, m), R31 = std::min(31, m);
                int u15 = 0, u31 = 0;
                const double c15 = best_window(R15, u15);
                const double c31 = R31 > R15 ? best_window(R31, u31) : 1e9;
                const double t15 = 0.43 + 1.3 * c15 * full, t31 = 0.80 + 1.3 * c31 * full;
                int pick = 0;  // 0 full scan, 1 packed, 2 wide
                if (W >= 2 && std::min(t15, t31) < 0.9 * full) pick = t15 <= t31 ? 1 : 2;
                if (getenv("BARBELL_AMD_VERBOSE"))
                    fprintf(stderr, "barbell_amd: group %zu flank %d nt, k %d: 15-row window at row %d: %.4f verification columns per column (cost %.2f), "
                            "31-row window at row %d: %.4f (cost %.2f), full scan %.2f -> %s\n", gi, m, k, u15, c15, t15, u31, c31 > 1e8 ? -1.0 : c31, t31, full,
                            pick == 0 ? "full scan" : pick == 1 ? "filtered scan (both strands in one word)" : "filtered scan (a word per strand)");
                if (c->scan_filter == 1 && pick == 0) pick = 1;
                if (getenv("BARBELL_AMD_FILTER_WIDE") && R31 > R15) pick = 2;  // test knob
                if (pick) {
                    const int R = pick == 1 ? R15 : R31, u = pick == 1 ? u15 : u31;
                    D.filt_rows = R; D.filt_off = u;
                    D.filt_mode = (int32_t)filt_mode_of(pol, alpha, m, k, R, u, pick == 2);
                }
            }
        }
        D.off_pv0 = blob.alloc((size_t)BB_MAX_W * 4);
        {
            uint32_t* t = reinterpret_cast<uint32_t*>(blob.b.data() + D.off_pv0);
            for (int j = 1; j <= m; ++j)
                if (overhang_cost(pol, alpha, j) - overhang_cost(pol, alpha, j - 1) == 1) t[(j - 1) >> 5] |= 1u << ((j - 1) & 31);
        }
        D.off_ovh = blob.alloc((size_t)(m + 1) * 4);
        {
            int32_t* t = reinterpret_cast<int32_t*>(blob.b.data() + D.off_ovh);
            for (int o = 0; o <= m; ++o) t[o] = overhang_cost(pol, alpha, o);
        }
        D.off_lut = blob.alloc(256);
        for (int ch = 0; ch < 256; ++ch) blob.b[D.off_lut + ch] = bb_text_code((uint8_t)ch);
        const uint32_t peq_bar = blob.alloc((size_t)2 * 16 * N * WB * 4);
        for (int s = 0; s < 2; ++s) {
            D.off_peq_bar[s] = peq_bar + (uint32_t)((size_t)s * 16 * N * WB * 4);
            uint32_t* t = reinterpret_cast<uint32_t*>(blob.b.data() + D.off_peq_bar[s]);
            for (int code = 0; code < 16; ++code)
                for (int p = 0; p < N; ++p)
                    for (int j = 0; j < mb; ++j)
                        if (bb_text_code((uint8_t)g.pat[s][p][j]) & code) t[((size_t)code * N + p) * WB + (j >> 5)] |= 1u << (j & 31);
        }
        // Row split (bb_common.h): per strand, the leading and trailing rows every barcode of the group has in common
        // (the pads; the rc patterns are the reverse complements, barcodes.rs:394-441, so their leading rows are the
        // forward patterns' trailing ones).  P leading rows + 32 rows per lane + T trailing rows = m_bar.
        for (int s = 0; s < 2; ++s) {
            D.split[s] = 0; D.pfx[s] = 0; D.tail[s] = 0;
            // a hit's N lanes must fit one block of the exact k_barcode_pfx and of the 64-column instantiation: 512 lanes (round 5: the exact
            // variant is compiled for 512-lane blocks and holds everything in registers; groups of 513..768 sequences take k_barcode)
            if (!(WB == 2 && mb <= 48 && N >= 13 && N <= 512) || getenv("BARBELL_AMD_NO_PFX") || c->generic_barcode) continue;
            int lcp = mb, lcs = mb;
            for (int p = 1; p < N; ++p) {
                int j = 0;
                while (j < lcp && bb_text_code((uint8_t)g.pat[s][p][j]) == bb_text_code((uint8_t)g.pat[s][0][j])) ++j;
                lcp = j;
                j = 0;
                while (j < lcs && bb_text_code((uint8_t)g.pat[s][p][mb - 1 - j]) == bb_text_code((uint8_t)g.pat[s][0][mb - 1 - j])) ++j;
                lcs = j;
            }
            const int need = mb - 32;                        // rows that do not fit the lane's word
            const int P = std::min(std::min(lcp, need), 16);  // as many as possible in front: they cost nothing per lane
            const int T = need - P;
            if (T < 0 || T > BB_MAX_TAIL || T > lcs) continue;
            if (getenv("BARBELL_AMD_NO_TAIL") && T > 0) continue;  // test knob: only tail-free splits
            D.split[s] = 1; D.pfx[s] = P; D.tail[s] = T;
            D.off_peq_pfx[s] = blob.alloc((size_t)16 * 4);
            D.off_peq_sub[s] = blob.alloc((size_t)16 * N * 4);
            D.off_tail_lut[s] = blob.alloc(16);
            uint32_t* tp = reinterpret_cast<uint32_t*>(blob.b.data() + D.off_peq_pfx[s]);
            uint32_t* ts = reinterpret_cast<uint32_t*>(blob.b.data() + D.off_peq_sub[s]);
            uint8_t* tl = blob.b.data() + D.off_tail_lut[s];
            for (int code = 0; code < 16; ++code) {
                for (int j = 0; j < P; ++j)
                    if (bb_text_code((uint8_t)g.pat[s][0][j]) & code) tp[code] |= 1u << j;
                for (int p = 0; p < N; ++p)
                    for (int j = P; j < P + 32; ++j)
                        if (bb_text_code((uint8_t)g.pat[s][p][j]) & code) ts[(size_t)code * N + p] |= 1u << (j - P);
                for (int t = 0; t < T; ++t)
                    if (bb_text_code((uint8_t)g.pat[s][0][P + 32 + t]) & code) tl[code] |= (uint8_t)(1u << t);
            }
        }
    }
    // TWIN filter windows (bb_ctx::filt_twin).  The right-hand pattern of a dual-end kit is the left-hand one's reverse complement, give or take a
    // few bases at the ends: R rows of group B's flank are then R rows of group A's read backwards and complemented, B's forward block in
    // k_flank_filter is A's rc block row for row (and the other way round), and ONE pass flags for both (k_flank_verify: swap_strands).  Each
    // group above chose its window alone; here pairs of groups with the same kind of pass get the quietest pair of MIRRORED windows — both
    // their own semi-global problem on both strands (no BB_FILT_TRUE_INIT, no rc-begin hint: neither at row 0) — if that beats two passes.
    for (size_t b = 1; b < c->groups.size() && !getenv("BARBELL_AMD_NO_TWIN_WINDOWS"); ++b) {
        bb_group_dev& B = c->gdev[b];
        for (size_t a = 0; a < b && B.filt_rows > 0 && c->filt_twin[b] < 0; ++a) {
            bb_group_dev& A = c->gdev[a];
            const int R = A.filt_rows;
            const bool wide = ((uint32_t)A.filt_mode & BB_FILT_WIDE) != 0;
            if (R <= 0 || R != B.filt_rows || wide != (((uint32_t)B.filt_mode & BB_FILT_WIDE) != 0) || c->filt_twin[a] >= 0) continue;
            if (std::min(A.flank_k, R) != std::min(B.flank_k, R)) continue;
            bool a_fixed = false;   // already another group's twin: its window stays
            for (size_t t = 0; t < c->groups.size(); ++t) a_fixed = a_fixed || c->filt_twin[t] == (int8_t)a;
            const std::string &fa = c->groups[a].flank, &fb = c->groups[b].flank;
            const int L = 1 << 15;
            struct Side { const uint32_t* t; int S, W, m, k; long per_run; double full; };
            auto side = [&](const bb_group_dev& D) {
                const int W = D.W;
                return Side{reinterpret_cast<const uint32_t*>(blob.b.data() + D.off_peq_flank[0]), bb_peq_stride_words(W), W, D.m, D.flank_k, (long)D.m + 3 * D.flank_k + 24,
                            W == 1 ? 0.6 : 0.5 * W};
            };
            const Side sa = side(A), sb = side(B);
            auto cost = [&](const Side& s, int u) { return (double)filt_window_cost(s.t, s.S, s.W, s.k, R, u, L, s.per_run, (long)L * s.per_run) / L; };
            auto ends = [&](const Side& s, uint32_t mode) {   // what the verification scans whatever the flags say: ~ (m + k) columns per read end and strand
                return 0.05 * s.full * (double)__builtin_popcount(mode & (BB_FILT_FWD_BEGIN_ALWAYS | BB_FILT_RC_BEGIN_ALWAYS | BB_FILT_END_ALWAYS));
            };
            const double pass = wide ? 0.80 : 0.43;
            const double t_sep = 2.0 * pass + 1.3 * (cost(sa, A.filt_off) * sa.full + cost(sb, B.filt_off) * sb.full) + ends(sa, (uint32_t)A.filt_mode) + ends(sb, (uint32_t)B.filt_mode);
            double best = t_sep;
            int best_ua = -1, best_ub = -1;
            bool best_swap = false;
            // (two relations: B's rows are A's read backwards and complemented — the two ends of a dual-end kit: strands swapped — or the very same rows:
            // groups of one kit that share most of their flank, e.g. the two of SQK-RBK114-96 --use-extended, which differ in their first 16 nt)
            for (int ua = a_fixed ? A.filt_off : 1; ua + R <= sa.m && (!a_fixed || ua == A.filt_off); ++ua)
                for (int ub = 1; ub + R <= sb.m; ++ub) {
                    bool mirror = true, same = true;
                    for (int i = 0; i < R && (mirror || same); ++i) {
                        const uint8_t cb = bb_text_code((uint8_t)fb[ub + i]);
                        mirror = mirror && cb == bb_comp_code(bb_text_code((uint8_t)fa[ua + R - 1 - i]));
                        same = same && cb == bb_text_code((uint8_t)fa[ua + i]);
                    }
                    if (!mirror && !same) continue;
                    const uint32_t ma = filt_mode_of(pol, alpha, sa.m, sa.k, R, ua, wide), mb = filt_mode_of(pol, alpha, sb.m, sb.k, R, ub, wide);
                    if ((ma | mb) & (BB_FILT_TRUE_INIT | BB_FILT_RC_BEGIN_HINT)) continue;
                    const double tt = pass + 1.3 * (cost(sa, ua) * sa.full + cost(sb, ub) * sb.full) + ends(sa, ma) + ends(sb, mb);
                    if (tt < best) { best = tt; best_ua = ua; best_ub = ub; best_swap = !same; }
                }
            if (best_ua < 0) continue;
            c->filt_twin_swap[b] = best_swap;
            if (getenv("BARBELL_AMD_VERBOSE"))
                fprintf(stderr, "barbell_amd: groups %zu and %zu: windows at rows %d and %d (their own choices: %d and %d) are %s: one filter pass "
                        "for both (cost %.2f against %.2f)\n", a, b, best_ua, best_ub, A.filt_off, B.filt_off, best_swap ? "each other's reverse complement" : "the same rows", best, t_sep);
            A.filt_off = best_ua; A.filt_mode = (int32_t)filt_mode_of(pol, alpha, sa.m, sa.k, R, best_ua, wide);
            B.filt_off = best_ub; B.filt_mode = (int32_t)filt_mode_of(pol, alpha, sb.m, sb.k, R, best_ub, wide);
            c->filt_twin[b] = (int8_t)a;
        }
    }
    c->counts_len = count_off;
    // k_emit keeps one block-local histogram of the whole context in LDS: 4 B per (group, barcode | flank-only) slot.  32 groups x 1024 sequences
    // are 131 KB — above the 64 KB a launch gets without asking, inside what the CU has
    if ((size_t)c->counts_len * 4 > BB_LDS_MAX) { c->last_error = "histogram of more than 36 864 (group, barcode) slots"; return BB_E_UNSUPPORTED; }
    if ((size_t)c->counts_len * 4 > 64 * 1024)
        HIPCHK(c, hipFuncSetAttribute((const void*)k_emit, hipFuncAttributeMaxDynamicSharedMemorySize, (int)((size_t)c->counts_len * 4)));
    HIPCHK(c, hipMalloc((void**)&c->d_tables, blob.b.size()));
    HIPCHK(c, hipMemcpy(c->d_tables, blob.b.data(), blob.b.size(), hipMemcpyHostToDevice));
    HIPCHK(c, hipMalloc((void**)&c->d_groups, sizeof(bb_group_dev) * c->gdev.size()));
    HIPCHK(c, hipMemcpy(c->d_groups, c->gdev.data(), sizeof(bb_group_dev) * c->gdev.size(), hipMemcpyHostToDevice));
    HIPCHK(c, hipMalloc((void**)&c->d_counts, sizeof(unsigned long long) * c->counts_len));
    HIPCHK(c, hipMemset(c->d_counts, 0, sizeof(unsigned long long) * c->counts_len));
    { int r = ensure_ctl(c, 0); if (r != BB_OK) return r; }
    HIPCHK(c, hipHostMalloc((void**)&c->h_ctl, sizeof(bb_ctl), hipHostMallocDefault));
    // synth tables
    std::vector<std::vector<std::string>> seqs;
    for (auto& g : c->groups) seqs.push_back(g.seqs);
    std::vector<uint8_t> table;
    build_synth_tables(seqs, c->synth, table);
    HIPCHK(c, hipMalloc((void**)&c->d_synth_table, table.size()));
    HIPCHK(c, hipMemcpy(c->d_synth_table, table.data(), table.size(), hipMemcpyHostToDevice));
    return BB_OK;
}

// the control block and, behind it, `flag_words` words for the filtered groups' flags: one allocation, one memset per batch (bb_ctl)
int ensure_ctl(bb_ctx* c, uint64_t flag_words) {
    if (c->d_ctl && flag_words <= c->cap_flags) return BB_OK;
    if (c->d_ctl) HIPCHK(c, hipFree(c->d_ctl));
    c->d_ctl = nullptr;
    const uint64_t cap = flag_words + flag_words / 4 + 64;
    uint8_t* p = nullptr;
    HIPCHK(c, hipMalloc((void**)&p, BB_CTL_BYTES + cap * sizeof(uint32_t)));
    c->d_ctl = reinterpret_cast<bb_ctl*>(p);
    c->d_flags = reinterpret_cast<uint32_t*>(p + BB_CTL_BYTES); c->cap_flags = cap;
    c->d_hitcount = c->d_ctl->hitcount; c->d_vqueue = c->d_ctl->vqueue; c->d_nflag = c->d_ctl->nflag;
    c->d_listcnt = c->d_ctl->listcnt; c->d_fbcnt = c->d_ctl->fbcnt;
    return BB_OK;
}
int ensure_reads(bb_ctx* c, uint32_t n) {
    const uint64_t M = (uint64_t)n * c->groups.size() * 2 + 1;
    if (M > c->cap_m) {
        uint64_t cap = 0;
        int r;
        if ((r = grow(c, c->d_cnt, cap, M))) return r;
        cap = 0;
        if ((r = grow(c, c->d_base, cap, M))) return r;
        c->cap_m = cap;
        uint64_t nb = (cap + 2047) / 2048 + 64, cs = 0;
        if ((r = grow(c, c->d_sums, cs, nb))) return r;
    }
    if (n > c->cap_reads) {
        uint64_t cap = 0;
        int r;
        if ((r = grow(c, c->d_nrows, cap, (uint64_t)n + 1))) return r;
        cap = 0;
        if ((r = grow(c, c->d_rowoff, cap, (uint64_t)n + 1))) return r;
        c->cap_reads = (uint32_t)(cap - 1);
    }
    return BB_OK;
}
int ensure_hits(bb_ctx* c, uint64_t need) {
    if (need <= c->cap_hits) return BB_OK;
    uint64_t cap = 0;
    int r;
    if ((r = grow(c, c->d_raw, cap, need))) return r;
    cap = 0;
    if ((r = grow(c, c->d_hits, cap, need))) return r;
    cap = 0;
    if ((r = grow(c, c->d_hitmeta, cap, need))) return r;
    cap = 0;
    {
        bool any_split = false;
        for (auto& d : c->gdev) any_split = any_split || d.split[0] || d.split[1];
        if (any_split) { if ((r = grow(c, c->d_pfx, cap, need))) return r; cap = 0; }
    }
    if ((r = grow(c, c->d_rows, cap, need))) return r;
    uint64_t lc = 0;
    if ((r = grow(c, c->d_lists, lc, cap * c->groups.size() * 4))) return r;
    lc = 0;
    if ((r = grow(c, c->d_fb_lists, lc, cap * c->groups.size() * 4))) return r;  // slot 4g + 2 wide + strand (k_hit_lists)
    c->cap_hits = (uint32_t)cap;
    return BB_OK;
}

void mark(bb_ctx* c, int i) {
    if (c->timing) (void)hipEventRecord(c->ev[i], c->stream);
}

}  // namespace

// What k_len_hist (bb_len.h) computes, from offsets the host holds: the staging buffer is 128-byte aligned, so a read's first line begins
// (offset mod 128) bytes before it.  Same record, same decisions downstream (bb_prepare_lengths).
static void bb_host_lenstat(const uint64_t* offsets, uint32_t n, uint32_t seg_lines, uint32_t split_above, bb_lenstat* st) {
    memset(st, 0, sizeof *st);
    st->off0 = offsets[0]; st->off1 = offsets[n];
    uint32_t lo = 0xFFFFFFFFu, hi = 0u;
    for (uint32_t i = 0; i < n; ++i) {
        const uint64_t off = offsets[i];
        const uint32_t len = (uint32_t)(offsets[i + 1] - off);
        const uint32_t nl = len ? (uint32_t)(((off & 127u) + len + 127u) >> 7) : 0u;
        lo = std::min(lo, nl); hi = std::max(hi, nl);
        if (nl <= split_above) { if (nl < BB_LEN_SEG_BINS) ++st->seg[nl]; }
        else {
            const uint32_t nseg = (nl + seg_lines - 1u) / seg_lines;
            st->seg[seg_lines] += nseg - 1u;
            ++st->seg[nl - (nseg - 1u) * seg_lines];
            ++st->n_cut_reads; st->n_cut_segs += nseg;
        }
    }
    st->min_nl = lo; st->max_nl = hi;
}

static inline double bb_now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// A binding's worker threads hold a context each (annotator.rs:88-101 keeps one Demuxer per thread) and every context owns two streams; HIP spreads a
// process's streams over FOUR hardware queues unless told otherwise, and streams that share a queue run one after the other.  Measured (round 6,
// ten threads, 8 192-read calls): 9.1 M reads/s on four queues, 13.1 M on sixteen.  The runtime reads the variable when it initialises, so
// it is set when this library is loaded, and only if the process has not chosen a value itself.
__attribute__((constructor)) static void bb_more_hw_queues() { setenv("GPU_MAX_HW_QUEUES", "16", 0); }

static thread_local std::string g_create_error;  // why the last bb_create on this thread failed (bb_last_error(NULL))

extern "C" {

const char* bb_strerror(int code) {
    switch (code) {
        case BB_OK: return "ok";
        case BB_E_INVALID: return "invalid argument";
        case BB_E_ONE_QUERY: return "a query group needs at least two sequences";
        case BB_E_UNEQUAL_LEN: return "all sequences per group must be equally long";
        case BB_E_NO_BARCODE: return "no barcode region found between shared prefix and suffix";
        case BB_E_NO_FLANK: return "no shared prefix or suffix found";
        case BB_E_NOT_IUPAC: return "sequence contains a character not supported by IUPAC";
        case BB_E_CAPACITY: return "row buffer too small";
        case BB_E_NO_DEVICE: return "no usable HIP device";
        case BB_E_HIP: return "HIP runtime error";
        case BB_E_UNSUPPORTED: return "query geometry outside the compiled kernel limits";
        case BB_E_NOMEM: return "out of memory";
        case BB_E_FASTQ: return "malformed FASTQ record";
        default: return "unknown error";
    }
}

int bb_create(const bb_group_desc* groups, uint32_t n_groups, const bb_params* params, bb_ctx** out) {
    // BARBELL_AMD_POLICY: the text form of include/barbell_amd_policy.h, for callers that cannot pass a struct (the CLI)
    bb_policy pol;
    bb_policy_default(&pol);
    if (const char* e = getenv("BARBELL_AMD_POLICY")) {
        if (bb_policy_parse(e, &pol) != 0) { g_create_error = std::string("BARBELL_AMD_POLICY: cannot parse \"") + e + "\""; return BB_E_INVALID; }
    }
    return bb_create_policy(groups, n_groups, params, &pol, out);
}

int bb_create_policy(const bb_group_desc* groups, uint32_t n_groups, const bb_params* params, const bb_policy* policy, bb_ctx** out) {
    g_create_error.clear();
    if (!groups || !params || !out || n_groups == 0) return BB_E_INVALID;
    if (n_groups > BB_MAX_GROUPS) { g_create_error = "more than 32 query groups"; return BB_E_UNSUPPORTED; }
    if (!(params->alpha >= 0.0f)) return BB_E_INVALID;
    if (policy && bb_policy_validate(policy) != 0) { g_create_error = "policy: field out of range"; return BB_E_INVALID; }
    bb_ctx* c = new bb_ctx();
    c->params = *params;
    if (policy) c->policy = *policy; else bb_policy_default(&c->policy);
    // The register-resident barcode kernels hard-wire Lodhi(3, 1/2) (exact power-of-two scaling: searcher.rs:209 pins both); their
    // traceback preference is a compile-time class (18 of them, one set of fast kernels each: bb_prio.h, bb_tu_class.hip), their
    // local-minimum rule, tie rule and decay exponents are run-time.  Another p or lambda runs the any-policy kernel k_barcode.
    // The fast path's score bound advances the time by eM on a Match column and by min(eS, eI) on any other text column (its table is
    // built from the policy's exponents): an upper bound under every setting of them.
    c->generic_barcode = c->policy.lodhi_p != 3 || c->policy.lodhi_lambda != 0.5;
    c->prio_class = bb_prio_class(BB_PRIO_PACK(c->policy.trace_prio[0], c->policy.trace_prio[1], c->policy.trace_prio[2], c->policy.trace_prio[3]));
    if (c->prio_class >= 0 && !bb_class_unit_of(c->prio_class).lane) {
        // the build holds the fast kernels of the classes the reference's own vectors leave open (Makefile CLASSES, tests/golden/policy_feasible.json);
        // any other order is still computed exactly, by the kernels that read the order at run time
        const uint8_t* t = c->policy.trace_prio;
        char note[256];
        snprintf(note, sizeof note, "note: the fast barcode kernels of traceback order trace=%c%c%c%c are not in this build (make CLASSES=all); its hits run "
                                    "the kernels that take the order at run time (k_barcode_pfx<.., BB_PRIO_RT>): same rows, slower",
                 "MSID"[t[0] & 3], "MSID"[t[1] & 3], "MSID"[t[2] & 3], "MSID"[t[3] & 3]);
        c->last_error = note;
        static bool said = false;
        if (!said) { said = true; fprintf(stderr, "barbell_amd: %s\n", note); }
    }
    c->force_generic = getenv("BARBELL_AMD_GENERIC") && atoi(getenv("BARBELL_AMD_GENERIC")) != 0;
    if (getenv("BARBELL_AMD_SCAN_FILTER")) c->scan_filter = atoi(getenv("BARBELL_AMD_SCAN_FILTER")) != 0 ? 1 : 0;
    if (getenv("BARBELL_AMD_NO_FAST") && atoi(getenv("BARBELL_AMD_NO_FAST")) != 0) c->fast_path = false;
    if (getenv("BARBELL_AMD_FAST_MARGIN")) c->fast_margin = atof(getenv("BARBELL_AMD_FAST_MARGIN"));
    if (getenv("BARBELL_AMD_ADAPT_FRAC")) c->adapt_frac = atof(getenv("BARBELL_AMD_ADAPT_FRAC"));
    if (getenv("BARBELL_AMD_SEG_LINES")) {   // lines (128 bytes) per segment of a cut read: a multiple of 4 up to 64; reads above twice that are cut; 0: lanes take whole reads in file order
        const int v = atoi(getenv("BARBELL_AMD_SEG_LINES"));
        c->seg_lines = v <= 0 ? 0u : (uint32_t)std::min(64, std::max(4, v & ~3));
        c->split_above = 2u * c->seg_lines;
        c->seg_from_env = true;
    }
    if (const char* e = getenv("BARBELL_AMD_DEFER_MAX")) c->defer_max = (uint32_t)strtoul(e, nullptr, 10);
    if (const char* e = getenv("BARBELL_AMD_SMALL_PFX_MAX")) c->small_pfx_max = (uint32_t)strtoul(e, nullptr, 10);
    memset(c->filt_twin, -1, sizeof(c->filt_twin)); memset(c->last_twin, -1, sizeof(c->last_twin));
    if (const char* e = getenv("BARBELL_AMD_FILTER_TWINS")) c->use_twins = atoi(e) != 0;
    if (const char* e = getenv("BARBELL_AMD_SMALL_SEG_MAX")) c->small_seg_max = (uint32_t)strtoul(e, nullptr, 10);
    if (const char* e = getenv("BARBELL_AMD_HOST_LEN_MAX")) c->host_len_max = (uint32_t)strtoul(e, nullptr, 10);
    c->phases = getenv("BARBELL_AMD_PHASES") && atoi(getenv("BARBELL_AMD_PHASES")) != 0;
    if (const char* e = getenv("BARBELL_AMD_LANE_FB_FRAC")) c->lane_fb_frac = atof(e);
    if (const char* e = getenv("BARBELL_AMD_LANE_PFX_GAIN")) c->lane_pfx_gain = atof(e);
    if (const char* e = getenv("BARBELL_AMD_LANE")) c->lane_kernel = std::max(0, std::min(2, atoi(e)));
    if (const char* e = getenv("BARBELL_AMD_LANE_NM")) c->lane_nm = atoi(e) != 0;
    if (getenv("BARBELL_AMD_PFX_THREADS")) { int t = atoi(getenv("BARBELL_AMD_PFX_THREADS")); if (t >= 64 && t <= 768) c->pfx_threads = (uint32_t)t; }
    if (getenv("BARBELL_AMD_REG_THREADS")) { int t = atoi(getenv("BARBELL_AMD_REG_THREADS")); if (t >= 64 && t <= 512) c->reg_threads = (uint32_t)t; }
    c->groups.resize(n_groups);
    for (uint32_t i = 0; i < n_groups; ++i) {
        int r = prep_group(groups[i], c->policy, c->groups[i]);
        if (r != BB_OK) { delete c; return r; }
        const bb_group_info& I = c->groups[i].info;
        // geometry the kernels were not built for (the reference has no such limits, barcodes.rs:105-197): the table in
        // include/barbell_amd.h; the detail is left for bb_last_error(NULL)
        char why[200] = "";
        if (I.flank_len > 32 * BB_MAX_W) snprintf(why, sizeof why, "group %u: flank of %u nt (prefix + barcode mask + suffix) exceeds %d", i, I.flank_len, 32 * BB_MAX_W);
        else if (I.pattern_len > 32 * BB_MAX_WB) snprintf(why, sizeof why, "group %u: padded barcode pattern of %u nt exceeds %d", i, I.pattern_len, 32 * BB_MAX_WB);
        else if (I.flank_k > BB_MAX_FLANK_K) snprintf(why, sizeof why, "group %u: flank error budget %d exceeds %d", i, I.flank_k, BB_MAX_FLANK_K);
        else if (I.mask_len + (uint32_t)I.flank_k + 2 * BB_PADDING > BB_MAX_WIN)
            snprintf(why, sizeof why, "group %u: barcode window (barcode %u + flank errors %d + 20) exceeds %d columns", i, I.mask_len, I.flank_k, BB_MAX_WIN);
        else if (groups[i].n_seqs > 1024) snprintf(why, sizeof why, "group %u: %u sequences exceed 1024", i, groups[i].n_seqs);
        if (why[0]) {
            g_create_error = why;
            delete c;
            return BB_E_UNSUPPORTED;
        }
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || params->device < 0 || params->device >= ndev) { delete c; return BB_E_NO_DEVICE; }
    c->device = params->device;
    if (hipSetDevice(c->device) != hipSuccess) { delete c; return BB_E_NO_DEVICE; }
    { hipDeviceProp_t prop; if (hipGetDeviceProperties(&prop, c->device) == hipSuccess && prop.multiProcessorCount > 0) c->n_cus = prop.multiProcessorCount; }
    if (getenv("BARBELL_AMD_REG_BLOCKS")) { int t = atoi(getenv("BARBELL_AMD_REG_BLOCKS")); if (t >= 1 && t <= 16) c->reg_blocks_mult = (uint32_t)t; }
    if (hipStreamCreate(&c->stream) != hipSuccess) { delete c; return BB_E_NO_DEVICE; }
    if (!getenv("BARBELL_AMD_NO_SIDE_STREAM") &&
        (hipStreamCreate(&c->side) != hipSuccess || hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming) != hipSuccess ||
         hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming) != hipSuccess)) { bb_destroy(c); return BB_E_NO_DEVICE; }
    for (int i = 0; i <= K_COUNT; ++i)
        if (hipEventCreate(&c->ev[i]) != hipSuccess) { delete c; return BB_E_NO_DEVICE; }
    int r = upload_tables(c);
    if (r != BB_OK) { fprintf(stderr, "barbell_amd: %s\n", c->last_error.c_str()); bb_destroy(c); return r; }
    *out = c;
    return BB_OK;
}

void bb_destroy(bb_ctx* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    void* ptrs[] = {c->d_groups, c->d_tables, c->d_counts, c->d_cnt, c->d_base, c->d_sums, c->d_nrows, c->d_rowoff, c->d_ctl,
                    c->d_lists, c->d_fb_lists, c->d_lenstat, c->d_lencur, c->d_vtab, c->d_vcut, c->d_cutread, c->d_cutlist, c->d_vcnt, c->d_raw, c->d_hits, c->d_hitmeta, c->d_pfx, c->d_rows, c->d_in_bases, c->d_in_offsets, c->d_out_rows, c->d_in_packed, c->d_in_poffs,
                    c->d_synth_table, c->d_fpats, c->d_felems, c->d_flabel_ok, c->d_flabel_ids, c->d_frows, c->d_fout, c->d_iout};
    for (void* p : ptrs)
        if (p) (void)hipFree(p);
    if (c->h_lencur) (void)hipHostFree(c->h_lencur);
    if (c->h_ctl) (void)hipHostFree(c->h_ctl);
    if (c->h_offs) (void)hipHostFree(c->h_offs);
    delete c->host_len;
    bb_trim_state_free(c->trim);
    bb_fastq_state_free(c->fastq);
    bb_format_state_free(c->format);
    for (int i = 0; i <= K_COUNT; ++i)
        if (c->ev[i]) (void)hipEventDestroy(c->ev[i]);
    for (auto& e : c->lev) { if (e.a) (void)hipEventDestroy(e.a); if (e.b) (void)hipEventDestroy(e.b); }
    if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
    if (c->ev_join) (void)hipEventDestroy(c->ev_join);
    if (c->side) (void)hipStreamDestroy(c->side);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}


int bb_get_policy(const bb_ctx* c, bb_policy* out) {
    if (!c || !out) return BB_E_INVALID;
    *out = c->policy;
    return BB_OK;
}
int bb_n_groups(const bb_ctx* c) { return c ? (int)c->groups.size() : BB_E_INVALID; }
int bb_group_get_info(const bb_ctx* c, uint32_t g, bb_group_info* info) {
    if (!c || g >= c->groups.size() || !info) return BB_E_INVALID;
    *info = c->groups[g].info;
    return BB_OK;
}
int bb_group_get_flank(const bb_ctx* c, uint32_t g, uint8_t* out) {
    if (!c || g >= c->groups.size() || !out) return BB_E_INVALID;
    memcpy(out, c->groups[g].flank.data(), c->groups[g].flank.size());
    return BB_OK;
}
int bb_group_get_pattern(const bb_ctx* c, uint32_t g, uint32_t idx, int rc, uint8_t* out) {
    if (!c || g >= c->groups.size() || !out || idx >= c->groups[g].seqs.size()) return BB_E_INVALID;
    const std::string& s = c->groups[g].pat[rc ? 1 : 0][idx];
    memcpy(out, s.data(), s.size());
    return BB_OK;
}

int bb_annotate_batch_dev(bb_ctx* c, const uint8_t* d_bases, const uint64_t* d_offsets, uint32_t n, bb_row* d_rows,
                          uint64_t rows_cap, uint64_t* n_rows) {
    if (!c || !d_offsets || !n_rows || (!d_bases && n)) return BB_E_INVALID;
    *n_rows = 0;
    bb_row* const spec_dst = c->spec_dst;   // (set by the host-pointer form for this call only)
    const uint64_t spec_cap = c->spec_cap;
    c->spec_dst = nullptr; c->spec_cap = 0; c->spec_done = 0;
    if (n == 0) return BB_OK;
    HIPCHK(c, hipSetDevice(c->device));
    bb_call_scope scope(c);
    const uint32_t G = (uint32_t)c->groups.size();
    int r;
    if ((r = ensure_reads(c, n))) return r;
    if ((r = ensure_hits(c, (uint64_t)n * 3 + 1024))) return r;
    const uint64_t M = (uint64_t)n * G * 2 + 1;
    // DEFERRED (bb_ctx::defer_max): a small batch's kernels take tens of microseconds each and its host thread's waits and launches the rest; nothing
    // between here and the rows waits for the device
    const bool deferred = n <= c->defer_max;
    c->batch_no_lane = n <= c->small_pfx_max;
    uint32_t n_hits = 0;
    uint64_t flag_words = 0, batch_bytes = 0, flag_total = 0;  // per strand; of all filtered groups
    double pt[5] = {c->phases ? bb_now() : 0.0, 0, 0, 0, 0};
    {
        // the batch's byte span and read lengths (bb_len.h: segments / reads sorted by length where they differ): one round trip, or none where the
        // host-pointer form took them from the offsets it holds
        uint64_t ends[2] = {0, 0};
        if ((r = bb_prepare_lengths(c, d_bases, d_offsets, n, &ends[0], &ends[1]))) return r;
        batch_bytes = ends[1] - ends[0];
        if (c->phases) pt[1] = bb_now();
        uint64_t n_filt = 0;   // one region of flag words per filtered group (their filter passes run as one launch)
        for (uint32_t g = 0; g < G; ++g) n_filt += c->gdev[g].filt_rows > 0 ? 1 : 0;
        if (n_filt) {  // the flag words of a read sit at (offset >> 9) + 3 * read: the batch's byte span sizes the array
            flag_words = ((ends[1] - ends[0]) >> 9) + 3ull * n + 3;
            flag_total = 2 * flag_words * n_filt;
            if ((r = ensure_ctl(c, flag_total))) return r;
        }
    }
    bool any_split = false;
    for (uint32_t g = 0; g < G; ++g) any_split = any_split || c->gdev[g].split[0] || c->gdev[g].split[1];
    const uint32_t* const hits_dev = deferred ? c->d_hitcount : nullptr;   // BB_HITS_ON_DEVICE
    uint32_t total = 0;
    for (int attempt = 0;; ++attempt) {
        mark(c, K_SCAN);
        // every counter of the batch and the filter's flag words: one memset (bb_ctl)
        HIPCHK(c, hipMemsetAsync(c->d_ctl, 0, BB_CTL_BYTES + flag_total * sizeof(uint32_t), c->stream));
        if ((r = bb_launch_scans(c, d_bases, d_offsets, n, flag_words, batch_bytes, deferred))) return r;   // every group's flank scan (bb_tu_scan.hip)
        HIPCHK(c, hipGetLastError());
        mark(c, K_PREFIX);
        if (!deferred) {
            HIPCHK(c, hipMemcpyAsync(&n_hits, c->d_hitcount, 4, hipMemcpyDeviceToHost, c->stream));
            BB_SYNC(c, c->stream);
            if (n_hits > c->cap_hits) {
                if (attempt > 2) { c->last_error = "flank hit buffer overflow"; return BB_E_HIP; }
                if ((r = ensure_hits(c, (uint64_t)n_hits + 1024))) return r;
                continue;
            }
        } else n_hits = c->cap_hits;   // an upper bound: the kernels read the count (hits_dev)
        if (c->phases) pt[2] = bb_now();
        if ((r = bb_scan_u32(c, c->d_cnt, c->d_base, M))) return r;
        mark(c, K_TRACE);
        if (n_hits) {
            uint32_t done = 0;
            for (uint32_t g = 0; g < G; ++g) {
                if ((done >> g) & 1u) continue;
                const int W = c->gdev[g].W, mode = bb_trace_mode(c, g);
                uint32_t gmask = 0;
                for (uint32_t g2 = g; g2 < G; ++g2)
                    if (c->gdev[g2].W == W && bb_trace_mode(c, g2) == mode) gmask |= 1u << g2;
                done |= gmask;
                bb_launch_trace(c, d_bases, d_offsets, n_hits, gmask, mode, W, hits_dev);
            }
            HIPCHK(c, hipGetLastError());
        }
        mark(c, K_LISTS);
        bool any_split_prefix = any_split;  // k_bar_prefix over every hit
        c->use_lists = true;  // one list per (group, strand)
        bool all_lane = any_split;
        for (uint32_t g = 0; g < G; ++g)
            for (uint32_t sd = 0; sd < 2; ++sd)
                if (c->gdev[g].split[sd]) {
                    const uint32_t win_max = c->groups[g].info.mask_len + (uint32_t)c->groups[g].info.flank_k + 2 * BB_PADDING - 1;
                    if (!(c->gdev[g].WB == 2 && !c->force_generic && !c->generic_barcode && bb_takes_lane(c, g, sd, false) && win_max <= 63 &&
                          (win_max <= 48 || bb_takes_lane(c, g, sd, true)))) all_lane = false;
                }
        c->lazy_prefix = all_lane && !getenv("BARBELL_AMD_FULL_PREFIX");
        if (c->lazy_prefix) any_split_prefix = false;
        // the second stream pays where launches are long enough to have a tail worth filling; in a small batch every fork and join is two more calls of a
        // host thread that has nothing else to spend
        hipStream_t const side = c->batch_no_lane ? nullptr : c->side;
        const bool prefix_aside = n_hits && any_split_prefix && side != nullptr;  // k_bar_prefix needs the hits, not their lists: alongside k_hit_lists
        if (prefix_aside) {
            HIPCHK(c, hipEventRecord(c->ev_fork, c->stream)); HIPCHK(c, hipStreamWaitEvent(c->side, c->ev_fork, 0));
            bb_launch_bar_prefix(c, n_hits, c->side, hits_dev);
            HIPCHK(c, hipEventRecord(c->ev_join, c->side));
        }
        if (n_hits)
            hipLaunchKernelGGL(k_hit_lists, dim3((n_hits + 255) / 256), dim3(256), 0, c->stream, (const uint32_t*)c->d_hitmeta, n_hits,
                               c->d_rows, c->d_lists, c->cap_hits, c->d_listcnt, G, (const bb_group_dev*)c->d_groups, hits_dev);
        mark(c, K_BARCODE);
        c->n_lev = 0;
        if (prefix_aside) HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_join, 0));
        if (n_hits) {
            if (any_split_prefix && !prefix_aside)  // shared rows of the padded barcodes, once per hit
                bb_launch_bar_prefix(c, n_hits, c->stream, hits_dev);
            for (int pass = 0; pass < 2; ++pass) {
                if (pass == 1) {
                    if (!(any_split && c->fast_path)) break;
                    // the exact score of every hit's best-bounded barcode; rows of the hits the bounds decide (k_barcode_lane has done that itself)
                    if (c->pfx_fast_launches)
                        hipLaunchKernelGGL(k_rows, dim3((n_hits + 255) / 256), dim3(256), 0, c->stream, (const bb_group_dev*)c->d_groups, (const bb_hit*)c->d_hits,
                                           n_hits, c->d_rows, c->params.min_score, c->params.min_score_diff, c->fast_margin, c->d_fb_lists, c->cap_hits, c->d_fbcnt, hits_dev);
                } else c->pfx_fast_launches = 0;
                const bool fork = side != nullptr;  // pass 1: the two strands' exact launches are small (the undecided hits) and overlap entirely
                if (fork) { HIPCHK(c, hipEventRecord(c->ev_fork, c->stream)); HIPCHK(c, hipStreamWaitEvent(c->side, c->ev_fork, 0)); }
                c->use_side = fork;
                for (uint32_t g = 0; g < G; ++g) {
                    bb_launch_barcode(c, d_bases, d_offsets, n_hits, g, pass);
                }
                c->use_side = false;
                if (fork) { HIPCHK(c, hipEventRecord(c->ev_join, c->side)); HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_join, 0)); }
            }
            HIPCHK(c, hipGetLastError());
        }
        mark(c, K_COLLAPSE);
        hipLaunchKernelGGL(k_collapse, dim3((n + 255) / 256), dim3(256), 0, c->stream, c->d_rows, (const uint32_t*)c->d_base, n, G,
                           c->d_nrows, hits_dev, c->cap_hits, (const uint32_t*)c->d_hitmeta);
        if ((r = bb_scan_u32(c, c->d_nrows, c->d_rowoff, (uint64_t)n + 1, &c->d_ctl->total_rows))) return r;
        if (deferred) {
            // rows out and counted on the device's own say-so (k_emit checks what the host is about to read), the batch's numbers into the page-locked
            // twin, the first rows to a host caller's buffer: then the batch's one wait
            mark(c, K_EMIT);
            if (d_rows)
                hipLaunchKernelGGL(k_emit, dim3((n + 255) / 256), dim3(256), (size_t)c->counts_len * 4, c->stream, (const bb_rowtmp*)c->d_rows,
                                   (const uint32_t*)c->d_base, (const uint32_t*)c->d_rowoff, n, G, (const bb_group_dev*)c->d_groups, d_rows,
                                   c->d_counts, c->counts_len, hits_dev, c->cap_hits, rows_cap);
            mark(c, K_COUNT);
            HIPCHK(c, hipGetLastError());
        }
        HIPCHK(c, hipMemcpyAsync(c->h_ctl, c->d_ctl, sizeof(bb_ctl), hipMemcpyDeviceToHost, c->stream));
        const uint64_t spec = deferred && spec_dst && d_rows ? std::min<uint64_t>(std::min(spec_cap, rows_cap), 2ull * n + 64) : 0;
        if (spec) HIPCHK(c, hipMemcpyAsync(spec_dst, d_rows, spec * sizeof(bb_row), hipMemcpyDeviceToHost, c->stream));
        BB_SYNC(c, c->stream);
        if (deferred && c->h_ctl->hitcount[0] > c->cap_hits) {   // the kernels after the scans left at once; again, with room
            if (attempt > 2) { c->last_error = "flank hit buffer overflow"; return BB_E_HIP; }
            if ((r = ensure_hits(c, (uint64_t)c->h_ctl->hitcount[0] + 1024))) return r;
            continue;
        }
        total = c->h_ctl->total_rows;
        c->spec_done = spec;
        break;
    }
    *n_rows = total;
    if (c->phases) pt[3] = bb_now();
    if (deferred) bb_note_flag_counts(c, c->h_ctl->nflag);
    const bool fb_stats = c->fast_path && c->lane_kernel == 1;
    for (uint32_t g = 0; g < G; ++g)
        for (uint32_t sd = 0; sd < 2; ++sd) {
            const bool probing = c->lane_off[g][sd] != 0;   // this batch ran k_barcode_pfx because the lane kernel had left too much undecided
            if (c->lane_off[g][sd]) --c->lane_off[g][sd];
            if (c->lane_noback[g][sd]) --c->lane_noback[g][sd];
            if (fb_stats) {
                const uint32_t *h_listed = c->h_ctl->listcnt, *h_undecided = c->h_ctl->fbcnt;
                const uint64_t listed = (uint64_t)h_listed[4 * g + sd] + h_listed[4 * g + 2 + sd], und = (uint64_t)h_undecided[4 * g + sd] + h_undecided[4 * g + 2 + sd];
                c->last_listed[g][sd] = listed; c->last_undecided[g][sd] = und;
                if (c->lane_used[g][sd] && listed >= 1024 && (double)und > c->lane_fb_frac * (double)listed && !c->lane_noback[g][sd]) {
                    c->lane_off[g][sd] = 32;
                    c->lane_und_frac[g][sd] = (float)((double)und / (double)listed);
                } else if (probing && !c->lane_used[g][sd] && listed >= 1024 &&
                           (double)c->lane_und_frac[g][sd] - (double)und / (double)listed < c->lane_pfx_gain) {
                    c->lane_off[g][sd] = 0;      // k_barcode_pfx decided hardly more (bb_ctx::lane_pfx_gain): back to the lane kernel ...
                    c->lane_noback[g][sd] = 64;  // ... for the next 64 batches whatever it leaves undecided
                }
            }
            c->lane_used[g][sd] = 0;
        }
    if (total > rows_cap || (total && !d_rows)) return BB_E_CAPACITY;
    if (!deferred) {
        mark(c, K_EMIT);
        if (total)
            hipLaunchKernelGGL(k_emit, dim3((n + 255) / 256), dim3(256), (size_t)c->counts_len * 4, c->stream, (const bb_rowtmp*)c->d_rows,
                               (const uint32_t*)c->d_base, (const uint32_t*)c->d_rowoff, n, G, (const bb_group_dev*)c->d_groups, d_rows,
                               c->d_counts, c->counts_len, (const uint32_t*)nullptr, 0u, (uint64_t)0);
        mark(c, K_COUNT);
        HIPCHK(c, hipGetLastError());
        BB_SYNC(c, c->stream);
    }
    if (c->phases) { pt[4] = bb_now(); for (int i = 0; i < 4; ++i) c->ph[3 + i] += pt[i + 1] - pt[i]; }
    if (c->timing) {
        for (int i = 0; i < K_COUNT; ++i) (void)hipEventElapsedTime(&c->ms[i], c->ev[i], c->ev[i + 1]);
        c->dom_ms = 0.f; c->dom_name[0] = 0;
        for (uint32_t i = 0; i < c->n_lev; ++i) {
            float t = 0.f;
            if (hipEventElapsedTime(&t, c->lev[i].a, c->lev[i].b) == hipSuccess && t > c->dom_ms) { c->dom_ms = t; memcpy(c->dom_name, c->lev[i].name, sizeof c->dom_name); }
        }
    }
    return BB_OK;
}

// One chunk of a host batch, synchronous: H2D, pipeline, rows D2H.  Everything goes over the context's own stream: a copy on the null stream
// waits for every other context's stream on the device, and a binding's worker threads (one context each) would run one after the other.
// (Measured, round 6: HIP's own path for pageable memory moves a 4 MB batch in 0.29 ms; copying it through page-locked slots of the context
// first took 1.3 ms — not kept.)
// A small host batch, every group on the filter pass: the reads cut into 512-byte segments by the host (bb_ctx::small_seg_max).  The table — one
// (read, segment) pair per lane of k_flank_filter, longest segments first, what k_len_scatter builds on the device for batches of differing
// lengths — is written behind the offsets in a page-locked buffer and goes up with them in ONE copy.  1: done (offsets uploaded too), 0: not
// applicable (the caller uploads the offsets), < 0: error.
static int bb_small_segments(bb_ctx* c, const uint64_t* offsets, uint32_t n) {
    c->host_vtab_valid = false;
    if (!c->small_seg_max || n > c->small_seg_max || n > c->defer_max || !c->seg_lines || c->seg_from_env || c->scan_filter == 0) return 0;
    for (size_t g = 0; g < c->groups.size(); ++g)   // the full scan's segments need more than a table (cells per cut read, k_seg_fold): filtered groups only
        if (!(c->gdev[g].filt_rows > 0) || (c->scan_off[g] > 0 && c->scan_filter != 1)) return 0;
    constexpr uint32_t SL = 4u, SA = 8u;
    uint64_t n_virtual = 0;
    uint32_t max_nl = 0, bins[SA + 1] = {};
    auto lines_of = [&](uint32_t i) { const uint64_t off = offsets[i]; const uint32_t len = (uint32_t)(offsets[i + 1] - off); return len ? (uint32_t)(((off & 127u) + len + 127u) >> 7) : 0u; };
    for (uint32_t i = 0; i < n; ++i) {
        const uint32_t nl = lines_of(i);
        max_nl = std::max(max_nl, nl);
        if (nl > SA) { const uint32_t ns = (nl + SL - 1u) / SL; bins[SL] += ns - 1u; ++bins[nl - (ns - 1u) * SL]; n_virtual += ns; }
        else { ++bins[nl]; ++n_virtual; }
    }
    if (max_nl <= SA || n_virtual > 32ull * n || ((uintptr_t)c->d_in_bases & 127u)) return 0;   // nothing to cut / reads so long that the table would outweigh the gain
    int r;
    if ((r = grow(c, c->d_in_offsets, c->cap_in_offsets, (uint64_t)n + 1 + n_virtual))) return r;
    if (c->cap_h_offs < (uint64_t)n + 1 + n_virtual) {
        if (c->h_offs) HIPCHK(c, hipHostFree(c->h_offs));
        c->h_offs = nullptr; c->cap_h_offs = 0;
        const uint64_t cap = ((uint64_t)n + 1 + n_virtual) * 5 / 4 + 64;
        HIPCHK(c, hipHostMalloc((void**)&c->h_offs, cap * 8, hipHostMallocDefault));
        c->cap_h_offs = cap;
    }
    memcpy(c->h_offs, offsets, ((uint64_t)n + 1) * 8);
    uint2* vt = reinterpret_cast<uint2*>(c->h_offs + n + 1);
    uint32_t at[SA + 1], pos = 0;
    for (int b = (int)SA; b >= 0; --b) { at[b] = pos; pos += bins[b]; }   // falling length: a wave's lanes finish together
    for (uint32_t i = 0; i < n; ++i) {
        const uint32_t nl = lines_of(i);
        if (nl > SA) {
            const uint32_t ns = (nl + SL - 1u) / SL;
            for (uint32_t t = 0; t + 1u < ns; ++t) vt[at[SL]++] = make_uint2(i, t);
            vt[at[nl - (ns - 1u) * SL]++] = make_uint2(i, ns - 1u);
        } else vt[at[nl]++] = make_uint2(i, 0u);
    }
    HIPCHK(c, hipMemcpyAsync(c->d_in_offsets, c->h_offs, ((uint64_t)n + 1 + n_virtual) * 8, hipMemcpyHostToDevice, c->stream));
    c->host_vtab = reinterpret_cast<const uint2*>(c->d_in_offsets + n + 1); c->host_n_virtual = (uint32_t)n_virtual; c->host_vtab_valid = true;
    return 1;
}

static int bb_upload_offsets(bb_ctx* c, const uint64_t* offsets, uint32_t n) {
    const int r = bb_small_segments(c, offsets, n);
    if (r < 0) return r;
    if (r == 0) HIPCHK(c, hipMemcpyAsync(c->d_in_offsets, offsets, ((uint64_t)n + 1) * 8, hipMemcpyHostToDevice, c->stream));
    return BB_OK;   // (r == 1: they went up with the segment table behind them)
}

// packed != nullptr: the reads come two bases per byte (bb_pack.h; read i at packed + packed_offsets[i], offsets zero-based in BASES) and are
// spread to one byte per base in HBM before anything looks at them — half the bytes over PCIe.
static int annotate_host_chunk(bb_ctx* c, const uint8_t* bases, const uint64_t* offsets, uint32_t n, bb_row* rows, uint64_t rows_cap,
                               uint64_t* n_rows, const uint8_t* packed = nullptr, const uint64_t* packed_offsets = nullptr) {
    const uint64_t nb = offsets[n];
    int r;
    const double t0 = c->phases ? bb_now() : 0.0;
    if ((r = grow(c, c->d_in_bases, c->cap_in_bases, nb + 16))) return r;
    if ((r = grow(c, c->d_in_offsets, c->cap_in_offsets, (uint64_t)n + 1))) return r;
    uint64_t want = rows_cap < (uint64_t)n * 4 + 64 ? rows_cap : (uint64_t)n * 4 + 64;
    if ((r = grow(c, c->d_out_rows, c->cap_out_rows, want ? want : 1))) return r;
    if (packed) {
        const uint64_t pb = packed_offsets[n] - packed_offsets[0];
        if ((r = grow(c, c->d_in_packed, c->cap_in_packed, pb + 16))) return r;
        if ((r = grow(c, c->d_in_poffs, c->cap_in_poffs, (uint64_t)n + 1))) return r;
        HIPCHK(c, hipMemcpyAsync(c->d_in_packed, packed + packed_offsets[0], pb, hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hipMemcpyAsync(c->d_in_poffs, packed_offsets, ((uint64_t)n + 1) * 8, hipMemcpyHostToDevice, c->stream));
        if ((r = bb_upload_offsets(c, offsets, n))) return r;
        // (the kernel subtracts nothing: the device copy of the packed bytes starts at the batch's first read)
        bb_launch_unpack_reads(c->stream, c->d_in_packed - packed_offsets[0], c->d_in_poffs, c->d_in_offsets, n, c->d_in_bases);
        HIPCHK(c, hipGetLastError());
    } else {
        HIPCHK(c, hipMemcpyAsync(c->d_in_bases, bases, nb, hipMemcpyHostToDevice, c->stream));
        if ((r = bb_upload_offsets(c, offsets, n))) return r;
    }
    if (n <= c->host_len_max && ((uintptr_t)c->d_in_bases & 127u) == 0) {
        // the batch's length statistics from the offsets in hand (bb_host_lenstat) instead of a kernel and a round trip
        if (!c->host_len) c->host_len = new bb_lenstat;
        const uint32_t seg_lines = c->seg_lines ? c->seg_lines : 32u, split_above = c->seg_lines ? c->split_above : 0xFFFFFFFFu;
        bb_host_lenstat(offsets, n, seg_lines, split_above, c->host_len);
        c->host_len_valid = true;
    }
    const double t1 = c->phases ? bb_now() : 0.0;
    uint64_t dev_cap = c->cap_out_rows < rows_cap ? c->cap_out_rows : rows_cap;
    c->spec_dst = rows; c->spec_cap = rows_cap;   // a deferred batch copies its first rows before its one wait
    if (!c->host_len_valid) c->host_vtab_valid = false;   // (the table rides on the host's length statistics)
    r = bb_annotate_batch_dev(c, c->d_in_bases, c->d_in_offsets, n, c->d_out_rows, dev_cap, n_rows);
    c->host_len_valid = false; c->host_vtab_valid = false;
    uint64_t have = c->spec_done;
    if (r == BB_E_CAPACITY && *n_rows <= rows_cap) {  // staging buffer was the limit, not the caller's
        if ((r = grow(c, c->d_out_rows, c->cap_out_rows, *n_rows))) return r;
        r = bb_annotate_batch_dev(c, c->d_in_bases, c->d_in_offsets, n, c->d_out_rows, c->cap_out_rows, n_rows);
        have = 0;
    }
    if (r != BB_OK) return r;
    const double t2 = c->phases ? bb_now() : 0.0;
    if (*n_rows > have) {
        HIPCHK(c, hipMemcpyAsync(rows + have, c->d_out_rows + have, (*n_rows - have) * sizeof(bb_row), hipMemcpyDeviceToHost, c->stream));
        BB_SYNC(c, c->stream);
    }
    if (c->phases) { const double t3 = bb_now(); c->ph[0] += t1 - t0; c->ph[1] += t2 - t1; c->ph[2] += t3 - t2; ++c->ph_calls; }
    return BB_OK;
}

// Host-pointer boundary.  Large batches are cut into ~256 MB chunks and the upload of chunk i+1 (a
// helper thread with its own stream and a second pair of input buffers) overlaps the kernels and the
// row download of chunk i; the hot path itself is unchanged (bb_annotate_batch_dev per chunk).
int bb_annotate_batch(bb_ctx* c, const uint8_t* bases, const uint64_t* offsets, uint32_t n, bb_row* rows, uint64_t rows_cap,
                      uint64_t* n_rows) {
    if (!c || !offsets || !n_rows || (!bases && n)) return BB_E_INVALID;
    *n_rows = 0;
    if (n == 0) return BB_OK;
    HIPCHK(c, hipSetDevice(c->device));
    bb_call_scope scope(c);
    const uint64_t total_bytes = offsets[n] - offsets[0];
    const uint64_t chunk_bytes = 256ull << 20;
    if (total_bytes <= 2 * chunk_bytes || n < 8192) {
        if (offsets[0] == 0) return annotate_host_chunk(c, bases, offsets, n, rows, rows_cap, n_rows);
        std::vector<uint64_t> rel((size_t)n + 1);
        for (uint32_t i = 0; i <= n; ++i) rel[i] = offsets[i] - offsets[0];
        return annotate_host_chunk(c, bases + offsets[0], rel.data(), n, rows, rows_cap, n_rows);
    }
    // chunk boundaries (by bytes)
    std::vector<uint32_t> cut{0};
    for (uint32_t i = 1; i <= n; ++i)
        if (offsets[i] - offsets[cut.back()] >= chunk_bytes || i == n) cut.push_back(i);
    const size_t nchunks = cut.size() - 1;
    uint64_t max_bytes = 0, max_reads = 0;
    for (size_t k = 0; k < nchunks; ++k) {
        max_bytes = std::max(max_bytes, offsets[cut[k + 1]] - offsets[cut[k]]);
        max_reads = std::max<uint64_t>(max_reads, cut[k + 1] - cut[k]);
    }
    struct InBuf { uint8_t* bases = nullptr; uint64_t* offs = nullptr; };
    InBuf ib[2];
    int r = BB_OK;
    for (auto& b : ib) {
        if (hipMalloc((void**)&b.bases, max_bytes + 16) != hipSuccess || hipMalloc((void**)&b.offs, (max_reads + 1) * 8) != hipSuccess) r = BB_E_NOMEM;
    }
    std::vector<unsigned long long> counts_backup(c->counts_len);
    if (r == BB_OK && hipMemcpy(counts_backup.data(), c->d_counts, sizeof(uint64_t) * c->counts_len, hipMemcpyDeviceToHost) != hipSuccess) r = BB_E_HIP;
    if (r == BB_OK) {
        std::mutex mu;
        std::condition_variable cv;
        int ready[2] = {-1, -1};   // chunk index whose data sits in buffer b (-1 = free)
        bool stop = false;
        int copy_err = BB_OK;
        std::thread loader([&]() {
            (void)hipSetDevice(c->device);
            hipStream_t cs;
            if (hipStreamCreate(&cs) != hipSuccess) { std::lock_guard<std::mutex> g(mu); copy_err = BB_E_HIP; cv.notify_all(); return; }
            std::vector<uint64_t> rel(max_reads + 1);
            for (size_t k = 0; k < nchunks; ++k) {
                const int b = (int)(k & 1);
                {
                    std::unique_lock<std::mutex> g(mu);
                    cv.wait(g, [&] { return ready[b] == -1 || stop; });
                    if (stop) break;
                }
                const uint32_t f = cut[k], cn = cut[k + 1] - cut[k];
                for (uint32_t i = 0; i <= cn; ++i) rel[i] = offsets[f + i] - offsets[f];
                hipError_t e1 = hipMemcpyAsync(ib[b].bases, bases + offsets[f], rel[cn], hipMemcpyHostToDevice, cs);
                hipError_t e2 = hipMemcpyAsync(ib[b].offs, rel.data(), ((size_t)cn + 1) * 8, hipMemcpyHostToDevice, cs);
                hipError_t e3 = hipStreamSynchronize(cs);
                std::lock_guard<std::mutex> g(mu);
                if (e1 != hipSuccess || e2 != hipSuccess || e3 != hipSuccess) { copy_err = BB_E_HIP; cv.notify_all(); break; }
                ready[b] = (int)k;
                cv.notify_all();
            }
            (void)hipStreamDestroy(cs);
        });
        uint64_t total = 0;
        bool overflow = false;
        for (size_t k = 0; k < nchunks && r == BB_OK; ++k) {
            const int b = (int)(k & 1);
            {
                std::unique_lock<std::mutex> g(mu);
                cv.wait(g, [&] { return ready[b] == (int)k || copy_err != BB_OK; });
                if (copy_err != BB_OK) { r = copy_err; break; }
            }
            const uint32_t f = cut[k], cn = cut[k + 1] - cut[k];
            uint64_t want = (uint64_t)cn * 4 + 64, got = 0;
            if ((r = grow(c, c->d_out_rows, c->cap_out_rows, want))) break;
            r = bb_annotate_batch_dev(c, ib[b].bases, ib[b].offs, cn, c->d_out_rows, c->cap_out_rows, &got);
            if (r == BB_E_CAPACITY) {
                if ((r = grow(c, c->d_out_rows, c->cap_out_rows, got))) break;
                r = bb_annotate_batch_dev(c, ib[b].bases, ib[b].offs, cn, c->d_out_rows, c->cap_out_rows, &got);
            }
            if (r != BB_OK) break;
            if (!overflow && total + got <= rows_cap && rows) {
                if (got && hipMemcpy(rows + total, c->d_out_rows, got * sizeof(bb_row), hipMemcpyDeviceToHost) != hipSuccess) { r = BB_E_HIP; break; }
                for (uint64_t i = 0; i < got; ++i) rows[total + i].read_idx += f;   // chunk-local -> batch-local read index
            } else {
                overflow = true;  // keep going to learn the required capacity
            }
            total += got;
            {
                std::lock_guard<std::mutex> g(mu);
                ready[b] = -1;
                cv.notify_all();
            }
        }
        {
            std::lock_guard<std::mutex> g(mu);
            stop = true;
            cv.notify_all();
        }
        loader.join();
        *n_rows = total;
        if (r == BB_OK && overflow) r = BB_E_CAPACITY;
    }
    if (r != BB_OK && r != BB_E_NOMEM)  // nothing of a failed call may stay in the histogram
        (void)hipMemcpy(c->d_counts, counts_backup.data(), sizeof(uint64_t) * c->counts_len, hipMemcpyHostToDevice);
    for (auto& b : ib) { if (b.bases) (void)hipFree(b.bases); if (b.offs) (void)hipFree(b.offs); }
    if (r == BB_E_HIP && c->last_error.empty()) c->last_error = "HIP error while streaming a host batch";
    return r;
}

// The host-pointer boundary with the sequences two bases per byte (bb_pack_bases): pieces of at most ~256 MB of bases, one after the other.
int bb_annotate_batch_packed(bb_ctx* c, const uint8_t* packed, const uint64_t* packed_offsets, const uint64_t* offsets, uint32_t n, bb_row* rows,
                             uint64_t rows_cap, uint64_t* n_rows) {
    if (!c || !offsets || !packed_offsets || !n_rows || (!packed && n)) return BB_E_INVALID;
    *n_rows = 0;
    if (n == 0) return BB_OK;
    HIPCHK(c, hipSetDevice(c->device));
    bb_call_scope scope(c);
    for (uint32_t i = 0; i < n; ++i)   // a read begins at a byte of its own and takes (length + 1) / 2 of them
        if (offsets[i + 1] < offsets[i] || packed_offsets[i + 1] - packed_offsets[i] < (offsets[i + 1] - offsets[i] + 1) / 2) {
            c->last_error = "bb_annotate_batch_packed: read " + std::to_string(i) + " has fewer packed bytes than (length + 1) / 2";
            return BB_E_INVALID;
        }
    const uint64_t chunk_bytes = 256ull << 20;
    std::vector<uint64_t> rel;
    std::vector<unsigned long long> counts_backup;
    uint64_t total = 0;
    bool overflow = false;
    int r = BB_OK;
    for (uint32_t f = 0; f < n && r == BB_OK;) {
        uint32_t e = f + 1;
        while (e < n && offsets[e + 1] - offsets[f] <= chunk_bytes) ++e;
        const uint32_t cn = e - f;
        const bool whole = f == 0 && e == n && offsets[0] == 0;
        if (!whole) {
            if (counts_backup.empty()) {   // nothing of a failed call may stay in the histogram
                counts_backup.resize(c->counts_len);
                if (hipMemcpy(counts_backup.data(), c->d_counts, sizeof(uint64_t) * c->counts_len, hipMemcpyDeviceToHost) != hipSuccess) { r = BB_E_HIP; break; }
            }
            rel.resize((size_t)cn + 1);
            for (uint32_t i = 0; i <= cn; ++i) rel[i] = offsets[f + i] - offsets[f];
        }
        uint64_t got = 0;
        const uint64_t room = overflow || !rows ? 0 : rows_cap - total;
        r = annotate_host_chunk(c, nullptr, whole ? offsets : rel.data(), cn, rows ? rows + total : nullptr, room, &got, packed, packed_offsets + f);
        if (r == BB_E_CAPACITY) { overflow = true; r = BB_OK; }   // keep going to learn the required capacity
        else if (r == BB_OK && f) for (uint64_t i = 0; i < got; ++i) rows[total + i].read_idx += f;   // piece-local -> batch-local read index
        total += got;
        f = e;
    }
    *n_rows = total;
    if (r == BB_OK && overflow) r = BB_E_CAPACITY;
    if (r != BB_OK && r != BB_E_NOMEM && !counts_backup.empty())
        (void)hipMemcpy(c->d_counts, counts_backup.data(), sizeof(uint64_t) * c->counts_len, hipMemcpyHostToDevice);
    return r;
}

int bb_last_host_syncs(const bb_ctx* c) { return c ? (int)c->last_syncs : BB_E_INVALID; }
int bb_host_phases(bb_ctx* c, int enable, double* ms, uint64_t* calls) {
    if (!c) return BB_E_INVALID;
    if (ms) for (int i = 0; i < BB_N_HOST_PHASES; ++i) ms[i] = 1e3 * c->ph[i];
    if (calls) *calls = c->ph_calls;
    for (double& x : c->ph) x = 0.0;
    c->ph_calls = 0;
    c->phases = enable != 0;
    return BB_OK;
}

uint32_t bb_counts_len(const bb_ctx* c) { return c ? c->counts_len : 0; }
int bb_counts(bb_ctx* c, uint64_t* out) {
    if (!c || !out) return BB_E_INVALID;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipMemcpy(out, c->d_counts, sizeof(uint64_t) * c->counts_len, hipMemcpyDeviceToHost));
    return BB_OK;
}
uint64_t* bb_counts_dev(bb_ctx* c) { return c ? reinterpret_cast<uint64_t*>(c->d_counts) : nullptr; }
int bb_counts_reset(bb_ctx* c) {
    if (!c) return BB_E_INVALID;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipMemset(c->d_counts, 0, sizeof(uint64_t) * c->counts_len));
    return BB_OK;
}

int bb_last_barcode_stats(const bb_ctx* c, uint32_t g, uint32_t strand, uint64_t* hits, uint64_t* undecided, int* lane_kernel) {
    if (!c || g >= c->groups.size() || strand > 1) return BB_E_INVALID;
    if (hits) *hits = c->last_listed[g][strand];
    if (undecided) *undecided = c->last_undecided[g][strand];
    // what the pair's next batch runs: the one-lane-per-hit kernel of the policy's traceback class, where the group is split and the class was built
    if (lane_kernel) *lane_kernel = c->gdev[g].split[strand] && c->gdev[g].WB == 2 && !c->force_generic && !c->generic_barcode && bb_lane_eligible(c, g, strand, false);
    return BB_OK;
}

int bb_last_scan_stats(const bb_ctx* c, uint32_t g, uint64_t* flagged_pieces, uint64_t* total_pieces, int* kind) {
    if (!c || g >= c->groups.size()) return BB_E_INVALID;
    if (flagged_pieces) *flagged_pieces = c->last_flagged[g];
    if (total_pieces) *total_pieces = c->last_pieces[g];
    if (kind) *kind = c->last_scan_kind[g];
    return BB_OK;
}
int bb_filter_twin(const bb_ctx* c, uint32_t g, int* twin_of, int* shared) {
    if (!c || g >= c->groups.size()) return BB_E_INVALID;
    if (twin_of) *twin_of = c->filt_twin[g];
    if (shared) *shared = c->last_twin[g] >= 0 ? (c->filt_twin_swap[g] ? 1 : 2) : 0;
    return BB_OK;
}
int bb_last_length_stats(const bb_ctx* c, uint32_t* min_lines, uint32_t* max_lines, uint32_t* work_items) {
    if (!c) return BB_E_INVALID;
    if (min_lines) *min_lines = c->last_min_lines;
    if (max_lines) *max_lines = c->last_max_lines;
    if (work_items) *work_items = c->last_segments;
    return BB_OK;
}
int bb_n_kernels(void) { return K_COUNT; }
const char* bb_kernel_name(int k) { return k >= 0 && k < K_COUNT ? kKernelNames[k] : ""; }
float bb_last_kernel_ms(const bb_ctx* c, int k) { return c && k >= 0 && k < K_COUNT ? c->ms[k] : 0.f; }
void bb_set_timing(bb_ctx* c, int enable) { if (c) c->timing = enable != 0; }
int bb_last_dominant_kernel(const bb_ctx* c, char* name, size_t name_cap, float* ms) {
    if (!c || !name || !name_cap || !ms) return BB_E_INVALID;
    snprintf(name, name_cap, "%s", c->dom_name);
    *ms = c->dom_ms;
    return BB_OK;
}
const char* bb_last_error(const bb_ctx* c) { return c ? c->last_error.c_str() : g_create_error.c_str(); }

// ---- filter step (include/barbell_amd_filter.h) ---------------------------------------------------
int bb_filter_set(bb_ctx* c, const bb_pattern* patterns, uint32_t n_patterns, const uint32_t* label_ids) {
    if (!c || (!patterns && n_patterns) || !label_ids) return BB_E_INVALID;
    HIPCHK(c, hipSetDevice(c->device));
    std::vector<bb_pat_dev> pats;
    std::vector<bb_pat_elem_dev> elems;
    std::vector<uint8_t> ok;
    for (uint32_t p = 0; p < n_patterns; ++p) {
        if (!patterns[p].elems || patterns[p].n_elems == 0) return BB_E_INVALID;
        pats.push_back({(uint32_t)elems.size(), patterns[p].n_elems});
        std::vector<int32_t> keys;
        for (uint32_t e = 0; e < patterns[p].n_elems; ++e) {
            const bb_pattern_elem& s = patterns[p].elems[e];
            if (s.n_cuts > BB_MAX_CUTS || s.match_type > BB_RFLANK || s.relative_to > BB_REL_PREV_LEFT) return BB_E_UNSUPPORTED;
            if (s.placeholder >= 0 && std::find(keys.begin(), keys.end(), s.placeholder) == keys.end()) keys.push_back(s.placeholder);
            bb_pat_elem_dev d;
            memset(&d, 0, sizeof(d));
            d.match_type = s.match_type; d.orientation = s.orientation; d.relative_to = s.relative_to; d.n_cuts = s.n_cuts;
            d.placeholder = s.placeholder; d.lo = s.range_lo; d.hi = s.range_hi;
            for (int q = 0; q < s.n_cuts; ++q) d.cuts[q] = s.cuts[q];
            if (s.label_ok) { d.label_off = (uint32_t)ok.size(); ok.insert(ok.end(), s.label_ok, s.label_ok + c->counts_len); }
            else d.label_off = 0xFFFFFFFFu;
            elems.push_back(d);
        }
        if (keys.size() > 16) return BB_E_UNSUPPORTED;
    }
    for (void* q : {(void*)c->d_fpats, (void*)c->d_felems, (void*)c->d_flabel_ok, (void*)c->d_flabel_ids})
        if (q) (void)hipFree(q);
    c->d_fpats = nullptr; c->d_felems = nullptr; c->d_flabel_ok = nullptr; c->d_flabel_ids = nullptr;
    c->n_fpats = n_patterns;
    HIPCHK(c, hipMalloc((void**)&c->d_fpats, sizeof(bb_pat_dev) * (pats.size() + 1)));
    HIPCHK(c, hipMalloc((void**)&c->d_felems, sizeof(bb_pat_elem_dev) * (elems.size() + 1)));
    HIPCHK(c, hipMalloc((void**)&c->d_flabel_ok, ok.size() + 16));
    HIPCHK(c, hipMalloc((void**)&c->d_flabel_ids, sizeof(uint32_t) * c->counts_len));
    if (!pats.empty()) HIPCHK(c, hipMemcpy(c->d_fpats, pats.data(), sizeof(bb_pat_dev) * pats.size(), hipMemcpyHostToDevice));
    if (!elems.empty()) HIPCHK(c, hipMemcpy(c->d_felems, elems.data(), sizeof(bb_pat_elem_dev) * elems.size(), hipMemcpyHostToDevice));
    if (!ok.empty()) HIPCHK(c, hipMemcpy(c->d_flabel_ok, ok.data(), ok.size(), hipMemcpyHostToDevice));
    HIPCHK(c, hipMemcpy(c->d_flabel_ids, label_ids, sizeof(uint32_t) * c->counts_len, hipMemcpyHostToDevice));
    return BB_OK;
}

int bb_filter_rows_dev(bb_ctx* c, const bb_row* d_rows, uint64_t n_rows, bb_row_verdict* d_out) {
    if (!c || (!d_rows && n_rows) || (!d_out && n_rows)) return BB_E_INVALID;
    if (!c->d_fpats) { c->last_error = "bb_filter_set has not been called"; return BB_E_INVALID; }
    if (n_rows == 0) return BB_OK;
    HIPCHK(c, hipSetDevice(c->device));
    hipLaunchKernelGGL(k_filter, dim3((unsigned)((n_rows + 255) / 256)), dim3(256), 0, c->stream, d_rows, n_rows,
                       (const bb_group_dev*)c->d_groups, (const bb_pat_dev*)c->d_fpats, c->n_fpats, (const bb_pat_elem_dev*)c->d_felems,
                       (const uint8_t*)c->d_flabel_ok, (const uint32_t*)c->d_flabel_ids, d_out);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return BB_OK;
}

int bb_filter_rows(bb_ctx* c, const bb_row* rows, uint64_t n_rows, bb_row_verdict* out) {
    if (!c || (!rows && n_rows) || (!out && n_rows)) return BB_E_INVALID;
    if (n_rows == 0) return BB_OK;
    HIPCHK(c, hipSetDevice(c->device));
    int r;
    if ((r = grow(c, c->d_frows, c->cap_frows, n_rows))) return r;
    if ((r = grow(c, c->d_fout, c->cap_fout, n_rows))) return r;
    HIPCHK(c, hipMemcpy(c->d_frows, rows, n_rows * sizeof(bb_row), hipMemcpyHostToDevice));
    if ((r = bb_filter_rows_dev(c, c->d_frows, n_rows, c->d_fout))) return r;
    HIPCHK(c, hipMemcpy(out, c->d_fout, n_rows * sizeof(bb_row_verdict), hipMemcpyDeviceToHost));
    return BB_OK;
}

// ---- device buffers for hosts without a HIP binding ----------------------------------------------
int bb_dev_malloc(bb_ctx* c, uint64_t bytes, void** d_ptr) {
    if (!c || !d_ptr) return BB_E_INVALID;
    *d_ptr = nullptr;
    HIPCHK(c, hipSetDevice(c->device));
    if (hipMalloc(d_ptr, bytes ? bytes : 16) != hipSuccess) { (void)hipGetLastError(); c->last_error = "hipMalloc failed"; return BB_E_NOMEM; }
    return BB_OK;
}
void bb_dev_free(bb_ctx* c, void* d_ptr) {
    if (!c || !d_ptr) return;
    (void)hipSetDevice(c->device);
    (void)hipFree(d_ptr);
}
int bb_dev_download(bb_ctx* c, void* dst, const void* d_src, uint64_t bytes) {
    if (!c || ((!dst || !d_src) && bytes)) return BB_E_INVALID;
    if (!bytes) return BB_OK;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipMemcpyAsync(dst, d_src, bytes, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return BB_OK;
}
int bb_dev_upload(bb_ctx* c, void* d_dst, const void* src, uint64_t bytes) {
    if (!c || ((!d_dst || !src) && bytes)) return BB_E_INVALID;
    if (!bytes) return BB_OK;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipMemcpyAsync(d_dst, src, bytes, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return BB_OK;
}

int bb_host_malloc(bb_ctx* c, uint64_t bytes, void** ptr) {
    if (!c || !ptr) return BB_E_INVALID;
    *ptr = nullptr;
    HIPCHK(c, hipSetDevice(c->device));
    if (hipHostMalloc(ptr, bytes ? bytes : 16, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); c->last_error = "hipHostMalloc failed"; return BB_E_NOMEM; }
    return BB_OK;
}
int bb_host_malloc_on(int device, uint64_t bytes, void** ptr) {
    if (!ptr) return BB_E_INVALID;
    *ptr = nullptr;
    if (hipSetDevice(device) != hipSuccess) { (void)hipGetLastError(); return BB_E_NO_DEVICE; }
    if (hipHostMalloc(ptr, bytes ? bytes : 16, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); return BB_E_NOMEM; }
    return BB_OK;
}
void bb_host_free_on(int device, void* ptr) {
    if (!ptr) return;
    (void)hipSetDevice(device);
    (void)hipHostFree(ptr);
}
void bb_host_free(bb_ctx* c, void* ptr) {
    if (!c || !ptr) return;
    (void)hipSetDevice(c->device);
    (void)hipHostFree(ptr);
}

// ---- inspect step (include/barbell_amd_inspect.h) ------------------------------------------------
int bb_inspect_rows_dev(bb_ctx* c, const bb_row* d_rows, const bb_row_verdict* d_ver, uint64_t n_rows, uint32_t bucket_size,
                        bb_inspect_elem* d_out) {
    if (!c || (!d_rows && n_rows) || (!d_out && n_rows) || bucket_size == 0) return BB_E_INVALID;
    if (n_rows == 0) return BB_OK;
    HIPCHK(c, hipSetDevice(c->device));
    hipLaunchKernelGGL(k_inspect, dim3((unsigned)((n_rows + 255) / 256)), dim3(256), 0, c->stream, d_rows, d_ver, n_rows, bucket_size, d_out);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return BB_OK;
}

int bb_inspect_rows(bb_ctx* c, const bb_row* rows, const bb_row_verdict* ver, uint64_t n_rows, uint32_t bucket_size, bb_inspect_elem* out) {
    if (!c || (!rows && n_rows) || (!out && n_rows) || bucket_size == 0) return BB_E_INVALID;
    if (n_rows == 0) return BB_OK;
    HIPCHK(c, hipSetDevice(c->device));
    int r;
    if ((r = grow(c, c->d_frows, c->cap_frows, n_rows))) return r;
    if ((r = grow(c, c->d_fout, c->cap_fout, n_rows))) return r;
    if ((r = grow(c, c->d_iout, c->cap_iout, n_rows))) return r;
    HIPCHK(c, hipMemcpy(c->d_frows, rows, n_rows * sizeof(bb_row), hipMemcpyHostToDevice));
    if (ver) HIPCHK(c, hipMemcpy(c->d_fout, ver, n_rows * sizeof(bb_row_verdict), hipMemcpyHostToDevice));
    if ((r = bb_inspect_rows_dev(c, c->d_frows, ver ? c->d_fout : nullptr, n_rows, bucket_size, c->d_iout))) return r;
    HIPCHK(c, hipMemcpy(out, c->d_iout, n_rows * sizeof(bb_inspect_elem), hipMemcpyDeviceToHost));
    return BB_OK;
}

// ---- synthetic reads (include/barbell_amd_synth.h) ---------------------------------------------
static int synth_params_from_descs(const bb_group_desc* groups, uint32_t n_groups, uint64_t seed, uint32_t len_min, uint32_t len_max,
                                   bb_synth_params& P, std::vector<uint8_t>& table) {
    if (!groups || n_groups == 0 || n_groups > BB_MAX_GROUPS || len_min > len_max || len_min == 0) return BB_E_INVALID;
    std::vector<std::vector<std::string>> seqs(n_groups);
    for (uint32_t g = 0; g < n_groups; ++g) {
        if (groups[g].n_seqs < 2) return BB_E_ONE_QUERY;
        for (uint32_t s = 0; s < groups[g].n_seqs; ++s) {
            if (groups[g].seq_lens[s] != groups[g].seq_lens[0]) return BB_E_UNEQUAL_LEN;
            seqs[g].emplace_back((const char*)groups[g].seqs[s], groups[g].seq_lens[s]);
        }
    }
    build_synth_tables(seqs, P, table);
    P.seed = seed; P.len_min = len_min; P.len_max = len_max;
    return BB_OK;
}

int bb_synth_offsets(uint64_t seed, uint32_t len_min, uint32_t len_max, uint64_t first_read, uint32_t n, uint64_t* offsets) {
    if (!offsets || len_min > len_max || len_min == 0) return BB_E_INVALID;
    bb_synth_params P{};
    P.seed = seed; P.len_min = len_min; P.len_max = len_max;
    uint64_t o = 0;
    for (uint32_t i = 0; i < n; ++i) { offsets[i] = o; o += bb_synth_len(P, first_read + i); }
    offsets[n] = o;
    return BB_OK;
}

int bb_synth_reads_host(const bb_group_desc* groups, uint32_t n_groups, uint64_t seed, uint32_t len_min, uint32_t len_max,
                        uint64_t first_read, uint32_t n, const uint64_t* offsets, uint8_t* bases) {
    if (!offsets || !bases) return BB_E_INVALID;
    bb_synth_params P{};
    std::vector<uint8_t> table;
    int r = synth_params_from_descs(groups, n_groups, seed, len_min, len_max, P, table);
    if (r != BB_OK) return r;
    for (uint32_t i = 0; i < n; ++i)
        bb_synth_fill(P, table.data(), first_read + i, bases + offsets[i], (uint32_t)(offsets[i + 1] - offsets[i]));
    return BB_OK;
}

int bb_synth_reads_dev(bb_ctx* c, uint64_t seed, uint32_t len_min, uint32_t len_max, uint64_t first_read, uint32_t n,
                       const uint64_t* d_offsets, uint8_t* d_bases) {
    if (!c || !d_offsets || !d_bases || len_min > len_max || len_min == 0) return BB_E_INVALID;
    HIPCHK(c, hipSetDevice(c->device));
    bb_synth_params P = c->synth;
    P.seed = seed; P.len_min = len_min; P.len_max = len_max;
    hipLaunchKernelGGL(k_synth, dim3((n + 255) / 256), dim3(256), 0, c->stream, P, (const uint8_t*)c->d_synth_table, first_read, n,
                       d_offsets, d_bases);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return BB_OK;
}

}  // extern "C"

bb_ctx_view bb_ctx_get_view(bb_ctx* c) {
    return bb_ctx_view{c->device, c->stream, (const bb_group_dev*)c->d_groups, (const uint32_t*)c->d_flabel_ids, &c->last_error, &c->trim, &c->fastq, &c->format};
}

