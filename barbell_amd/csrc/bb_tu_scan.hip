// bb_tu_scan.hip — the flank scan's kernels (bb_k_scan.h) and their launches: one translation unit of libbarbell_amd.so (bb_launch.h).
#include <algorithm>
#include <cstdlib>
#include <vector>

#include "bb_launch.h"
#include "bb_k_scan.h"
#include "bb_len.h"

// exclusive scan of in[0 .. n - 2]; in[n - 1] is a place holder (read as 0), out[n - 1] the total, also left in *total where that is given
int bb_scan_u32(bb_ctx* c, const uint32_t* in, uint32_t* out, uint64_t n, uint32_t* total) {
    if (n <= (uint64_t)BB_SCAN_ONE_MAX + 1u) {
        hipLaunchKernelGGL(k_scan_one, dim3(1), dim3(1024), 0, c->stream, in, out, (uint32_t)n, total);
        HIPCHK(c, hipGetLastError());
        return BB_OK;
    }
    const uint32_t nb = (uint32_t)((n + 2047) / 2048);
    hipLaunchKernelGGL(k_scan_block, dim3(nb), dim3(256), 0, c->stream, in, out, n, c->d_sums);
    hipLaunchKernelGGL(k_scan_sums, dim3(1), dim3(64), 0, c->stream, c->d_sums, nb);
    hipLaunchKernelGGL(k_scan_add, dim3(nb), dim3(256), 0, c->stream, out, n, (const uint32_t*)c->d_sums, total);
    HIPCHK(c, hipGetLastError());
    return BB_OK;
}


namespace {
// one launch of k_flank_scan2<W> for a list of groups (all of width W), in pieces of at most 28 groups
template <int W>
void launch_scan2(bb_ctx* c, const uint8_t* d_bases, const uint64_t* d_offsets, uint32_t n, const std::vector<uint32_t>& gs) {
    for (size_t at = 0; at < gs.size(); at += sizeof(bb_glist::g)) {
        bb_glist gl{};
        gl.n = (uint32_t)std::min(gs.size() - at, sizeof(bb_glist::g));
        for (uint32_t i = 0; i < gl.n; ++i) gl.g[i] = (uint8_t)gs[at + i];
        if (c->vtab) {   // reads of differing lengths: a lane per segment (bb_len.h), counts and hits folded back by seg_fixup
            hipLaunchKernelGGL(k_flank_scan_seg<W>, dim3(bb_coscheduled_blocks(gl.n, 2u, (c->n_virtual + 255u) / 256u)), dim3(256), 0, c->stream, d_bases, d_offsets, n,
                               (const uint8_t*)c->d_tables, (const bb_group_dev*)c->d_groups, gl, (uint32_t)c->groups.size(), c->d_cnt,
                               c->d_raw, c->cap_hits, c->d_hitcount, c->vtab, c->n_virtual, c->batch_seg_lines, c->batch_split_above, (const uint32_t*)c->d_vcut, c->d_vcnt);
            continue;
        }
        hipLaunchKernelGGL(k_flank_scan2<W>, dim3(bb_coscheduled_blocks(gl.n, 2u, (n + 255u) / 256u)), dim3(256), 0, c->stream, d_bases, d_offsets, n,
                           (const uint8_t*)c->d_tables, (const bb_group_dev*)c->d_groups, gl, (uint32_t)c->groups.size(), c->d_cnt,
                           c->d_raw, c->cap_hits, c->d_hitcount);
    }
}
void launch_scan2_w(bb_ctx* c, int W, const uint8_t* d_bases, const uint64_t* d_offsets, uint32_t n, const std::vector<uint32_t>& gs) {
    if (gs.empty()) return;
    switch (W) {
        case 1: launch_scan2<1>(c, d_bases, d_offsets, n, gs); break;
        case 2: launch_scan2<2>(c, d_bases, d_offsets, n, gs); break;
        case 3: launch_scan2<3>(c, d_bases, d_offsets, n, gs); break;
        case 4: launch_scan2<4>(c, d_bases, d_offsets, n, gs); break;
        case 5: launch_scan2<5>(c, d_bases, d_offsets, n, gs); break;
        case 6: launch_scan2<6>(c, d_bases, d_offsets, n, gs); break;
        case 7: launch_scan2<7>(c, d_bases, d_offsets, n, gs); break;
        default: launch_scan2<8>(c, d_bases, d_offsets, n, gs); break;
    }
}
template <int W>
void launch_verify(bb_ctx* c, const uint8_t* d_bases, const uint64_t* d_offsets, uint32_t n, uint32_t g, const uint32_t* flags, uint64_t flag_words, bool swap) {
    const uint32_t vblocks = std::min((n + 255u) / 256u, (uint32_t)c->n_cus * 3u);  // persistent: lanes draw (read, strand) items from a queue
    hipLaunchKernelGGL(k_flank_verify<W>, dim3(vblocks, 2), dim3(256), 0, c->stream, d_bases, d_offsets, n,
                       (const uint8_t*)c->d_tables, (const bb_group_dev*)c->d_groups, g, (uint32_t)c->groups.size(),
                       flags, flag_words, c->d_cnt, c->d_raw, c->cap_hits, c->d_hitcount, c->d_vqueue + 2u * g, swap ? 1u : 0u);   // (the queue counters of every group were zeroed with the batch's control block)
}
}  // namespace

// The batch's read lengths, once per batch and before its scans (bb_len.h).  Batches of (nearly) equal reads — the benchmark's — pay one small
// kernel in a round trip the filtered scan made already; others get their segments / reads sorted by falling length.
int bb_prepare_lengths(bb_ctx* c, const uint8_t* d_bases, const uint64_t* d_offsets, uint32_t n, uint64_t* off0, uint64_t* off1) {
    c->vtab = nullptr; c->n_virtual = 0; c->n_cut_reads = 0; c->n_cut_segs = 0;
    c->batch_seg_lines = c->seg_lines; c->batch_split_above = c->split_above;
    if (!c->d_lenstat) {
        HIPCHK(c, hipMalloc((void**)&c->d_lenstat, sizeof(bb_lenstat)));
        HIPCHK(c, hipMalloc((void**)&c->d_lencur, sizeof(bb_lencur)));
        HIPCHK(c, hipHostMalloc((void**)&c->h_lencur, sizeof(bb_lencur), hipHostMallocDefault));   // (uploaded asynchronously: not from the stack)
    }
    const uint32_t seg_lines = c->seg_lines ? c->seg_lines : 32u, split_above = c->seg_lines ? c->split_above : 0xFFFFFFFFu;
    bb_lenstat st;
    if (c->host_len_valid) {   // the host-pointer form of a small batch took them from the offsets it holds (bb_host_lenstat): no kernel, no round trip
        st = *c->host_len;
        c->host_len_valid = false;
    } else {
        HIPCHK(c, hipMemsetAsync(c->d_lenstat, 0, sizeof(bb_lenstat), c->stream));
        HIPCHK(c, hipMemsetAsync(&c->d_lenstat->min_nl, 0xFF, sizeof(uint32_t), c->stream));
        hipLaunchKernelGGL(k_len_hist, dim3(std::min((n + 255u) / 256u, 1024u)), dim3(256), 0, c->stream, d_bases, d_offsets, n, seg_lines, split_above, c->d_lenstat);
        HIPCHK(c, hipMemcpyAsync(&st, c->d_lenstat, sizeof(st), hipMemcpyDeviceToHost, c->stream));
        BB_SYNC(c, c->stream);
    }
    *off0 = st.off0; *off1 = st.off1;
    c->last_min_lines = st.min_nl; c->last_max_lines = st.max_nl; c->last_segments = n;
    if (st.off1 < st.off0) { c->last_error = "offsets are not ascending"; return BB_E_INVALID; }
    if (c->host_vtab_valid) {   // a small host batch whose segment table came up with its offsets (annotate_host_chunk): 512-byte segments for the filter pass
        c->host_vtab_valid = false;
        c->vtab = c->host_vtab; c->n_virtual = c->host_n_virtual; c->last_segments = c->host_n_virtual;
        c->batch_seg_lines = 4u; c->batch_split_above = 8u;
        return BB_OK;
    }
    if (!c->seg_lines || (st.max_nl <= c->split_above && st.max_nl - st.min_nl <= 2u)) return BB_OK;   // lanes of a wave finish together as they are
    bb_lencur& cur = *c->h_lencur;   // (the last batch's upload of it has long been consumed: every batch ends with a round trip)
    uint64_t at = 0;
    for (int b = (int)BB_LEN_SEG_BINS - 1; b >= 0; --b) { cur.seg[b] = (uint32_t)at; at += st.seg[b]; }
    const uint64_t n_virtual = at;
    if (n_virtual >= 0xFFFFFFFFull) { c->last_error = "more than 2^32 read segments in a batch"; return BB_E_UNSUPPORTED; }
    int r;
    if ((r = grow(c, c->d_vtab, c->cap_vtab, n_virtual + 1))) return r;
    if ((r = grow(c, c->d_vcut, c->cap_vcut, n_virtual + 1))) return r;
    if ((r = grow(c, c->d_cutread, c->cap_cutread, (uint64_t)st.n_cut_segs + 1))) return r;
    if ((r = grow(c, c->d_cutlist, c->cap_cutlist, (uint64_t)st.n_cut_reads + 1))) return r;
    if ((r = grow(c, c->d_vcnt, c->cap_vcnt, (uint64_t)st.n_cut_segs * c->groups.size() * 2 + 1))) return r;
    cur.cut_reads = 0; cur.cut_segs = 0;
    HIPCHK(c, hipMemcpyAsync(c->d_lencur, c->h_lencur, sizeof(bb_lencur), hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(k_len_scatter, dim3((n + 255u) / 256u), dim3(256), 0, c->stream, d_bases, d_offsets, n, seg_lines, split_above, c->d_lencur, c->d_vtab,
                       c->d_vcut, c->d_cutread, c->d_cutlist);
    HIPCHK(c, hipGetLastError());
    c->n_cut_reads = st.n_cut_reads; c->n_cut_segs = st.n_cut_segs;
    c->vtab = c->d_vtab; c->n_virtual = (uint32_t)n_virtual; c->last_segments = (uint32_t)n_virtual;
    return BB_OK;
}

// The flank scan of EVERY group of the context on one batch (searcher.rs:433-438: `for group in groups { search(flank, read) }`).
// Groups are sorted by what their scan is — the full-height streaming scan (no filter window says enough, or the group's last probed batch
// flagged above the break-even), or filter + verification — and each kind goes out as ONE launch whose blocks for the same reads sit on the
// same XCD (bb_coscheduled): the batch is streamed from HBM once per kind and direction instead of once per group.  The decisions per
// group (filter or not, back-off, verification or full scan after the filter) are the ones bb_launch_scan made group by group until round 4.
// deferred: no round trip for the filter's flag counts — every filtered group is verified, and the counts (read with the batch's other
// numbers when it ends: bb_note_flag_counts) decide for the NEXT batches whether the group skips its filter pass.
int bb_launch_scans(bb_ctx* c, const uint8_t* d_bases, const uint64_t* d_offsets, uint32_t n, uint64_t flag_words, uint64_t batch_bytes, bool deferred) {
    const uint32_t G = (uint32_t)c->groups.size();
    std::vector<uint32_t> plain[9], plain2[9], filt[2];   // by width; filt[wide]
    for (uint32_t g = 0; g < G; ++g) {
        const uint64_t probed_flagged = c->last_flagged[g], probed_pieces = c->last_pieces[g];
        c->last_scan_kind[g] = 0; c->last_flagged[g] = 0; c->last_pieces[g] = 2 * ((batch_bytes + 15) / 16);
        const int W = std::min(8, std::max(1, (int)c->gdev[g].W));
        if (c->gdev[g].filt_rows > 0 && c->scan_off[g] > 0 && c->scan_filter != 1) {
            // the group's last probed batch flagged above the break-even: data like that comes in runs (a library full of adapter-like decoys), so the
            // next batches skip the filter pass that would be thrown away and the group is probed again after sixteen of them
            --c->scan_off[g];
            c->last_scan_kind[g] = 3;
            c->last_flagged[g] = probed_flagged; c->last_pieces[g] = probed_pieces;   // the counts stay those of the batch that was probed
            plain[W].push_back(g);
        } else if (c->gdev[g].filt_rows > 0) filt[(c->gdev[g].filt_mode & BB_FILT_WIDE) ? 1 : 0].push_back(g);
        else plain[W].push_back(g);
    }
    // flag regions: one per filter pass, in launch order (narrow windows first).  A group whose window mirrors another filtered group's
    // (bb_ctx::filt_twin) has no pass of its own: its verification reads the other group's flags with the strands swapped.
    std::vector<uint32_t> region(G, 0);
    const uint32_t n_filt = (uint32_t)(filt[0].size() + filt[1].size());
    std::vector<uint32_t> pass[2];
    for (uint32_t g = 0; g < G; ++g) c->last_twin[g] = -1;
    for (int wide = 0; wide < 2; ++wide)
        for (uint32_t g : filt[wide]) {
            const int a = c->filt_twin[g];
            if (a >= 0 && c->use_twins && std::find(filt[wide].begin(), filt[wide].end(), (uint32_t)a) != filt[wide].end()) c->last_twin[g] = (int8_t)a;
            else pass[wide].push_back(g);
        }
    if (n_filt) {
        // (the flag words — the filter writes the ones that hold a flag — and the flag counters were zeroed with the batch's control block: bb_zero_ctl)
        uint32_t reg = 0;
        for (int wide = 0; wide < 2; ++wide)
            for (size_t at = 0; at < pass[wide].size(); at += sizeof(bb_glist::g)) {
                bb_glist gl{};
                gl.n = (uint32_t)std::min(pass[wide].size() - at, sizeof(bb_glist::g));
                for (uint32_t i = 0; i < gl.n; ++i) { gl.g[i] = (uint8_t)pass[wide][at + i]; region[pass[wide][at + i]] = reg + i; }
                uint32_t* fl = c->d_flags + (uint64_t)reg * 2ull * flag_words;
                const dim3 grid(bb_coscheduled_blocks(gl.n, 1u, ((c->vtab ? c->n_virtual : n) + 255u) / 256u));   // a lane per read, or per segment (bb_len.h)
                if (wide)
                    hipLaunchKernelGGL(k_flank_filter<true>, grid, dim3(256), 0, c->stream, d_bases, d_offsets, n, (const uint8_t*)c->d_tables,
                                       (const bb_group_dev*)c->d_groups, gl, fl, flag_words, c->d_nflag, c->vtab, c->n_virtual, c->batch_seg_lines, c->batch_split_above);
                else
                    hipLaunchKernelGGL(k_flank_filter<false>, grid, dim3(256), 0, c->stream, d_bases, d_offsets, n, (const uint8_t*)c->d_tables,
                                       (const bb_group_dev*)c->d_groups, gl, fl, flag_words, c->d_nflag, c->vtab, c->n_virtual, c->batch_seg_lines, c->batch_split_above);
                reg += gl.n;
            }
        for (uint32_t g = 0; g < G; ++g) if (c->last_twin[g] >= 0) region[g] = region[(uint32_t)c->last_twin[g]];
    }
    for (int W = 1; W <= 8; ++W) launch_scan2_w(c, W, d_bases, d_offsets, n, plain[W]);
    if (n_filt) {
        // The choice made at bb_create on pseudo-random text is re-made on the batch in hand: the verification's cost grows with
        // the number of flagged pieces (each costs its columns plus m + k of lead-in; low-complexity text, adapter-like decoys
        // and chimeric reads flag many), the full scan's does not.  Above the break-even (measured: DESIGN.md §4) the flags are
        // dropped and the full-height streaming scan does the batch.  One round trip for all groups' counts.
        unsigned long long nf[BB_MAX_GROUPS] = {};
        if (!deferred) {
            HIPCHK(c, hipMemcpyAsync(nf, c->d_nflag, sizeof(nf), hipMemcpyDeviceToHost, c->stream));  // a failure must not be read as "no flags"
            BB_SYNC(c, c->stream);
            for (uint32_t g = 0; g < G; ++g) if (c->last_twin[g] >= 0) nf[g] = nf[(uint32_t)c->last_twin[g]];   // (the same pieces, the strands swapped)
        }
        for (int wide = 0; wide < 2; ++wide)
            for (uint32_t g : filt[wide]) {
                const int W = std::min(8, std::max(1, (int)c->gdev[g].W));
                const bool swap = c->last_twin[g] >= 0 && c->filt_twin_swap[g];
                c->last_flagged[g] = nf[g]; c->last_scan_kind[g] = 1;
                if (!deferred && c->scan_filter != 1 && (double)nf[g] > c->adapt_frac * (double)c->last_pieces[g]) {
                    c->last_scan_kind[g] = 2;
                    c->scan_off[g] = 16;
                    plain2[W].push_back(g);
                    continue;
                }
                const uint32_t* fl = c->d_flags + (uint64_t)region[g] * 2ull * flag_words;
                switch (W) {
                    case 1: launch_verify<1>(c, d_bases, d_offsets, n, g, fl, flag_words, swap); break;
                    case 2: launch_verify<2>(c, d_bases, d_offsets, n, g, fl, flag_words, swap); break;
                    case 3: launch_verify<3>(c, d_bases, d_offsets, n, g, fl, flag_words, swap); break;
                    case 4: launch_verify<4>(c, d_bases, d_offsets, n, g, fl, flag_words, swap); break;
                    case 5: launch_verify<5>(c, d_bases, d_offsets, n, g, fl, flag_words, swap); break;
                    case 6: launch_verify<6>(c, d_bases, d_offsets, n, g, fl, flag_words, swap); break;
                    case 7: launch_verify<7>(c, d_bases, d_offsets, n, g, fl, flag_words, swap); break;
                    default: launch_verify<8>(c, d_bases, d_offsets, n, g, fl, flag_words, swap); break;
                }
            }
        for (int W = 1; W <= 8; ++W) launch_scan2_w(c, W, d_bases, d_offsets, n, plain2[W]);
    }
    if (c->vtab && c->n_cut_reads) {   // the segmented full scans' counts and hits made the reads' (k_seg_fold, k_seg_hits)
        uint32_t gmask = 0;
        for (int W = 1; W <= 8; ++W) { for (uint32_t g : plain[W]) gmask |= 1u << g; for (uint32_t g : plain2[W]) gmask |= 1u << g; }
        if (gmask) {
            const uint32_t items = c->n_cut_reads * G * 2u;
            hipLaunchKernelGGL(k_seg_fold, dim3((items + 255u) / 256u), dim3(256), 0, c->stream, (const uint4*)c->d_cutlist, c->n_cut_reads, G, gmask, c->d_vcnt, c->d_cnt);
            hipLaunchKernelGGL(k_seg_hits, dim3(1024), dim3(256), 0, c->stream, c->d_raw, (const uint32_t*)c->d_hitcount, c->cap_hits, G, (const uint32_t*)c->d_vcnt,
                               (const uint32_t*)c->d_cutread);
        }
    }
    return BB_OK;
}

// A deferred batch's flag counts, read when the batch has ended: a group whose filter flagged more than the break-even fraction was verified all the
// same (correct, slower than the full scan would have been); its next sixteen batches go straight to the full scan, as after a batch of kind 2.
void bb_note_flag_counts(bb_ctx* c, const unsigned long long* nf) {
    for (uint32_t g = 0; g < c->groups.size(); ++g) {
        if (c->last_scan_kind[g] != 1) continue;
        const unsigned long long f = nf[c->last_twin[g] >= 0 ? (uint32_t)c->last_twin[g] : g];   // (a twin's flags are the other group's)
        c->last_flagged[g] = f;
        if (c->scan_filter != 1 && (double)f > c->adapt_frac * (double)c->last_pieces[g]) c->scan_off[g] = 16;
    }
}
