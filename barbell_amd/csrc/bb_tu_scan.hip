// bb_tu_scan.hip — the flank scan's kernels (bb_k_scan.h) and their launches: one translation unit of libbarbell_amd.so (bb_launch.h).
#include <algorithm>
#include <cstdlib>

#include "bb_launch.h"
#include "bb_k_scan.h"

int bb_scan_u32(bb_ctx* c, const uint32_t* in, uint32_t* out, uint64_t n) {
    const uint32_t nb = (uint32_t)((n + 2047) / 2048);
    hipLaunchKernelGGL(k_scan_block, dim3(nb), dim3(256), 0, c->stream, in, out, n, c->d_sums);
    hipLaunchKernelGGL(k_scan_sums, dim3(1), dim3(64), 0, c->stream, c->d_sums, nb);
    hipLaunchKernelGGL(k_scan_add, dim3(nb), dim3(256), 0, c->stream, out, n, (const uint32_t*)c->d_sums);
    HIPCHK(c, hipGetLastError());
    return BB_OK;
}


namespace {
template <int W>
int launch_scan(bb_ctx* c, const uint8_t* d_bases, const uint64_t* d_offsets, uint32_t n, uint32_t g, uint64_t flag_words, uint64_t batch_bytes) {
    const uint64_t probed_flagged = c->last_flagged[g], probed_pieces = c->last_pieces[g];
    c->last_scan_kind[g] = 0; c->last_flagged[g] = 0; c->last_pieces[g] = 2 * ((batch_bytes + 15) / 16);
    if (c->gdev[g].filt_rows > 0 && c->scan_off[g] > 0 && c->scan_filter != 1) {
        // the group's last probed batch flagged above the break-even: data like that comes in runs (a library full of adapter-like decoys), so the
        // next batches skip the filter pass that would be thrown away and the group is probed again after sixteen of them
        --c->scan_off[g];
        c->last_scan_kind[g] = 3;
        c->last_flagged[g] = probed_flagged; c->last_pieces[g] = probed_pieces;   // the counts stay those of the batch that was probed
    } else if (c->gdev[g].filt_rows > 0) {
        (void)hipMemsetAsync(c->d_flags, 0, (size_t)2 * flag_words * sizeof(uint32_t), c->stream);  // the filter writes the words that hold a flag
        (void)hipMemsetAsync(c->d_nflag + g, 0, sizeof(unsigned long long), c->stream);
        if (c->gdev[g].filt_mode & BB_FILT_WIDE)
            hipLaunchKernelGGL(k_flank_filter<true>, dim3((n + 255) / 256), dim3(256), 0, c->stream, d_bases, d_offsets, n, (const uint8_t*)c->d_tables,
                               (const bb_group_dev*)c->d_groups, g, c->d_flags, flag_words, c->d_nflag + g);
        else
            hipLaunchKernelGGL(k_flank_filter<false>, dim3((n + 255) / 256), dim3(256), 0, c->stream, d_bases, d_offsets, n, (const uint8_t*)c->d_tables,
                               (const bb_group_dev*)c->d_groups, g, c->d_flags, flag_words, c->d_nflag + g);
        // The choice made at bb_create on pseudo-random text is re-made on the batch in hand: the verification's cost grows with
        // the number of flagged pieces (each costs its columns plus m + k of lead-in; low-complexity text, adapter-like decoys
        // and chimeric reads flag many), the full scan's does not.  Above the break-even (measured: DESIGN.md §4) the flags are
        // dropped and the full-height streaming scan does the batch.
        unsigned long long nf = 0;
        HIPCHK(c, hipMemcpyAsync(&nf, c->d_nflag + g, sizeof(nf), hipMemcpyDeviceToHost, c->stream));  // a failure must not be read as "no flags"
        HIPCHK(c, hipStreamSynchronize(c->stream));
        c->last_flagged[g] = nf; c->last_scan_kind[g] = 1;
        if (c->scan_filter != 1 && (double)nf > c->adapt_frac * (double)c->last_pieces[g]) {
            c->last_scan_kind[g] = 2;
            c->scan_off[g] = 16;
            hipLaunchKernelGGL(k_flank_scan2<W>, dim3((n + 255) / 256, 2), dim3(256), 0, c->stream, d_bases, d_offsets, n,
                               (const uint8_t*)c->d_tables, (const bb_group_dev*)c->d_groups, g, (uint32_t)c->groups.size(), c->d_cnt,
                               c->d_raw, c->cap_hits, c->d_hitcount);
            return BB_OK;
        }
        (void)hipMemsetAsync(c->d_vqueue, 0, 2 * sizeof(uint32_t), c->stream);
        const uint32_t vblocks = std::min((n + 255u) / 256u, (uint32_t)c->n_cus * 3u);  // persistent: lanes draw (read, strand) items from a queue
        hipLaunchKernelGGL(k_flank_verify<W>, dim3(vblocks, 2), dim3(256), 0, c->stream, d_bases, d_offsets, n,
                           (const uint8_t*)c->d_tables, (const bb_group_dev*)c->d_groups, g, (uint32_t)c->groups.size(),
                           (const uint32_t*)c->d_flags, flag_words, c->d_cnt, c->d_raw, c->cap_hits, c->d_hitcount, c->d_vqueue);
        return BB_OK;
    }
    hipLaunchKernelGGL(k_flank_scan2<W>, dim3((n + 255) / 256, 2), dim3(256), 0, c->stream, d_bases, d_offsets, n,
                       (const uint8_t*)c->d_tables, (const bb_group_dev*)c->d_groups, g, (uint32_t)c->groups.size(), c->d_cnt,
                       c->d_raw, c->cap_hits, c->d_hitcount);
    return BB_OK;
}
}  // namespace

int bb_launch_scan(bb_ctx* c, const uint8_t* d_bases, const uint64_t* d_offsets, uint32_t n, uint32_t g, uint64_t flag_words, uint64_t batch_bytes) {
    switch (c->gdev[g].W) {
        case 1: return launch_scan<1>(c, d_bases, d_offsets, n, g, flag_words, batch_bytes);
        case 2: return launch_scan<2>(c, d_bases, d_offsets, n, g, flag_words, batch_bytes);
        case 3: return launch_scan<3>(c, d_bases, d_offsets, n, g, flag_words, batch_bytes);
        case 4: return launch_scan<4>(c, d_bases, d_offsets, n, g, flag_words, batch_bytes);
        case 5: return launch_scan<5>(c, d_bases, d_offsets, n, g, flag_words, batch_bytes);
        case 6: return launch_scan<6>(c, d_bases, d_offsets, n, g, flag_words, batch_bytes);
        case 7: return launch_scan<7>(c, d_bases, d_offsets, n, g, flag_words, batch_bytes);
        default: return launch_scan<8>(c, d_bases, d_offsets, n, g, flag_words, batch_bytes);
    }
}
