// bb_synth.h — deterministic synthetic read generator (bench/test input).
// One counter-based splitmix64 stream per read index, so any shard of the read stream can be
// regenerated on any GPU (or on the host) bit-identically.  Modelled on the DESIGN of the
// reference's simulator (benchmarks/src/simulations/sim_data.rs: random ACGT body, adapter =
// front+barcode+rear, truncated adapters, double/mid-read artefacts, a few random edits), with the
// read mix of BASELINE.md §3.
#pragma once
#include "bb_common.h"

struct bb_synth_group {
    uint32_t n_seqs, seq_len;
    uint32_t off;  // byte offset of this group's sequences (n_seqs x seq_len ASCII) in the table
};
struct bb_synth_params {
    uint64_t seed;
    uint32_t len_min, len_max;  // read length uniform in [len_min, len_max]
    uint32_t n_groups;
    bb_synth_group g[BB_MAX_GROUPS];
};

// Stress mixes (bench.py `stress`, tests): the top byte of the seed selects what the read bodies and the read mix look like —
// the filtered flank scan's cost depends on how often unrelated text comes within k edits of a window of the flank.
#define BB_SYNTH_MODE(seed) ((uint32_t)((seed) >> 56))
#define BB_SYNTH_LOWCPLX 1u    // ~30 % of every body is homopolymer / dinucleotide runs of 20..60 nt
#define BB_SYNTH_DECOYS 2u     // a near-copy (2-3 substitutions) of the constructs' shared prefix every ~200 nt
#define BB_SYNTH_ARTEFACTS 3u  // the artefact class (a second barcode mid-read) raised from 5 % to 50 % of the reads
#define BB_SYNTH_DECOYS_DENSE 4u  // the same near-copies every ~60 nt: most 16-byte pieces near a flagged one

struct bb_rng {
    uint64_t s;
    BB_HD uint64_t next() {
        uint64_t z = (s += 0x9E3779B97F4A7C15ull);
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        return z ^ (z >> 31);
    }
};
BB_HD bb_rng bb_rng_for_read(uint64_t seed, uint64_t read_index) {
    bb_rng r;
    r.s = seed ^ (read_index * 0xD1342543DE82EF95ull + 0x2545F4914F6CDD1Dull);
    r.next();
    return r;
}
BB_HD uint32_t bb_synth_len(const bb_synth_params& P, uint64_t read_index) {
    bb_rng r = bb_rng_for_read(P.seed, read_index);
    uint32_t span = P.len_max - P.len_min + 1;
    return P.len_min + (uint32_t)(r.next() % span);
}
BB_HD uint8_t bb_comp_ascii(uint8_t c) {
    switch (c) { case 'A': return 'T'; case 'C': return 'G'; case 'G': return 'C'; case 'T': return 'A'; default: return c; }
}

// writes a mutated copy of construct (group g, barcode b) at `pos`; 2% sub, 1% ins, 1% del per base
BB_HD void bb_synth_place(const bb_synth_params& P, const uint8_t* table, bb_rng& rng, uint8_t* out, uint32_t L,
                          uint32_t g, uint32_t b, bool rc, uint32_t skip, int64_t pos) {
    const char ACGT[4] = {'A', 'C', 'G', 'T'};
    const uint8_t* seq = table + P.g[g].off + (uint64_t)b * P.g[g].seq_len;
    uint32_t cl = P.g[g].seq_len;
    for (uint32_t t = skip; t < cl; ++t) {
        uint8_t c = rc ? bb_comp_ascii(seq[cl - 1 - t]) : seq[t];
        uint64_t r = rng.next();
        uint32_t u = (uint32_t)(r % 100u);
        uint8_t rb = (uint8_t)ACGT[(r >> 8) & 3];
        if (u < 2) {  // substitution (may pick the same base: then it is a silent no-op)
            c = rb;
        } else if (u < 3) {  // insertion before this base
            if (pos >= 0 && pos < (int64_t)L) out[pos] = rb;
            ++pos;
        } else if (u < 4) {  // deletion
            continue;
        }
        if (pos >= 0 && pos < (int64_t)L) out[pos] = c;
        ++pos;
    }
}

// fills out[0..L) for read `read_index`; L must equal bb_synth_len(P, read_index)
BB_HD void bb_synth_fill(const bb_synth_params& P, const uint8_t* table, uint64_t read_index, uint8_t* out, uint32_t L) {
    const char ACGT[4] = {'A', 'C', 'G', 'T'};
    bb_rng rng = bb_rng_for_read(P.seed, read_index);
    rng.next();  // length draw
    for (uint32_t p = 0; p < L; p += 32) {
        uint64_t r = rng.next();
        uint32_t lim = L - p < 32 ? L - p : 32;
        for (uint32_t q = 0; q < lim; ++q) out[p + q] = (uint8_t)ACGT[(r >> (2 * q)) & 3];
    }
    const uint32_t mode = BB_SYNTH_MODE(P.seed);
    if (mode == BB_SYNTH_LOWCPLX) {
        for (uint32_t p = 0; p < L;) {
            uint64_t r = rng.next();
            const uint32_t gap = 40u + (uint32_t)(r % 100u), run = 20u + (uint32_t)((r >> 8) % 41u);   // mean 90 random, 40 repeat
            p += gap;
            const uint8_t a = (uint8_t)ACGT[(r >> 16) & 3], b = ((r >> 20) & 1) ? (uint8_t)ACGT[(r >> 18) & 3] : a;
            for (uint32_t q = 0; q < run && p + q < L; ++q) out[p + q] = (q & 1) ? b : a;
            p += run;
        }
    } else if (mode == BB_SYNTH_DECOYS || mode == BB_SYNTH_DECOYS_DENSE) {
        const uint8_t* s0 = table + P.g[0].off;
        const uint8_t* s1 = s0 + P.g[0].seq_len;
        uint32_t lcp = 0;
        while (lcp < P.g[0].seq_len && s0[lcp] == s1[lcp]) ++lcp;
        if (lcp >= 6) {
            for (uint32_t p = 100; p + lcp < L;) {
                uint64_t r = rng.next();
                for (uint32_t q = 0; q < lcp; ++q) out[p + q] = ((r >> 32) & 1) ? bb_comp_ascii(s0[lcp - 1 - q]) : s0[q];
                for (uint32_t e = 0; e < 2u + (uint32_t)((r >> 33) & 1); ++e) out[p + (uint32_t)((r >> (8 * e)) & 0xFF) % lcp] = (uint8_t)ACGT[(r >> (24 + 2 * e)) & 3];
                p += mode == BB_SYNTH_DECOYS ? 150u + (uint32_t)((r >> 40) % 100u) : 40u + (uint32_t)((r >> 40) % 40u);
            }
        }
    }
    uint32_t cls = (uint32_t)(rng.next() % 100u);
    if (mode == BB_SYNTH_ARTEFACTS && cls >= 45 && cls < 90) cls = 97;  // 45 % of the 80 + 10 classes become artefacts
    uint32_t g5 = 0, g3 = P.n_groups > 1 ? 1 : 0;
    bool rc3 = P.n_groups == 1;
    uint32_t b1 = (uint32_t)(rng.next() % P.g[g5].n_seqs);
    uint32_t lead = (uint32_t)(rng.next() % 61u);
    if (cls < 80) {
        bb_synth_place(P, table, rng, out, L, g5, b1, false, 0, lead);
        if (rng.next() & 1) {  // native double-ended: construct also near the 3' end
            uint32_t b3 = P.n_groups > 1 ? (uint32_t)(rng.next() % P.g[g3].n_seqs) : b1;
            uint32_t tail = (uint32_t)(rng.next() % 61u);
            int64_t pos = (int64_t)L - (int64_t)P.g[g3].seq_len - 2 - (int64_t)tail;
            bb_synth_place(P, table, rng, out, L, g3, b3, rc3, 0, pos);
        }
    } else if (cls < 90) {
        // no adapter
    } else if (cls < 95) {  // 5'-truncated adapter at offset 0 (exercises the overhang cost)
        uint32_t t = 1 + (uint32_t)(rng.next() % 20u);
        bb_synth_place(P, table, rng, out, L, g5, b1, false, t, 0);
    } else {  // artefact: a second, different barcode mid-read
        bb_synth_place(P, table, rng, out, L, g5, b1, false, 0, lead);
        uint32_t b2 = (b1 + 1 + (uint32_t)(rng.next() % (P.g[g5].n_seqs - 1))) % P.g[g5].n_seqs;
        int64_t pos = (int64_t)(L / 2) + (int64_t)(rng.next() % 100u);
        bb_synth_place(P, table, rng, out, L, g5, b2, false, 0, pos);
    }
}
