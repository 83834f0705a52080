"""Parity tests proper: the HIP path (through the C-ABI) against the CPU oracle, bit-exact rows."""
import os
import zlib

import numpy as np
import pytest

from barbell_amd import _abi
from barbell_amd.parallel import effective_cpus  # noqa: E402
from tests.common import config_groups

pytestmark = pytest.mark.gpu
NT = effective_cpus()

# Tests whose batches are far above the small-batch thresholds (same path under both settings) or that set the thresholds themselves
ONE_BATCHING = {"test_large_batch_properties_without_oracle": "small-batch", "test_small_filtered_batches_are_cut_by_the_host": "small-batch", "test_offsets_beyond_4gib_against_oracle": "small-batch",
                "test_small_batches_run_deferred": "small-batch", "test_deferred_batch_grows_its_hit_buffers": "small-batch",
                "test_lane_kernel_backs_off_when_its_bound_decides_too_little": "classic"}   # (batches of 3 000 reads: one lane per hit only there)


@pytest.fixture(autouse=True, params=["small-batch", "classic"])
def batching(request, monkeypatch):
    """Every test of this module under both treatments of a batch (round 6): as the library runs a small batch by default — DEFERRED (no round
    trip between upload and rows, the hit count stays on the device, flag counts decide for the next batch: bb_ctx::defer_max) and, up to
    4 096 reads, with one lane per (hit, barcode) in the barcode stage — and `classic`: both thresholds 0, the path every 2 M-read step of
    the benchmark takes (a round trip per decision, k_barcode_lane), on the same small inputs."""
    if ONE_BATCHING.get(request.node.originalname, request.param) != request.param:
        pytest.skip("one treatment only")
    if request.param == "classic":
        monkeypatch.setenv("BARBELL_AMD_DEFER_MAX", "0")
        monkeypatch.setenv("BARBELL_AMD_SMALL_PFX_MAX", "0")
    return request.param


def run_both(groups, bases, offsets, **kw):
    from barbell_amd import annotate as A
    from oracle import pyoracle as po

    dm = A.Demuxer(**kw)
    for g in groups:
        dm.add_query_group(g)
    got = dm.demux_packed(bases, offsets)
    okw = {}
    if "alpha" in kw:
        okw["alpha"] = kw["alpha"]
    if "min_score_frac" in kw:
        okw["min_score"] = kw["min_score_frac"]
    if "min_score_diff_frac" in kw:
        okw["min_score_diff"] = kw["min_score_diff_frac"]
    if "policy" in kw:
        okw["policy"] = kw["policy"]
    want = po.Oracle([g.as_tuple() for g in groups], **okw).annotate(bases, offsets, n_threads=NT)
    return dm, got, want


def assert_same(got, want):
    if got.tobytes() != want.tobytes():
        n = min(len(got), len(want))
        bad = [i for i in range(n) if got[i].tobytes() != want[i].tobytes()]
        msg = f"{len(got)} vs {len(want)} rows; first diffs:\n"
        for i in bad[:5]:
            msg += f"  got  {got[i]}\n  want {want[i]}\n"
        raise AssertionError(msg)


@pytest.mark.parametrize("cfg,n,lmin,lmax", [
    ("nbd96", 1500, 4000, 4000),   # configs[1] shape
    ("nbd96", 1500, 1, 700),       # ragged, incl. reads shorter than the flank and than 16 nt
    ("rbk24", 600, 600, 3999),     # configs[0]
    ("dual", 600, 4000, 4000),     # configs[3]
    ("rbk96x", 300, 600, 4000),    # configs[4] (RBK, 2 groups)
    ("nbd96x", 300, 4000, 4000),   # configs[4] literal (no-op extended)
])
def test_synthetic_parity(cfg, n, lmin, lmax):
    from barbell_amd import annotate as A

    groups = config_groups(cfg)
    bases, offsets = A.synth_reads_host(groups, 0xBA7BE11 ^ (zlib.crc32(cfg.encode()) % 1000), lmin, lmax, 0, n)
    dm, got, want = run_both(groups, bases, offsets)
    assert len(want) > 0
    assert_same(got, want)
    # histogram = rows per (group, barcode|flank)
    cnt = dm.counts()
    off = 0
    for gi, g in enumerate(groups):
        rg = got[got["group_idx"] == gi]
        for b in range(len(g.seqs)):
            assert cnt[off + b] == np.sum(rg["barcode_idx"] == b)
        assert cnt[off + len(g.seqs)] == np.sum(rg["barcode_idx"] < 0)
        off += len(g.seqs) + 1


def test_generic_kernels_match_too(monkeypatch):
    """BARBELL_AMD_GENERIC=1 forces the any-geometry kernels (private-memory move bits) that configs
    outside the register-resident limits would use; they must give the same rows."""
    from barbell_amd import annotate as A

    monkeypatch.setenv("BARBELL_AMD_GENERIC", "1")
    for cfg in ("nbd96", "dual"):
        groups = config_groups(cfg)
        bases, offsets = A.synth_reads_host(groups, 31337, 400, 2500, 0, 500)
        _, got, want = run_both(groups, bases, offsets)
        assert_same(got, want)


def test_two_word_kernel_without_prefix_split(monkeypatch):
    """BARBELL_AMD_NO_PFX=1: forward-strand hits of a splittable group (SQK-NBD114-96: 10 shared pad rows + 32
    rows per barcode) go through the two-word register kernel like the rc hits do."""
    from barbell_amd import annotate as A

    monkeypatch.setenv("BARBELL_AMD_NO_PFX", "1")
    monkeypatch.setenv("BARBELL_AMD_TRACE_FULL", "1")  # and full-height move bits in k_flank_trace (LDS mode 1)
    for cfg in ("nbd96", "dual"):
        groups = config_groups(cfg)
        bases, offsets = A.synth_reads_host(groups, 4711, 300, 2500, 0, 1500)
        _, got, want = run_both(groups, bases, offsets)
        assert_same(got, want)


@pytest.mark.parametrize("knob", [("BARBELL_AMD_NO_FAST", "1"), ("BARBELL_AMD_FAST_MARGIN", "10"), ("BARBELL_AMD_NO_TAIL", "1"),
                                  ("BARBELL_AMD_LANE", "0"), ("BARBELL_AMD_LANE", "2"), ("BARBELL_AMD_FULL_PREFIX", "1"),
                                  ("BARBELL_AMD_NO_SIDE_STREAM", "1"), ("BARBELL_AMD_TRACE_BAND16", "1")])
def test_barcode_stage_variants(monkeypatch, knob):
    """The split barcode kernel has a fast variant (score BOUNDS for every barcode, the exact score of the best-bounded
    one in k_rows, hits the bounds do not decide redone by the exact variant).  NO_FAST: exact variant only; a huge
    FAST_MARGIN: the bounds decide nothing, every hit with two candidates takes the fallback; NO_TAIL: strands that
    need trailing shared rows (rc hits of SQK-NBD114-96, both strands of the 44-row kits) use the two-word kernel;
    LANE = 0 / 2: the bounds by one lane per (hit, barcode) (k_barcode_pfx) everywhere / by one lane per hit
    (k_barcode_lane, walk-free bound, 48- and 64-column instantiations) everywhere — the default picks per group;
    FULL_PREFIX: k_bar_prefix over every hit although k_barcode_lane computes its own shared rows (the records then serve the
    exact kernel instead of k_bar_prefix_list's); NO_SIDE_STREAM: every launch on one stream; TRACE_BAND16: the 16-row band in
    k_flank_trace also where 8 rows suffice."""
    from barbell_amd import annotate as A

    monkeypatch.setenv(*knob)
    for cfg, n in (("nbd96", 1500), ("dual", 600), ("rbk96x", 300), ("rbk24", 400)):
        groups = config_groups(cfg)
        bases, offsets = A.synth_reads_host(groups, 1234, 300, 2500, 0, n)
        _, got, want = run_both(groups, bases, offsets)
        assert len(want) > n // 3
        assert_same(got, want)


@pytest.mark.parametrize("mode", ["0", "1", "ends", "wide"])
def test_scan_filter_variants(monkeypatch, mode):
    """The flank scan runs either as the full-height streaming scan (0) or as filter + windowed verification (1: forced
    for every group, also where a 15-row window says nothing and every quarter is flagged).  Same hits either way:
    ragged and tiny reads, reads shorter than the flank, constructs at the very ends (overhang), low-complexity text
    and repeats of the filter's own rows (dense flags), overhang factors 0 / 0.4 / 1; "ends": both ends of every read
    verified; "wide": the 31-row filter (a word per strand) forced."""
    from barbell_amd import annotate as A

    monkeypatch.setenv("BARBELL_AMD_SCAN_FILTER", "1" if mode in ("ends", "wide") else mode)
    if mode == "ends":  # both ends of every read scanned, whatever the filter's hints say
        monkeypatch.setenv("BARBELL_AMD_FILTER_ENDS", "1")
    if mode == "wide":  # 31-row windows, a word per strand
        monkeypatch.setenv("BARBELL_AMD_FILTER_WIDE", "1")
    for cfg, n, lmin, lmax in (("nbd96", 1500, 1, 700), ("nbd96", 800, 3000, 4200), ("dual", 500, 200, 3000), ("rbk96x", 200, 600, 2000), ("rbk24", 300, 50, 1500)):
        groups = config_groups(cfg)
        bases, offsets = A.synth_reads_host(groups, 77 + n, lmin, lmax, 0, n)
        _, got, want = run_both(groups, bases, offsets)
        assert len(want) > n // 4
        assert_same(got, want)
    groups = config_groups("nbd96")
    flank = bytes(groups[0].seqs[0])
    rng = np.random.default_rng(11)
    reads = []
    for i in range(600):
        kind = i % 6
        body = bytes(rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), int(rng.integers(0, 900))))
        if kind == 0: r = flank[int(rng.integers(0, 30)):] + body                 # construct cut at the read's start
        elif kind == 1: r = body + flank[: len(flank) - int(rng.integers(0, 30))]  # ... and at its end
        elif kind == 2: r = (b"A" * 300 + body)[: 50 + int(rng.integers(0, 700))]  # homopolymers
        elif kind == 3: r = flank[:14] * int(rng.integers(1, 40)) + body           # the filter's rows over and over
        elif kind == 4: r = body[:200] + flank + body[200:] + flank[::-1]
        else: r = body[: int(rng.integers(0, 20))]                                 # shorter than anything
        reads.append(r)
    bases = np.frombuffer(b"".join(reads), dtype=np.uint8).copy()
    offsets = np.concatenate([[0], np.cumsum([len(r) for r in reads])]).astype(np.uint64)
    for alpha in (0.0, 0.4, 1.0):
        _, got, want = run_both(groups, bases, offsets, alpha=alpha)
        assert len(want) > 100
        assert_same(got, want)
    # misaligned batch start: the flag words are addressed relative to offsets[0]
    _, got, want = run_both(groups, bases[3:], offsets[1:] - np.uint64(3))
    assert_same(got, want)
    # k = 0 with free overhang: only column 0 of a window at row 0 says that a construct hangs over the read's start, and
    # the reads start anywhere within their first streamed line (found by the 400-seed soak of test_fuzz_geometry)
    from barbell_amd import kits
    from tests.common import EX
    g0 = [kits.group_from_fasta(os.path.join(EX, "native_left.fasta"), _abi.BB_FTAG, 0)]  # 44 informative rows before the mask
    q = bytes(g0[0].seqs[5])
    reads = [q[int(rng.integers(15, 40)):] + bytes(rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), int(rng.integers(20, 400)))) for _ in range(400)]
    bases = np.frombuffer(b"".join(reads), dtype=np.uint8).copy()
    offsets = np.concatenate([[0], np.cumsum([len(r) for r in reads])]).astype(np.uint64)
    _, got, want = run_both(g0, bases, offsets, alpha=0.0)
    assert len(want) > 300
    assert_same(got, want)


def test_noisy_reads_exercise_the_fallback():
    """reads with heavy errors inside the constructs: the runner-up's bound often comes within min_score_diff of the top,
    so a good share of hits is decided by the exact kernel"""
    from barbell_amd import annotate as A

    groups = config_groups("nbd96")
    bases, offsets = A.synth_reads_host(groups, 99, 300, 1500, 0, 1500)
    rng = np.random.default_rng(3)
    b = bases.copy()
    pos = rng.random(len(b)) < 0.12          # 12 % substitutions everywhere
    b[pos] = rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), int(pos.sum()))
    _, got, want = run_both(groups, b, offsets, min_score_frac=0.05, min_score_diff_frac=0.02)
    assert len(want) > 100
    assert_same(got, want)
    _, got, want = run_both(groups, b, offsets)
    assert_same(got, want)


def test_prefix_split_with_insertions_in_the_shared_rows():
    """k_barcode_pfx walks the shared pad rows with a 16-column register window and falls back to a loop when
    the path needs more: reads with up to 12 extra bases inside the left pad (flank-max-errors 12 keeps the
    flank hit; windows then exceed 48 columns -> the 64-column variant as well)."""
    from barbell_amd import kits

    groups = kits.groups_from_kit("SQK-NBD114-96", flank_max_errors=12)
    rng = np.random.default_rng(77)
    reads = []
    for i in range(400):
        full = bytes(groups[0].seqs[int(rng.integers(0, 96))])        # flank + barcode + flank, 70 nt
        cut = int(rng.integers(5, 14))                                  # inside the 10 pad rows (flank[4:14])
        n_ins = int(rng.integers(0, 13))
        ins = bytes(rng.choice(list(b"ACGT"), size=n_ins).astype(np.uint8))
        body = full[:cut] + ins + full[cut:]
        if i % 3 == 0:                                                   # a second lump of insertions
            c2 = int(rng.integers(4, 12))
            body = body[:c2] + bytes(rng.choice(list(b"ACGT"), size=int(rng.integers(1, 6))).astype(np.uint8)) + body[c2:]
        head = bytes(rng.choice(list(b"ACGT"), size=int(rng.integers(0, 30))).astype(np.uint8))
        tail = bytes(rng.choice(list(b"ACGT"), size=int(rng.integers(100, 400))).astype(np.uint8))
        rd = head + body + tail
        if i % 4 == 1:
            rd = bytes(A_rc(rd))
        reads.append(rd)
    bases, offsets = _abi.pack_reads(reads)
    _, got, want = run_both(groups, bases, offsets)
    assert_same(got, want)
    assert len(got) > 300


def test_trace_band_edges():
    """k_flank_trace keeps only the 16-row band around the end cell's diagonal when flank-max-errors <= 6: flank
    matches whose k errors are all insertions, or all deletions, push the traced path to the band's two edges"""
    from barbell_amd import kits

    rng = np.random.default_rng(3)
    for k in (3, 6):
        groups = kits.groups_from_kit("SQK-NBD114-96", flank_max_errors=k)
        reads = []
        for i in range(300):
            full = bytearray(groups[0].seqs[int(rng.integers(0, 96))])
            e = int(rng.integers(0, k + 1))
            pos = sorted(int(x) for x in rng.integers(1, 13, size=e))       # inside the 14-nt left flank part
            if i % 2:                                                        # e insertions
                for q, p in enumerate(pos):
                    full[p + q:p + q] = bytes(rng.choice(list(b"ACGT"), size=1).astype(np.uint8))
            else:                                                            # e deletions
                for q, p in enumerate(pos):
                    if p - q < len(full):
                        del full[max(p - q, 0)]
            head = bytes(rng.choice(list(b"ACGT"), size=int(rng.integers(0, 40))).astype(np.uint8))
            tail = bytes(rng.choice(list(b"ACGT"), size=int(rng.integers(60, 300))).astype(np.uint8))
            rd = head + bytes(full) + tail
            reads.append(A_rc(rd) if i % 3 == 0 else rd)
        bases, offsets = _abi.pack_reads(reads)
        _, got, want = run_both(groups, bases, offsets)
        assert_same(got, want)
        assert len(got) > 150


def A_rc(b):
    return bytes(b.translate(bytes.maketrans(b"ACGT", b"TGCA"))[::-1])


@pytest.mark.parametrize("cfg", ["nbd96", "rbk96x"])
def test_megabase_reads_among_ordinary_ones(cfg):
    """nanopore reads reach megabases: reads of 0.3 / 1 / 2.5 M nt (glued from synthetic constructs, so flank hits lie all along them) first,
    in the middle and last in a batch of ordinary reads — the filtered scan (nbd96) and the full k = 20 scan of two groups (rbk96x): one
    lane's (read, strand) item is 600 x the others'; rows equal the oracle's"""
    from tests.common import long_batch

    groups = config_groups(cfg)
    bases, offsets = long_batch(groups, 11)
    dm, got, want = run_both(groups, bases, offsets)
    assert len(want) > 2000
    assert_same(got, want)


@pytest.mark.parametrize("cfg,seg,policy", [
    ("nbd96", None, None), ("nbd96", "4", None), ("dual", "4", None), ("rbk96x", None, None), ("rbk96x", "4", None), ("rbk96x", "8", "lm=left"),
    ("rbk24", "4", "lm=strict"), ("nbd96", "0", None)])
def test_reads_of_differing_lengths(monkeypatch, batching, cfg, seg, policy):
    """A run's reads differ in length by three orders of magnitude; the scans then give their lanes SEGMENTS of reads, sorted by falling
    length (bb_len.h), instead of reads in file order.  The filter pass cuts a read where it likes (flags are addressed by position); the
    full scan's segments divide the hits by valley (flank_scan_lane<.., SEG>): same rows as the oracle's, with the default segments
    (4 KB, reads above 8 KB cut), with 512-byte / 1 KB segments that cut nearly every read many times (BARBELL_AMD_SEG_LINES), under the
    other local-minimum rules, and with the whole thing off (=0)."""
    from tests.common import heavy_tailed_batch

    if seg is not None:
        monkeypatch.setenv("BARBELL_AMD_SEG_LINES", seg)
    groups = config_groups(cfg)
    bases, offsets = heavy_tailed_batch(groups, 1500, seed=5, scale=0.5)
    kw = {"policy": policy} if policy else {}
    dm, got, want = run_both(groups, bases, offsets, **kw)
    assert len(want) > 1000
    assert_same(got, want)
    ls = dm.length_stats()
    n = len(offsets) - 1
    assert ls["max_lines"] > 300 and ls["min_lines"] < 4
    assert ls["work_items"] == n if seg == "0" else ls["work_items"] > n + 100
    # a batch of equal reads is scanned as it always was: a lane per read, file order
    from barbell_amd import annotate as A

    b2, o2 = A.synth_reads_host(groups, 21, 2000, 2000, 0, 300)
    dm.demux_packed(b2, o2)
    # (with 512-byte segments forced these are cut as well — and in a SMALL batch whose groups all take the filter pass: the host form cuts such
    # reads into 512-byte segments itself, bb_ctx::small_seg_max)
    host_cut = batching == "small-batch" and cfg == "nbd96" and seg is None
    items = dm.length_stats()["work_items"]
    assert (300 * 4 <= items <= 300 * 5 if host_cut else items == 300) or seg not in (None, "0")   # (2000 nt: 16 or 17 lines)


def test_empty_and_tiny_reads():
    groups = config_groups("nbd96")
    reads = [b"", b"A", b"ACGT", b"", bytes(groups[0].seqs[5]), bytes(groups[0].seqs[5])[:20], b"N" * 50, b""]
    bases, offsets = _abi.pack_reads(reads)
    _, got, want = run_both(groups, bases, offsets)
    assert_same(got, want)
    assert any(r["read_idx"] == 4 and r["barcode_idx"] == 5 for r in got)


def test_iupac_lowercase_and_junk_in_reads():
    groups = config_groups("nbd96")
    rng = np.random.default_rng(5)
    from barbell_amd import annotate as A

    bases, offsets = A.synth_reads_host(groups, 99, 200, 600, 0, 400)
    b = bases.copy()
    idx = rng.choice(len(b), len(b) // 20, replace=False)
    repl = np.frombuffer(b"NnacgtRYKM-*0123xX", dtype=np.uint8)
    b[idx] = repl[rng.integers(0, len(repl), len(idx))]
    _, got, want = run_both(groups, b, offsets)
    assert_same(got, want)


def test_thresholds_and_alpha_variants():
    from barbell_amd import annotate as A

    groups = config_groups("nbd96")
    bases, offsets = A.synth_reads_host(groups, 1234, 500, 1500, 0, 500)
    for kw in (dict(alpha=0.0), dict(alpha=1.0), dict(min_score_frac=0.6, min_score_diff_frac=0.3),
               dict(min_score_frac=0.0, min_score_diff_frac=0.0)):
        _, got, want = run_both(groups, bases, offsets, **kw)
        assert_same(got, want)


def test_many_hits_per_read_and_collapse():
    # concatemer-like reads: the construct repeated back to back and overlapping, both strands
    groups = config_groups("nbd96")
    s = [bytes(x) for x in groups[0].seqs]
    comp = bytes.maketrans(b"ACGT", b"TGCA")
    rc = lambda x: x.translate(comp)[::-1]
    reads = [
        s[0] + s[1] + s[2] + b"ACGTACGTAC" * 10 + rc(s[3]) + rc(s[4]),
        (s[7] + b"TTTT") * 12,
        s[9][:30] + s[10],            # overlapping flank hits -> collapse
        s[11] + s[11][14:],            # shared prefix region
        rc(s[12]) + s[12],
    ]
    bases, offsets = _abi.pack_reads(reads)
    _, got, want = run_both(groups, bases, offsets)
    assert len(want) >= 8
    assert_same(got, want)


def test_hit_buffer_growth_and_dense_hits():
    """More flank hits than the initial hit capacity (3 per read + 1024) and more than the 4 hits a scan
    lane buffers in registers: exercises the grow-and-rerun path and the in-loop overflow emission."""
    groups = config_groups("nbd96")
    s = [bytes(x) for x in groups[0].seqs]
    comp = bytes.maketrans(b"ACGT", b"TGCA")
    rng = np.random.default_rng(3)
    reads = []
    for r in range(700):
        parts = []
        for k in range(9):
            b = s[int(rng.integers(0, 96))]
            parts.append(b if rng.random() < 0.6 else b.translate(comp)[::-1])
            parts.append(bytes(rng.choice(list(b"ACGT"), int(rng.integers(5, 60))).tolist()))
        reads.append(b"".join(parts))
    bases, offsets = _abi.pack_reads(reads)
    _, got, want = run_both(groups, bases, offsets)
    assert len(want) > 4 * len(reads)
    assert_same(got, want)


def test_large_batch_properties_without_oracle():
    """BASELINE-size shape (4 kb reads, 200 k of them) checked through size-independent properties:
    rows sorted by (read, flank start); a sub-batch gives exactly the rows of the full batch restricted
    to it (reads are independent); the histogram equals the row counts; rerun is idempotent."""
    import torch

    from barbell_amd import annotate as A

    groups = config_groups("nbd96")
    n, L = 200_000, 4000
    dm = A.Demuxer()
    for g in groups:
        dm.add_query_group(g)
    d_off = torch.arange(0, n + 1, dtype=torch.int64, device="cuda") * L
    d_bases = torch.empty(n * L, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    dm.synth_dev(0xBA7BE11 ^ 2, L, L, 0, n, d_off.data_ptr(), d_bases.data_ptr())
    d_rows = torch.empty(4 * n * 48, dtype=torch.uint8, device="cuda")

    def run(first, cnt):
        nr = dm.demux_dev(d_bases.data_ptr() + first * L, d_off.data_ptr(), cnt, d_rows.data_ptr(), 4 * n)
        return np.frombuffer(d_rows.cpu().numpy().tobytes()[: nr * 48], dtype=_abi.ROW_DTYPE).copy()

    dm.counts_reset()
    full = run(0, n)
    cnt = dm.counts()
    assert cnt.sum() == len(full) and cnt[96] == np.sum(full["barcode_idx"] < 0)
    key = full["read_idx"].astype(np.int64) * (1 << 32) + full["read_start_flank"]
    assert (np.diff(key) >= 0).all()
    assert (full["read_len"] == L).all() and (full["read_end_flank"] <= L).all()
    again = run(0, n)
    assert again.tobytes() == full.tobytes()
    first, m = 123_456, 10_000
    sub = run(first, m)
    ref = full[(full["read_idx"] >= first) & (full["read_idx"] < first + m)].copy()
    ref["read_idx"] -= first
    assert sub.tobytes() == ref.tobytes()
    # ~80 % of the synthetic reads carry a 5' construct: most of them must be tagged with the right strand
    tags = full[full["barcode_idx"] >= 0]
    assert len(tags) > 0.7 * n and set(np.unique(tags["strand"]).tolist()) == {0, 1}


@pytest.mark.parametrize("cfg,w,fast", [("nbd96", 6000, False), ("dual", 4000, True), ("rbk96x", 2000, True)])
def test_offsets_beyond_4gib_against_oracle(cfg, w, fast):
    """One full BASELINE batch (2 M x 4000 nt = 8 GB: byte offsets pass 2^32 half-way) of configs[1] / configs[3] / configs[4]'s query sets
    through the device-pointer entry point; the rows of four windows of reads — head, middle, the window straddling the 4 GiB line and the
    LAST reads of the batch — are compared bit-exact with the oracle run on those reads alone (reads are independent, so the full batch
    restricted to a window must equal the window annotated by itself).  dual: two groups, k = 5; rbk96x: two groups, k = 20 (chance hits,
    four lane launches per group, the checkpointed flank traceback) — there the oracle's bit-parallel path does the windows (itself checked
    against the scalar one in tests/test_oracle_fast.py, and here on the first 300 reads)."""
    import torch

    from barbell_amd import annotate as A
    from oracle import pyoracle as po

    groups = config_groups(cfg)
    n, L = 2_000_000, 4000
    assert (n - w) * L > 2 ** 32
    dm = A.Demuxer()
    for g in groups:
        dm.add_query_group(g)
    d_off = torch.arange(0, n + 1, dtype=torch.int64, device="cuda") * L
    d_bases = torch.empty(n * L, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    dm.synth_dev(0xBA7BE11 ^ 2, L, L, 0, n, d_off.data_ptr(), d_bases.data_ptr())
    cap = (4 if cfg == "nbd96" else 6) * n
    d_rows = torch.empty(cap * 48, dtype=torch.uint8, device="cuda")
    nr = dm.demux_dev(d_bases.data_ptr(), d_off.data_ptr(), n, d_rows.data_ptr(), cap)
    full = np.frombuffer(d_rows[: nr * 48].cpu().numpy().tobytes(), dtype=_abi.ROW_DTYPE)
    assert nr > n // 2 and int(full["read_idx"].max()) > n - 100
    assert set(np.unique(full["group_idx"]).tolist()) == set(range(len(groups)))
    orc = po.Oracle([g.as_tuple() for g in groups])
    offs = np.arange(w + 1, dtype=np.uint64) * np.uint64(L)
    for first in (0, n // 2 - w // 2, (2 ** 32) // L - w // 2, n - w):   # incl. the window straddling the 4 GiB line
        host = d_bases[first * L: (first + w) * L].cpu().numpy()
        want = orc.annotate(host, offs, n_threads=NT, fast=fast)
        got = full[(full["read_idx"] >= first) & (full["read_idx"] < first + w)].copy()
        got["read_idx"] -= first
        assert len(want) > w // 2
        assert_same(got, want)
        if fast and first == 0:
            ws = 300
            assert_same(got[got["read_idx"] < ws], orc.annotate(host[: ws * L], offs[: ws + 1], n_threads=NT))


def test_hit_buffer_grows_with_two_groups():
    """The flank-hit buffers start at 3 hits per read (+ 1024) and grow when a batch has more (the scan is redone once with the larger
    buffers): reads that each carry six constructs of the custom dual-end set's two groups — both strands, well apart — give more than
    three hits per read; rows against the oracle, and a second, ordinary batch through the same (grown) context."""
    from barbell_amd import annotate as A

    groups = config_groups("dual")
    rng = np.random.default_rng(31)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    comp = bytes.maketrans(b"ACGT", b"TGCA")
    reads = []
    n = 3000
    for i in range(n):
        parts = [acgt[rng.integers(0, 4, int(rng.integers(0, 40)))].tobytes()]
        for j in range(6):
            g = groups[(i + j) % 2]
            s = bytes(g.seqs[int(rng.integers(len(g.seqs)))])
            parts.append(s if rng.random() < 0.5 else s.translate(comp)[::-1])
            parts.append(acgt[rng.integers(0, 4, int(rng.integers(120, 260)))].tobytes())
        reads.append(b"".join(parts))
    bases, offsets = _abi.pack_reads(reads)
    dm, got, want = run_both(groups, bases, offsets)
    assert len(want) > 3 * n + 1024                      # more rows than the initial hit capacity: the buffers grew
    assert_same(got, want)
    b2, o2 = A.synth_reads_host(groups, 5, 300, 2500, 0, 2000)
    got2 = dm.demux_packed(b2, o2)
    from oracle import pyoracle as po

    want2 = po.Oracle([g.as_tuple() for g in groups]).annotate(b2, o2, n_threads=NT)
    assert_same(got2, want2)
    dm.close()


def test_wide_barcode_windows_take_the_64_column_kernels():
    """a large flank error budget makes the widest POSSIBLE barcode window exceed 48 columns; hits are split by their actual
    window (k_hit_lists): <= 48 columns -> 48-column kernels, wider -> 64-column kernels.  Reads with 6..16 bases inserted
    inside the barcode produce the wide ones."""
    from barbell_amd import annotate as A

    groups = config_groups("rbk96x")
    rng = np.random.default_rng(21)
    rnd = lambda n: bytes(rng.choice(list(b"ACGT"), n).tolist())
    reads = []
    for i in range(400):
        g = groups[i % 2]
        seq = g.seqs[int(rng.integers(len(g.seqs)))]
        cut = 16 + int(rng.integers(4, 20))                    # inside the barcode (prefix is 16 nt)
        ins = rnd(int(rng.integers(6, 17))) if i % 3 else b""
        construct = seq[:cut] + ins + seq[cut:]
        body = rnd(int(rng.integers(300, 900)))
        if i % 4 == 0:                                          # reverse-complement strand, construct at the 3' end
            comp = bytes.maketrans(b"ACGT", b"TGCA")
            reads.append(body + construct.translate(comp)[::-1] + rnd(int(rng.integers(0, 30))))
        else:
            reads.append(rnd(int(rng.integers(0, 40))) + construct + body)
    bases, offsets = _abi.pack_reads(reads)
    dm, got, want = run_both(groups, bases, offsets)
    assert_same(got, want)
    wide = (want["read_end_bar"].astype(np.int64) - want["read_start_bar"]) > 30
    assert len(want) > 300 and int(wide.sum()) > 20   # tag rows that span an inserted barcode


@pytest.mark.parametrize("seg,policy", [("4", None), ("4", "lm=left"), ("8", "lm=strict"), (None, None)])
def test_valleys_that_span_segments(monkeypatch, seg, policy):
    """The segmented full scan divides a read's hits by valley (flank_scan_lane<.., SEG>).  A low-complexity flank — (AC)15 N24 (AC)10 — on
    reads with (AC)n runs of 100 .. 3000 nt, a few of them mutated: the bottom-row cost stays within k over thousands of columns (valleys far
    longer than a 512-byte segment: the owner follows them through several segments, the lanes in between never report), with a local
    minimum every other column (hundreds of hits per (read, strand): the ordinals run on from segment to segment and past the four a lane
    buffers), plateaus under every rule.  Full scan and filtered scan, rows equal the oracle's."""
    from barbell_amd.kits import QueryGroup

    if seg is not None:
        monkeypatch.setenv("BARBELL_AMD_SEG_LINES", seg)
    rng = np.random.default_rng(77)
    rnd = lambda n: bytes(rng.choice(list(b"ACGT"), n).tolist())
    g = [QueryGroup([b"AC" * 15 + rnd(24) + b"AC" * 10 for _ in range(12)], [f"v{i}" for i in range(12)], _abi.BB_FTAG, 6)]
    reads = []
    for i in range(60):
        parts = []
        for _ in range(int(rng.integers(1, 4))):
            parts.append(rnd(int(rng.integers(50, 1500))))
            run = bytearray(b"AC" * int(rng.integers(50, 1500)))
            for _ in range(int(rng.integers(0, 4))):                      # a few substitutions / deletions inside the run
                q = int(rng.integers(0, len(run)))
                if rng.random() < 0.5: run[q:q + 1] = b"G"
                else: del run[q:q + 1]
            if rng.random() < 0.3: run[len(run) // 2:len(run) // 2] = bytes(g[0].seqs[int(rng.integers(0, 12))])   # a whole construct inside
            parts.append(bytes(run))
        if i % 7 == 0: parts = parts[1:]                                  # the run at the read's very start: left overhang inside a valley
        if i % 5 == 0: parts.append(rnd(40))
        reads.append(b"".join(parts))
    reads += [b"AC" * 4000, b"CA" * 2500 + b"T"]                          # one valley from end to end, over the overhang positions too
    bases, offsets = _abi.pack_reads(reads)
    kw = {"policy": policy} if policy else {}
    for env in ({}, {"BARBELL_AMD_SCAN_FILTER": "0"}):
        for k, v in env.items(): monkeypatch.setenv(k, v)
        dm, got, want = run_both(g, bases, offsets, **kw)
        assert len(want) > 60
        assert_same(got, want)
        # most of the hits collapse into few rows (interval.rs:4-79): the number of flank matches themselves, against the oracle's count
        from oracle import pyoracle as po

        d = po.Oracle([x.as_tuple() for x in g], **kw).annotate_diag(bases, offsets, n_threads=NT, fast=False)
        hits = sum(dm.barcode_stats(0, sd)["hits"] for sd in (0, 1))
        assert d["flank_matches"] > 20000 and hits == d["flank_matches"] - d["region_none"], (hits, d)


def test_wide_flanks_up_to_256():
    """custom adapters longer than any kit's: flanks of 150 and 230 nt (W = 5 and 8 words in the scan / trace)"""
    from barbell_amd import annotate as A
    from barbell_amd.kits import QueryGroup

    rng = np.random.default_rng(5)
    rnd = lambda n: bytes(rng.choice(list(b"ACGT"), n).tolist())
    for pre_n, suf_n, k in ((70, 30, 7), (90, 36, 8), (120, 86, 12)):   # W = 4 (two 64-bit pairs), 5, 8
        pre, suf = rnd(pre_n), rnd(suf_n)
        g = [QueryGroup([pre + rnd(24) + suf for _ in range(24)], [f"w{i}" for i in range(24)], _abi.BB_FTAG, k)]
        bases, offsets = A.synth_reads_host(g, 31 + pre_n, 700, 2500, 0, 250)
        dm, got, want = run_both(g, bases, offsets)
        assert dm.group_info(0).flank_len == pre_n + 24 + suf_n and len(want) > 100
        assert_same(got, want)


def test_iupac_queries_and_custom_geometry():
    """Custom query sets: IUPAC codes inside barcodes and flanks, a one-sided flank (no suffix), short
    barcodes with a single-word pattern (WB = 1), few and many barcodes per group."""
    from barbell_amd import annotate as A
    from barbell_amd.kits import QueryGroup

    rng = np.random.default_rng(11)

    def rnd(n, alphabet=b"ACGT"):
        return bytes(rng.choice(list(alphabet), n).tolist())

    sets = []
    # (a) IUPAC in the shared flank and in some barcodes, 40 barcodes of 16 nt
    pre, suf = b"ACGTTRGCAYGT", b"GGNTCAGWC"
    sets.append([QueryGroup([pre + rnd(16, b"ACGTACGTACGTRYN") + suf for _ in range(40)], [f"q{i}" for i in range(40)], _abi.BB_FTAG, 2)])
    # (b) prefix only (no shared suffix), 12-nt barcodes -> pattern <= 32 rows (WB = 1), 7 barcodes, auto cutoff
    pre = rnd(20)
    seqs = [pre + rnd(12) for _ in range(7)]
    seqs[0] = seqs[0][:-1] + b"A"; seqs[1] = seqs[1][:-1] + b"C"  # make sure the common suffix is empty
    sets.append([QueryGroup(seqs, [f"p{i}" for i in range(7)], _abi.BB_RTAG, None)])
    # (c) 200 barcodes in one group + a second small group
    pre, suf = rnd(18), rnd(15)
    sets.append([QueryGroup([pre + rnd(24) + suf for _ in range(200)], [f"b{i}" for i in range(200)], _abi.BB_FTAG, 4),
                 QueryGroup([rnd(9) + b"T" + rnd(10) + rnd(1, b"A") + b"GGGGCCCCAAAA" for _ in range(3)][:3], ["x", "y", "z"], _abi.BB_RTAG, 1)])
    for groups in sets:
        # the small second group of (c) needs a shared prefix/suffix: rebuild it deterministically
        if len(groups) == 2:
            p2, s2 = rnd(10), rnd(12)
            groups[1] = QueryGroup([p2 + rnd(10) + s2 for _ in range(3)], ["x", "y", "z"], _abi.BB_RTAG, 1)
        plain = [QueryGroup([bytes(q).translate(bytes.maketrans(b"RYNWSKMBDHV", b"ACGATGACAAC")) for q in g.seqs], g.labels, g.match_type, g.flank_k)
                 for g in groups]
        bases, offsets = A.synth_reads_host(plain, 5, 150, 900, 0, 500)  # reads built from concrete instances of the queries
        _, got, want = run_both(groups, bases, offsets)
        assert len(want) > 50
        assert_same(got, want)


@pytest.mark.parametrize("seed", list(range(int(os.environ.get("BARBELL_FUZZ_SEEDS", "120")))))
def test_fuzz_geometry(seed):
    """Random query geometry (barcode length, one- or two-sided flanks, group count and size, k, alpha,
    thresholds) within the library's documented limits, reads with planted noisy constructs."""
    from barbell_amd import annotate as A
    from barbell_amd.kits import QueryGroup

    rng = np.random.default_rng(1000 + seed)

    def rnd(n):
        return bytes(rng.choice(list(b"ACGT"), int(n)).tolist())

    groups = []
    for gi in range(int(rng.integers(1, 4))):
        blen = int(rng.integers(4, 31))
        pre = rnd(rng.integers(0, 41))
        suf = rnd(rng.integers(0 if len(pre) else 3, 41))
        n = int(rng.integers(2, 130))
        seqs = []
        while len(seqs) < n:
            b = rnd(blen)
            if b not in [q[len(pre):len(pre) + blen] for q in seqs]:
                seqs.append(pre + b + suf)
        # keep the shared prefix/suffix exactly pre/suf: first barcode characters must differ somewhere
        k = int(rng.integers(0, 9)) if rng.random() < 0.7 else None
        groups.append(QueryGroup(seqs, [f"g{gi}_{i}" for i in range(n)], int(rng.integers(0, 2)), k))
    try:
        probe = A.Demuxer()
        for g in groups:
            probe.add_query_group(g)
        probe.group_info(0)
    except A.BarbellError as e:
        assert e.code in (_abi.BB_E_UNSUPPORTED, _abi.BB_E_NO_BARCODE, _abi.BB_E_NO_FLANK)
        from oracle import pyoracle as po
        if e.code != _abi.BB_E_UNSUPPORTED:  # the oracle rejects the same query sets
            with pytest.raises(ValueError):
                po.Oracle([g.as_tuple() for g in groups])
        return
    bases, offsets = A.synth_reads_host(groups, 50 + seed, 80, 1500, 0, 300)
    kw = dict(alpha=float(rng.choice([0.0, 0.4, 0.7])), min_score_frac=float(rng.choice([0.1, 0.2, 0.5])),
              min_score_diff_frac=float(rng.choice([0.0, 0.1, 0.2])))
    _, got, want = run_both(groups, bases, offsets, **kw)
    assert_same(got, want)


def test_long_reads():
    """100 kb reads: many 128-byte lines per lane, offsets beyond 2^24, hits deep inside the read."""
    from barbell_amd import annotate as A

    groups = config_groups("nbd96")
    bases, offsets = A.synth_reads_host(groups, 8, 60_000, 110_000, 0, 48)
    _, got, want = run_both(groups, bases, offsets)
    assert len(want) > 30
    assert_same(got, want)


def test_streamed_host_batch_equals_device_batch():
    """A host batch larger than two 256 MB chunks goes through the overlapped upload path; rows, row
    order, read indices and the histogram must equal the single device-resident batch; a too-small row
    buffer reports the needed capacity and leaves the histogram untouched."""
    import ctypes as C

    import torch

    from barbell_amd import annotate as A
    from barbell_amd._lib import lib

    groups = config_groups("nbd96")
    n, L = 180_000, 4000  # 720 MB of bases -> 3 chunks
    dm = A.Demuxer()
    for g in groups:
        dm.add_query_group(g)
    d_off = torch.arange(0, n + 1, dtype=torch.int64, device="cuda") * L
    d_bases = torch.empty(n * L, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    dm.synth_dev(77, L, L, 0, n, d_off.data_ptr(), d_bases.data_ptr())
    d_rows = torch.empty(4 * n * 48, dtype=torch.uint8, device="cuda")
    dm.counts_reset()
    nr = dm.demux_dev(d_bases.data_ptr(), d_off.data_ptr(), n, d_rows.data_ptr(), 4 * n)
    want = np.frombuffer(d_rows.cpu().numpy().tobytes()[: nr * 48], dtype=_abi.ROW_DTYPE)
    cnt_dev = dm.counts()
    h_bases = d_bases.cpu().numpy()
    h_off = d_off.cpu().numpy().astype(np.uint64)
    dm.counts_reset()
    got = dm.demux_packed(h_bases, h_off)
    assert_same(got, want)
    assert (dm.counts() == cnt_dev).all()
    # capacity error path through the chunked code
    small = np.zeros(1000, dtype=_abi.ROW_DTYPE)
    need = C.c_uint64()
    rc = lib().bb_annotate_batch(dm._ctx(), h_bases.ctypes.data, h_off.ctypes.data, n, small.ctypes.data, len(small), C.byref(need))
    assert rc == _abi.BB_E_CAPACITY and need.value == len(want)
    assert (dm.counts() == cnt_dev).all()


def test_device_pointer_api_and_device_synth():
    import torch

    from barbell_amd import annotate as A

    groups = config_groups("nbd96")
    n = 3000
    hb, ho = A.synth_reads_host(groups, 42, 1000, 1000, 100, n)
    dm = A.Demuxer()
    for g in groups:
        dm.add_query_group(g)
    d_off = torch.from_numpy(ho.astype(np.int64)).cuda()
    d_bases = torch.empty(int(ho[-1]), dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    dm.synth_dev(42, 1000, 1000, 100, n, d_off.data_ptr(), d_bases.data_ptr())
    assert d_bases.cpu().numpy().tobytes() == hb.tobytes()          # device generator == host generator
    d_rows = torch.empty(4 * n * 48, dtype=torch.uint8, device="cuda")
    nr = dm.demux_dev(d_bases.data_ptr(), d_off.data_ptr(), n, d_rows.data_ptr(), 4 * n)
    got = np.frombuffer(d_rows.cpu().numpy().tobytes()[: nr * 48], dtype=_abi.ROW_DTYPE)
    want = dm.demux_packed(hb, ho)
    assert_same(got, want)
    # rows ordered by (read_idx, read_start_flank)
    key = got["read_idx"].astype(np.int64) * (1 << 32) + got["read_start_flank"]
    assert (np.diff(key) >= 0).all()
    # too-small row buffer is an error code, not a crash
    with pytest.raises(A.BarbellError) as e:
        dm.demux_dev(d_bases.data_ptr(), d_off.data_ptr(), n, d_rows.data_ptr(), 3)
    assert e.value.code == _abi.BB_E_CAPACITY


def test_geometry_matches_oracle_on_device_ctx():
    from barbell_amd import annotate as A
    from oracle import pyoracle as po

    for cfg in ("nbd96", "rbk24", "dual"):
        groups = config_groups(cfg)
        dm = A.Demuxer()
        for g in groups:
            dm.add_query_group(g)
        o = po.Oracle([g.as_tuple() for g in groups])
        for gi in range(len(groups)):
            a, b = dm.group_info(gi), o.info(gi)
            assert bytes(a) == bytes(b)
            assert dm.flank(gi) == o.flank(gi)
            assert dm.pattern(gi, 3, True) == o.pattern(gi, 3, True)


@pytest.mark.parametrize("mode", [1, 2, 3])
@pytest.mark.parametrize("frac", [None, "0", "1"])
def test_stress_mixes_and_adaptive_scan(monkeypatch, mode, frac, batching):
    """The read mixes bench.py's `stress` object measures (bb_synth.h: low-complexity bodies, near-copies of the constructs' shared
    prefix every ~200 nt, half the reads chimeric) against the checker, with the per-batch choice between verification and the full
    scan left alone, forced to the full scan (BARBELL_AMD_ADAPT_FRAC=0: any flag is too many) and switched off (=1)."""
    from barbell_amd import annotate as A

    if frac is not None:
        monkeypatch.setenv("BARBELL_AMD_ADAPT_FRAC", frac)
    groups = config_groups("nbd96")
    bases, offsets = A.synth_reads_host(groups, (mode << 56) | 4321, 300, 3000, 0, 900)
    dm, got, want = run_both(groups, bases, offsets)
    assert len(want) > 300
    assert_same(got, want)
    st = dm.scan_stats(0)
    if batching == "classic":
        assert st["total_pieces"] > 0 and st["kind"] == (2 if frac == "0" and st["flagged_pieces"] else 1 if frac in ("0", "1") else st["kind"])
    else:   # a deferred batch is verified whatever its filter flagged; the count decides for the batches after it
        assert st["total_pieces"] > 0 and st["kind"] == 1
    if mode == 1:
        assert st["flagged_pieces"] / st["total_pieces"] > 0.001


def test_scan_backs_off_after_an_over_flagged_batch(monkeypatch, batching):
    """A batch whose flags exceed the break-even sends the group's next sixteen batches straight to the full scan (kind 3: no filter
    pass to throw away), then the group is probed again; the rows are the same whichever scan ran.  A deferred batch (the default for
    a batch this small) learns its flag count when it has ended: it is verified (kind 1) and the sixteen batches after it skip the filter."""
    monkeypatch.setenv("BARBELL_AMD_ADAPT_FRAC", "0")   # any flag is too many: the first batch is of kind 2
    groups = config_groups("nbd96")
    bases, offsets = A_synth(groups, 977, 300, 2500, 500)
    dm, got, want = run_both(groups, bases, offsets)
    assert_same(got, want)
    probe_kind = 2 if batching == "classic" else 1
    assert dm.scan_stats(0)["kind"] == probe_kind
    probed = dm.scan_stats(0)["flagged_pieces"]
    assert probed > 0
    kinds = []
    for _ in range(18):
        assert_same(dm.demux_packed(bases, offsets), want)
        st = dm.scan_stats(0)
        kinds.append(st["kind"])
        assert st["flagged_pieces"] == probed
    assert kinds == [3] * 16 + [probe_kind, 3]
    dm.close()


def _dual_ragged(groups, seed):
    """reads of 30..4000 nt plus reads that begin / end inside a construct (matches hanging over a read end by a few rows: the windows of a
    twinned pair sit in the middle of their flanks and no read end is verified unless a flag lies near it) and bare pieces of constructs"""
    b1, o1 = A_synth(groups, seed, 30, 4000, 700)
    reads = [b1[int(o1[i]):int(o1[i + 1])].tobytes() for i in range(len(o1) - 1)]
    rng = np.random.default_rng(seed)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    comp = bytes.maketrans(b"ACGT", b"TGCA")
    for i in range(400):
        g = groups[i & 1]
        q = g.seqs[int(rng.integers(0, len(g.seqs)))]
        q = q.encode() if isinstance(q, str) else bytes(q)
        if i & 2: q = q.translate(comp)[::-1]
        body = bytes(rng.choice(acgt, int(rng.integers(0, 500))))
        cut = int(rng.integers(1, 14))
        reads.append([q[cut:] + body, body + q[:len(q) - cut], q[cut:len(q) - cut], body[:40] + q + body[40:] + q[cut:]][(i >> 2) & 3])
    order = rng.permutation(len(reads))
    return _abi.pack_reads([reads[int(i)] for i in order])


def test_twin_filter_windows(monkeypatch, batching):
    """The custom dual-end set (BASELINE configs[3]): the right-hand flank is the left-hand one's reverse complement minus a few bases, so
    bb_finalize lays the two groups' filter windows where they mirror each other and a batch runs ONE filter pass for both — the right-hand
    group's verification reads the left-hand group's flags with the strands swapped (include/barbell_amd.h: bb_filter_twin).  Same rows as
    the checker's and as with a pass per group (BARBELL_AMD_FILTER_TWINS=0: the mirrored windows, two passes) and with each group's own
    window (BARBELL_AMD_NO_TWIN_WINDOWS=1); the flag counts of the two groups are one number; after an over-flagged batch both groups back
    off together and come back together; a kit whose groups are not each other's reverse complement has no twins."""
    from oracle import pyoracle as po

    groups = config_groups("dual")
    bases, offsets = _dual_ragged(groups, 31)
    dm, got, want = run_both(groups, bases, offsets)
    assert len(want) > 1200
    assert_same(got, want)
    assert dm.filter_twin(0) == (-1, 0) and dm.filter_twin(1) == (0, 1)
    s0, s1 = dm.scan_stats(0), dm.scan_stats(1)
    assert s0["kind"] == 1 and s1["kind"] == 1 and s0["flagged_pieces"] == s1["flagged_pieces"] > 0
    hb, ho = A_synth(groups, 5, 4000, 4000, 1500)
    want_h = po.Oracle([g.as_tuple() for g in groups]).annotate(hb, ho, n_threads=NT)
    assert_same(dm.demux_packed(hb, ho), want_h)
    assert_same(dm.demux_nibbles(bases, offsets), want)
    dm.close()
    for var in ("BARBELL_AMD_FILTER_TWINS", "BARBELL_AMD_NO_TWIN_WINDOWS"):
        monkeypatch.setenv(var, "0" if var.endswith("TWINS") else "1")
        dm, got, _ = run_both(groups, bases, offsets)
        assert_same(got, want)
        assert dm.filter_twin(1) == ((0, 0) if var.endswith("TWINS") else (-1, 0))
        if var.endswith("TWINS"):
            assert dm.scan_stats(0)["flagged_pieces"] == dm.scan_stats(1)["flagged_pieces"] == s0["flagged_pieces"]   # (the same pieces, the strands swapped)
        dm.close()
        monkeypatch.delenv(var)
    # both groups back off together (any flag is too many) and are probed again together
    monkeypatch.setenv("BARBELL_AMD_ADAPT_FRAC", "0")
    dm, got, _ = run_both(groups, bases, offsets)
    assert_same(got, want)
    probe_kind = 2 if batching == "classic" else 1
    assert [dm.scan_stats(g)["kind"] for g in (0, 1)] == [probe_kind] * 2
    kinds = []
    for _ in range(17):
        assert_same(dm.demux_packed(bases, offsets), want)
        kinds.append((dm.scan_stats(0)["kind"], dm.scan_stats(1)["kind"], dm.filter_twin(1)[1]))
    assert kinds == [(3, 3, 0)] * 16 + [(probe_kind, probe_kind, 1)]
    dm.close()
    monkeypatch.delenv("BARBELL_AMD_ADAPT_FRAC")
    # two groups of one kit that share most of their flank (SQK-RBK114-96 --use-extended: they differ in their first 16 nt), filtered at k = 5:
    # windows on the same rows, one pass, no swap
    from barbell_amd import kits

    g3 = kits.groups_from_kit("SQK-RBK114-96", use_extended=True, flank_max_errors=5)
    b3, o3 = A_synth(g3, 8, 100, 3000, 800)
    dm, got, want3 = run_both(g3, b3, o3)
    assert len(want3) > 500
    assert_same(got, want3)
    assert dm.filter_twin(1) == (0, 2) and dm.scan_stats(0)["kind"] == 1
    dm.close()
    monkeypatch.setenv("BARBELL_AMD_FILTER_TWINS", "0")
    dm, got, _ = run_both(g3, b3, o3)
    assert_same(got, want3)
    assert dm.filter_twin(1) == (0, 0)
    dm.close()
    monkeypatch.delenv("BARBELL_AMD_FILTER_TWINS")
    # other kits: no twins
    for cfg in ("nbd96", "rbk96x", "rbk24"):
        g2 = config_groups(cfg)
        dm = A_demuxer(g2)
        assert all(dm.filter_twin(g) == (-1, 0) for g in range(len(g2)))
        dm.close()


def test_small_batches_run_deferred(monkeypatch):
    """What the boundary's small-batch treatment promises (include/barbell_amd.h, bb_last_host_syncs): one wait of the host per call up to
    bb_ctx::defer_max reads, the classic number of round trips beyond it, the same rows either way and whatever the thresholds."""
    groups = config_groups("nbd96")
    bases, offsets = A_synth(groups, 4242, 200, 3000, 3000)
    dm, got, want = run_both(groups, bases, offsets)
    assert_same(got, want)
    assert dm.host_syncs() == 1
    dm.close()
    # classic: the filter's flag counts, the hit count, the row count, the drained stream, the rows' copy (the read lengths come from the offsets in hand)
    for defer, pfx, syncs in (("0", "0", 5), ("100000", "0", 1), ("100000", "100000", 1), ("0", "100000", 5)):
        monkeypatch.setenv("BARBELL_AMD_DEFER_MAX", defer)
        monkeypatch.setenv("BARBELL_AMD_SMALL_PFX_MAX", pfx)
        dm, got, _ = run_both(groups, bases, offsets)
        assert_same(got, want)
        assert dm.host_syncs() == syncs, (defer, pfx, dm.host_syncs())
        dm.close()


@pytest.mark.parametrize("cfg,cut", [("nbd96", True), ("dual", True), ("rbk96x", False)])
def test_small_filtered_batches_are_cut_by_the_host(monkeypatch, cfg, cut):
    """A small host-form batch whose groups all take the filter pass (bb_ctx::small_seg_max): the host cuts the reads into 512-byte segments
    from the offsets in hand and the table goes up behind them in the same copy — 8 lanes of the filter pass per 4 kb read instead of one
    (0.57 -> 0.40 ms for a lone caller's 1 024-read call).  Same rows with the table, without it (BARBELL_AMD_SMALL_SEG_MAX=0), in the
    two-bases-per-byte form, on ragged reads and with empty ones; a context with a full-scan group (k = 20: rbk96x) is left alone (its
    segments need cells per cut read: bb_len.h)."""
    from tests.common import heavy_tailed_batch

    groups = config_groups(cfg)
    bases, offsets = A_synth(groups, 77, 3000, 4000, 1200)
    dm, got, want = run_both(groups, bases, offsets)
    assert_same(got, want)
    n = len(offsets) - 1
    items = dm.length_stats()["work_items"]
    assert items >= 6 * n if cut else items == n, items
    assert dm.host_syncs() == 1
    assert_same(dm.demux_nibbles(bases, offsets), want)
    assert (dm.length_stats()["work_items"] >= 6 * n) == cut
    # ragged: short reads whole (up to 8 lines), long ones cut, empty reads in between
    hb, ho = heavy_tailed_batch(groups, 900, seed=9, scale=0.3)
    reads = [hb[int(ho[i]):int(ho[i + 1])].tobytes() for i in range(len(ho) - 1)]
    for at in (0, 17, 400, len(reads)): reads.insert(at, b"")
    rb, ro = _abi.pack_reads(reads)
    from oracle import pyoracle as po

    want2 = po.Oracle([g.as_tuple() for g in groups]).annotate(rb, ro, n_threads=NT)
    assert_same(dm.demux_packed(rb, ro), want2)
    assert dm.length_stats()["work_items"] > len(reads) + (100 if cut else 20)   # (512-byte segments / the device's 4 KB ones)
    dm.close()
    monkeypatch.setenv("BARBELL_AMD_SMALL_SEG_MAX", "0")
    dm, got, _ = run_both(groups, bases, offsets)
    assert_same(got, want)
    assert dm.length_stats()["work_items"] == n
    assert_same(dm.demux_packed(rb, ro), want2)
    dm.close()


def test_deferred_batch_grows_its_hit_buffers():
    """A deferred batch sizes its launches by the hit buffers' capacity and learns the hit count when it has ended: more hits than room (a read
    full of constructs: the buffers are sized for three hits a read) and every kernel after the scans leaves at once, nothing is emitted or
    counted, the host grows the buffers and runs the batch again."""
    from barbell_amd import annotate as A

    groups = config_groups("nbd96")
    construct = groups[0].seqs[5]
    construct = construct.encode() if isinstance(construct, str) else bytes(construct)
    body = np.random.default_rng(5).choice(np.frombuffer(b"ACGT", dtype=np.uint8), 60).tobytes()
    read = (construct + body) * 120    # ~120 constructs in one read
    reads = [read] * 40
    bases, offsets = _abi.pack_reads(reads)
    dm, got, want = run_both(groups, bases, offsets)
    assert len(want) > 40 * 100
    assert_same(got, want)
    cnt = dm.counts()
    assert int(cnt.sum()) == len(got)     # counted once, not once per attempt
    assert_same(dm.demux_packed(bases, offsets), want)
    assert int(dm.counts().sum()) == 2 * len(got)
    dm.close()


def A_demuxer(groups, **kw):
    from barbell_amd import annotate as A

    dm = A.Demuxer(**kw)
    for g in groups:
        dm.add_query_group(g)
    return dm


def A_synth(groups, seed, lmin, lmax, n):
    from barbell_amd import annotate as A

    return A.synth_reads_host(groups, seed, lmin, lmax, 0, n)


def test_geometry_beyond_the_tuned_kernels():
    """What the reference's BarcodeGroup::new (barcodes.rs:105-197) accepts and only the any-geometry kernels compute: padded barcodes
    longer than 64 nt (three Myers words per lane), a 150-nt flank under its automatic error budget (> 63), windows wider than
    128 columns; ~600 barcodes with wide windows (the 64-column split kernel cannot hold a hit's lanes in one block)."""
    from barbell_amd import annotate as A
    from barbell_amd.kits import QueryGroup

    rng = np.random.default_rng(2024)

    def rnd(n):
        return bytes(rng.choice(list(b"ACGT"), int(n)).tolist())

    def group(pre, blen, suf, n, typ, k):
        seqs = []
        while len(seqs) < n:
            b = rnd(blen)
            if b not in [q[len(pre):len(pre) + blen] for q in seqs]:
                seqs.append(pre + b + suf)
        return QueryGroup(seqs, [f"b{i}" for i in range(n)], typ, k)

    cases = [
        [group(rnd(20), 60, rnd(15), 24, 0, 6)],        # 80-nt padded patterns
        [group(rnd(90), 20, rnd(60), 16, 0, None)],     # 170-nt flank, automatic cutoff (> 63), windows up to ~190 columns
        [group(rnd(14), 24, rnd(8), 600, 0, 8)],        # 600 barcodes, windows up to 51 columns
    ]
    for groups in cases:
        dm = A.Demuxer()
        for g in groups:
            dm.add_query_group(g)
        info = dm.group_info(0)
        bases, offsets = A.synth_reads_host(groups, 9, 200, 1500, 0, 250)
        _, got, want = run_both(groups, bases, offsets)
        assert len(want) > 100, (info.pattern_len, info.flank_k)
        assert_same(got, want)


def test_lane_kernel_backs_off_when_its_bound_decides_too_little(monkeypatch):
    """k_barcode_lane's walk-free bound is only as sharp as the shared pad rows match.  The undecided fraction of every batch is read
    back; above the threshold the (group, strand) pair takes k_barcode_pfx for the next 32 batches.  With the threshold at 0 the first
    batch with any undecided hit flips the pair; rows are the oracle's before and after."""
    from barbell_amd import annotate as A
    from oracle import pyoracle as po
    from tests.common import noisy_reads

    monkeypatch.setenv("BARBELL_AMD_LANE_FB_FRAC", "0")
    groups = config_groups("nbd96")
    orc = po.Oracle([g.as_tuple() for g in groups])
    # the first backed-off batch is a probe (round 6): k_barcode_pfx costs a hit twice the lane kernel's time, so the pair stays with it only if it
    # leaves bb_ctx::lane_pfx_gain of the hits fewer undecided — here it does not (a few per cent either way): back to the lane kernel, and no
    # back-off for the next 64 batches.  With the gain asked for at -1 the pair stays away for its 32 batches, as through round 5.
    for gain, expect in ((None, [False, True, True, True]), ("-1", [False, False, False, False])):
        if gain is not None:
            monkeypatch.setenv("BARBELL_AMD_LANE_PFX_GAIN", gain)
        dm = A.Demuxer()
        for g in groups:
            dm.add_query_group(g)
        seen = []
        for batch in range(4):
            _, bases, offsets = noisy_reads("nbd96", 77 + batch, 3000, 300, 1500, rate=0.1)
            got = dm.demux_packed(bases, offsets)
            want = orc.annotate(bases, offsets, n_threads=NT)
            assert_same(got, want)
            st = [dm.barcode_stats(0, s) for s in (0, 1)]
            assert all(x["hits"] > 500 for x in st)
            if batch == 0:
                assert all(x["undecided"] > 0 for x in st)  # noisy reads: some hits always go on to the exact pass
            seen.append([x["lane_kernel"] for x in st])
            big = [x["hits"] >= 1024 for x in st]   # (a pair with fewer hits in a batch decides nothing)
        assert any(big)
        for sd in (0, 1):
            assert [b[sd] for b in seen] == (expect if big[sd] else [True] * 4), seen
        dm.close()


def test_windows_of_exactly_64_columns():
    """A flank budget of 21 on SQK-NBD114-96 makes 64-column barcode windows possible (mask 24 + k 21 + 2 x padding 10 - 1): end
    positions 0..64 do not fit the 64-bit column masks of the register kernels, so such groups take the any-geometry kernel.
    Reads whose barcode carries 21 inserted bases produce the widest windows; rows must be the oracle's."""
    from barbell_amd import annotate as A
    from barbell_amd import kits

    groups = kits.groups_from_kit("SQK-NBD114-96", flank_max_errors=21)
    rng = np.random.default_rng(64)
    g0 = groups[0]
    reads = []
    for i in range(60):
        q = bytes(g0.seqs[int(rng.integers(0, len(g0.seqs)))])
        # the construct with 17..21 random bases inserted in the middle of the barcode (the masked region of the flank: cheap insertions)
        mid = len(q) // 2
        ins = bytes(rng.choice(list(b"ACGT"), int(rng.integers(17, 22))).tolist())
        body = bytes(rng.choice(list(b"ACGT"), int(rng.integers(150, 400))).tolist())
        reads.append(q[:mid] + ins + q[mid:] + body)
    bases, offsets = _abi.pack_reads(reads)
    dm, got, want = run_both(groups, bases, offsets)
    assert len(want) > 40
    assert max(int(r["read_end_bar"]) - int(r["read_start_bar"]) for r in want) >= 24
    assert_same(got, want)
    dm.close()


def test_dominant_kernel_is_named_and_timed():
    """bb_last_dominant_kernel: with timing on, the longest launch of the barcode stage carries the name rocprofv3 prints for it and a
    duration inside the stage's (bench.py's roofline.kernel / avg_launch_ms)."""
    from barbell_amd import annotate as A

    groups = config_groups("nbd96")
    bases, offsets = A.synth_reads_host(groups, 3, 2000, 4000, 0, 20000)
    dm = A.Demuxer()
    for g in groups:
        dm.add_query_group(g)
    dm.set_timing(True)
    dm.demux_packed(bases, offsets)
    name, ms = dm.dominant_kernel()
    stage = dm.kernel_ms()["k_barcode"]
    assert name.startswith("k_barcode_lane<48, ") and name.endswith(", 216u, false>") and 0.0 < ms <= stage * 1.05, (name, ms, stage)
    # another traceback order: a class of this build runs its own k_barcode_lane instantiation (MSID is in every build: one of the five the
    # reference's vectors leave open); a class outside the build (MDSI, unless `make CLASSES=all`) the kernels that read the order at run time
    for pol, in_build in (("trace=MSID", True), ("trace=MDSI", len(A.build_trace_classes()) == 18)):
        dm2 = A.Demuxer(policy=pol)
        for g in groups:
            dm2.add_query_group(g)
        dm2.set_timing(True)
        dm2.demux_packed(bases, offsets)
        nm2 = dm2.dominant_kernel()[0]
        if in_build:
            assert nm2.startswith("k_barcode_lane<48, ") and ", 216u, " not in nm2, nm2
        else:
            assert nm2.startswith("k_barcode_pfx<48, ") and nm2.endswith("4294967295u>"), nm2
        dm2.close()
    dm.close()


def test_large_flank_budget_lane_kernel_and_its_round3_alternative(monkeypatch, batching):
    """Groups with flank budgets above 8 (k = 20 on the rapid kits) take k_barcode_lane with the per-entry-column Match masks of the shared
    rows' walk (NM) since round 4; BARBELL_AMD_LANE_NM=0 sends them to k_barcode_pfx as in round 3.  Same rows either way, and the
    kernel choice shows in bb_last_barcode_stats."""
    from barbell_amd import annotate as A
    from tests.common import noisy_reads

    groups, bases, offsets = noisy_reads("rbk96x", 77, 400, 300, 3000, 0.03)
    kinds = {}
    for nm in ("1", "0"):
        monkeypatch.setenv("BARBELL_AMD_LANE_NM", nm)
        dm, got, want = run_both(groups, bases, offsets)
        assert len(want) > 100
        assert_same(got, want)
        kinds[nm] = dm.barcode_stats(0, 0)["lane_kernel"]
        dm.close()
    assert kinds == {"1": True, "0": False}   # (what the pair's NEXT batch takes if it is large enough for one lane per hit to pay: bb_ctx::small_pfx_max)


def test_twelve_query_groups():
    """More than eight query groups in one context (the limit until round 4; 32 now — the reference's BarcodeGroup::new has none): twelve
    custom groups of different geometries and types, reads carrying constructs of several of them."""
    from barbell_amd import annotate as A, kits

    rng = np.random.default_rng(12)
    rnd = lambda n: bytes(rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), n))
    groups = []
    for gi in range(12):
        pre, suf, blen, n = rnd(int(rng.integers(10, 30))), rnd(int(rng.integers(6, 25))), 24, int(rng.integers(13, 40))
        seqs = [pre + rnd(blen) + suf for _ in range(n)]
        groups.append(kits.QueryGroup(seqs, [f"g{gi}b{i}" for i in range(n)], _abi.BB_FTAG if gi % 2 == 0 else _abi.BB_RTAG, int(rng.integers(2, 6))))
    comp = bytes.maketrans(b"ACGT", b"TGCA")
    reads = []
    for _ in range(600):   # one to three constructs of random groups per read, either strand, random spacers
        r = rnd(int(rng.integers(0, 80)))
        for _ in range(int(rng.integers(1, 4))):
            gq = groups[int(rng.integers(0, 12))]
            s = bytes(gq.seqs[int(rng.integers(0, len(gq.seqs)))])
            r += (s if rng.random() < 0.6 else s.translate(comp)[::-1]) + rnd(int(rng.integers(100, 700)))
        reads.append(r)
    bases = np.frombuffer(b"".join(reads), dtype=np.uint8).copy()
    offsets = np.concatenate([[0], np.cumsum([len(r) for r in reads])]).astype(np.uint64)
    dm, got, want = run_both(groups, bases, offsets)
    assert len(want) > 600 and len(set(want["group_idx"].tolist())) == 12
    assert_same(got, want)
    dm.close()


def test_histogram_beyond_64_kib_of_lds():
    """k_emit's block-local histogram is 4 B per (group, barcode | flank-only) slot of the whole context; 18 groups x 1000 sequences
    are 72 KB, more than a launch gets without hipFuncSetAttribute (bb_create raises the kernel's limit; up to 32 x 1024 fits the CU).  Rows
    against the oracle, the histogram against the rows."""
    from barbell_amd import annotate as A, kits

    rng = np.random.default_rng(77)
    rnd = lambda n: bytes(rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), n))
    groups = []
    for gi in range(18):
        pre, suf = rnd(int(rng.integers(12, 24))), rnd(int(rng.integers(8, 20)))
        seqs = list({pre + rnd(24) + suf for _ in range(1000)})
        groups.append(kits.QueryGroup(seqs, [f"g{gi}b{i}" for i in range(len(seqs))], _abi.BB_FTAG if gi % 2 == 0 else _abi.BB_RTAG, 3))
    comp = bytes.maketrans(b"ACGT", b"TGCA")
    reads = []
    for _ in range(250):
        r = rnd(int(rng.integers(0, 60)))
        for _ in range(int(rng.integers(1, 3))):
            gq = groups[int(rng.integers(0, len(groups)))]
            s = bytes(gq.seqs[int(rng.integers(0, len(gq.seqs)))])
            r += (s if rng.random() < 0.6 else s.translate(comp)[::-1]) + rnd(int(rng.integers(100, 500)))
        reads.append(r)
    bases = np.frombuffer(b"".join(reads), dtype=np.uint8).copy()
    offsets = np.concatenate([[0], np.cumsum([len(r) for r in reads])]).astype(np.uint64)
    dm, got, want = run_both(groups, bases, offsets)
    assert sum(len(g.seqs) + 1 for g in groups) * 4 > 64 * 1024
    assert len(want) > 250
    assert_same(got, want)
    cnt = dm.counts()
    assert int(cnt.sum()) == len(got)
    off = 0
    for gi, g in enumerate(groups):
        rows = got[got["group_idx"] == gi]
        assert int(cnt[off + len(g.seqs)]) == int((rows["barcode_idx"] < 0).sum())
        for b in np.unique(rows["barcode_idx"][rows["barcode_idx"] >= 0]):
            assert int(cnt[off + int(b)]) == int((rows["barcode_idx"] == b).sum())
        off += len(g.seqs) + 1
    dm.close()


def test_two_bases_per_byte_at_the_boundary():
    """bb_annotate_batch_packed: every read packed by bb_pack_bases (4-bit IUPAC base sets, a read from a byte of its own) gives the rows of
    the one-byte-per-base call — on synthetic reads, on reads of odd and tiny lengths, on lower case, U, IUPAC codes, N runs and junk, and in
    pieces (a batch beyond 256 MB of bases is cut)."""
    from barbell_amd import annotate as A

    for cfg, n, lmin, lmax in (("nbd96", 1500, 1, 900), ("dual", 500, 3999, 4000), ("rbk96x", 200, 500, 3000)):
        groups = config_groups(cfg)
        bases, offsets = A.synth_reads_host(groups, 99, lmin, lmax, 0, n)
        b = bases.copy()
        rng = np.random.default_rng(7)
        for frac, alphabet in ((0.02, b"acgtu"), (0.01, b"NRYSWKMBDHVnrys"), (0.005, b"X-*.@ \t1")):
            pos = rng.random(len(b)) < frac
            b[pos] = rng.choice(np.frombuffer(alphabet, dtype=np.uint8), int(pos.sum()))
        dm, got, want = run_both(groups, b, offsets)
        assert_same(got, want)
        assert_same(dm.demux_nibbles(b, offsets), want)
        dm.close()
    groups = config_groups("nbd96")
    n, L = 80_000, 4000    # 320 MB of bases: two pieces
    bases, offsets = A.synth_reads_host(groups, 5, L, L, 0, n)
    dm = A.Demuxer()
    for g in groups:
        dm.add_query_group(g)
    a = dm.demux_packed(bases, offsets)
    b2 = dm.demux_nibbles(bases, offsets)
    assert len(a) > n // 2 and int(a["read_idx"].max()) > n - 50
    assert_same(b2, a)
    assert int(dm.counts().sum()) == 2 * len(a)
    dm.close()


def test_a_row_buffer_that_is_too_small_leaves_nothing_behind():
    """BB_E_CAPACITY from every host form of a small (deferred) batch: *n_rows says how many rows the batch has, nothing of the failed call
    stays in the histogram (k_emit checks the count the host is about to read before it emits or counts), and the call again with room
    gives the rows."""
    import ctypes as C

    from barbell_amd import annotate as A
    from barbell_amd._lib import lib

    groups = config_groups("nbd96")
    bases, offsets = A.synth_reads_host(groups, 21, 500, 3000, 0, 2000)
    dm, want, _ = run_both(groups, bases, offsets)
    dm.counts_reset()
    small = np.zeros(100, dtype=_abi.ROW_DTYPE)
    need = C.c_uint64()
    for _ in range(2):
        rc = lib().bb_annotate_batch(dm._ctx(), bases.ctypes.data, offsets.ctypes.data, 2000, small.ctypes.data, len(small), C.byref(need))
        assert rc == _abi.BB_E_CAPACITY and need.value == len(want) > 1000
        assert int(dm.counts().sum()) == 0
    assert_same(dm.demux_packed(bases, offsets), want)
    assert int(dm.counts().sum()) == len(want)
    assert_same(dm.demux_nibbles(bases, offsets), want)
    assert int(dm.counts().sum()) == 2 * len(want)
    dm.close()
