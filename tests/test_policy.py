"""The switchable assumptions about sassy 0.2.1 / cigar-lodhi-rs 0.1.0 (include/barbell_amd_policy.h, hazards H1-H8 of
SURVEY.md §8c).  CPU part: every alternative of every hazard is live in the checker — it changes the answer on an input
built to discriminate it — the default is what the policy-free entry points compute, and the reference's own sassy
known-answer tests (cigar_parse.rs:104-176) hold under every policy that should not touch them.  GPU part
(`-m gpu`): the HIP path is bit-identical to the checker under each alternative, on SQK-NBD114-96 and the custom
dual-end set (searcher.rs:209-211,282-301,364-396,438 are the call sites the hazards sit under)."""
import itertools
import os

import numpy as np
import pytest

from barbell_amd import _abi
from barbell_amd.parallel import effective_cpus  # noqa: E402
from oracle import pyoracle as po
from tests.common import ALTERNATIVES, GPU_POLICIES, TRACE_CLASSES, config_groups

NT = effective_cpus()

def test_text_form_round_trips_and_rejects_nonsense():
    for alts in ALTERNATIVES.values():
        for a in alts:
            p = _abi.policy_from_str(a)
            assert _abi.policy_to_str(_abi.policy_from_str(_abi.policy_to_str(p))) == _abi.policy_to_str(p)
    assert _abi.policy_to_str(_abi.policy_from_str()) == _abi.POLICY_DEFAULT
    for bad in ("lm=up", "trace=MMID", "lodhi=3:0.5:111", "lodhi=9:0.5:1111", "lodhi=259:0.5:1111", "what=1", "rcpath=up"):
        with pytest.raises(ValueError):
            _abi.policy_from_str(bad)
        bad_p = _abi.Policy()
        bad_p.lodhi_p = 9
    assert po.lib().bbo_set_policy(bad_p) != 0


def _search(pat, text, k, alpha=None, rc=True):
    ms, h = po.search(pat, text, k, alpha=alpha, rc=rc)
    out = [(m.text_start, m.text_end, m.cost, m.strand, m.cigar) for m in ms]
    po.free_matches(h)
    return out


def test_h1_plateau_end_and_strict_minima():
    # AAAA in ..AAAAA..: ending after the 4th and after the 5th A both cost 0 -> a plateau of two minimal positions
    pat, text = b"AAAA", b"GGAAAAAGG"
    with po.policy("lm=right"):
        right = _search(pat, text, 0, rc=False)
    with po.policy("lm=left"):
        left = _search(pat, text, 0, rc=False)
    with po.policy("lm=strict"):
        strict = _search(pat, text, 0, rc=False)
    assert right == _search(pat, text, 0, rc=False)            # the default
    assert [m[1] for m in right] == [7] and [m[1] for m in left] == [6] and strict == []
    # a strict minimum is reported by all three, at the same place
    for pol in ("lm=right", "lm=left", "lm=strict"):
        with po.policy(pol):
            assert [m[:3] for m in _search(b"ACGT", b"GGACGTGG", 0, rc=False)] == [(2, 6, 0)]


def test_h2_order_of_rc_matches():
    # two reverse-complement occurrences of ACGTTG (= CAACGT forward) in one text
    text = b"TTTTCAACGTTTTTTTTTCAACGTTTTT"
    with po.policy("rc=scan"):
        a = [m for m in _search(b"ACGTTG", text, 0) if m[3] == 1]
    with po.policy("rc=fwd"):
        b = [m for m in _search(b"ACGTTG", text, 0) if m[3] == 1]
    assert len(a) == 2 and a == b[::-1] and b[0][0] < b[1][0] and a[0][0] > a[1][0]


def test_h3_traceback_preference():
    # one extra text character inside a run: the gap can sit anywhere in the run -> the preference decides the CIGAR
    seen = set()
    for pol in ["trace=MISD"] + ALTERNATIVES["H3"]:
        with po.policy(pol):
            ms = _search(b"ACCGTT", b"GGACCCGTTGG", 1, rc=False)
            seen.add(tuple(m[4] for m in ms))
    assert len(seen) >= 2
    # the reference's KAT cigar_parse.rs:163-176 (Sub before Del) holds for the default and is broken by Del-first
    with po.policy("trace=MDSI"):
        ms, h = po.search(b"AAAAACCCAAAA", b"GCAAAAGGGGGGGGGGGG", 8, rc=True)
        got = po.map_pat_to_text_with_cost(h, 0, 5, 8)
        po.free_matches(h)
    ms, h = po.search(b"AAAAACCCAAAA", b"GCAAAAGGGGGGGGGGGG", 8, rc=True)
    want = po.map_pat_to_text_with_cost(h, 0, 5, 8)
    po.free_matches(h)
    assert want == ((5, 8), (0, 2), 2) and got != want


def test_h3_every_class_is_distinguishable_and_spellings_of_a_class_agree():
    """18 classes: for every pair some input tells them apart (the checker's scalar traceback, trace_match), and the two spellings
    of a class (adjacent M / S swapped) never differ.  Inputs: short patterns against texts with a repeated character around one edit,
    where several alignments of the same cost exist."""
    import random

    rng = random.Random(5)
    cases = []
    for _ in range(400):
        m = rng.randint(4, 9)
        pat = bytes(rng.choice(b"AC") for _ in range(m))
        text = bytearray(pat)
        for _ in range(rng.randint(1, 3)):
            p = rng.randrange(len(text) + 1)
            r = rng.random()
            if r < 0.4 and len(text) > 2: del text[min(p, len(text) - 1)]
            elif r < 0.8: text.insert(p, rng.choice(b"AC"))
            else: text[min(p, len(text) - 1)] = rng.choice(b"AC")
        cases.append((pat, bytes(rng.choice(b"AC") for _ in range(2)) + bytes(text) + bytes(rng.choice(b"AC") for _ in range(2)), 3))
    sig = {}
    for order in ["".join(p) for p in itertools.permutations("MSID")]:
        with po.policy("trace=" + order):
            sig[order] = tuple(tuple(m[4] for m in _search(pat, text, k, rc=False)) for pat, text, k in cases)
    canon = lambda o: o.replace("SM", "MS")
    for a, b in itertools.combinations(sig, 2):
        assert (sig[a] == sig[b]) == (canon(a) == canon(b)), (a, b)
    assert len({canon(o) for o in sig}) == 18 and sorted({canon(o) for o in sig}) == sorted(TRACE_CLASSES)


def test_h5_pattern_indices_of_rc_matches():
    """get_matching_region (cigar_parse.rs:71-82) on an Rc match: with rcpath=mirror the region (start, end) selects the rows
    m-1-end .. m-1-start of the flank, i.e. another stretch of the text, unless the region is symmetric."""
    pat = b"AACCGGTTACGTTTGCA"          # 17 nt, region rows 2..6 (asymmetric: mirrored it is rows 10..14)
    rc = bytes({65: 84, 67: 71, 71: 67, 84: 65}[c] for c in pat[::-1])
    text = b"GGGGGGGG" + rc + b"GGGGGGGG"
    spans = {}
    for pol in ("rcpath=fwd", "rcpath=mirror"):
        with po.policy(pol):
            ms, h = po.search(pat, text, 0, rc=True)
            k = [i for i, m in enumerate(ms) if m.strand == 1]
            assert len(k) == 1
            spans[pol] = po.get_matching_region(h, k[0], 2, 6)
            sym = po.get_matching_region(h, k[0], 5, 11)   # rows 5..11 of 17 are their own mirror image
            fwd = po.get_matching_region(h, [i for i, m in enumerate(ms) if m.strand == 0][0], 2, 6) if any(m.strand == 0 for m in ms) else None
            po.free_matches(h)
            spans[pol + ":sym"] = sym
            assert fwd is None
    # the rc occurrence starts at text 8; flank row r sits at text 8 + 16 - r
    assert spans["rcpath=fwd"] == (8 + 16 - 6, 8 + 16 - 2) and spans["rcpath=mirror"] == (8 + 2, 8 + 6)
    assert spans["rcpath=fwd:sym"] == spans["rcpath=mirror:sym"]


def test_h4_overhang_rounding():
    # a pattern whose last three characters hang over the text end: cost round(alpha * 3)
    costs = {}
    for pol in ("ovh=floor", "ovh=ceil", "ovh=near"):
        with po.policy(pol):
            ms = _search(b"ACGTACGT", b"GGGGGGACGTA", 8, alpha=0.5, rc=False)
            costs[pol] = min(m[2] for m in ms if m[1] == 11)
    assert costs == {"ovh=floor": 1, "ovh=ceil": 2, "ovh=near": 2}
    # f32 vs f64 product: 10 * 0.7f is 7.0 in f32 and 6.99999988 in f64
    with po.policy("ovh=floor"):
        a = _search(b"A" * 10 + b"CGCG", b"CGCG" + b"T" * 8, 7, alpha=0.7, rc=False)
    with po.policy("ovh=floor:f64"):
        b = _search(b"A" * 10 + b"CGCG", b"CGCG" + b"T" * 8, 7, alpha=0.7, rc=False)
    assert a != b


def _one_read(groups, read, **kw):
    return po.Oracle([g.as_tuple() for g in groups], **kw).annotate_reads([read])


def test_h7_and_h8_change_rows_on_some_read():
    """`tie` and the Lodhi variants act inside collect_candidates / score_and_push_result (searcher.rs:294-300, 364-396).  On
    clean synthetic reads none of them changes a single row (the decisions are far from the thresholds — real data is
    where they matter); with 10 % substitutions each changes a few rows and leaves most alone."""
    from tests.common import noisy_reads

    groups, bases, offsets = noisy_reads("nbd96", 99, 1500, 200, 900, 0.10)
    base = po.Oracle([g.as_tuple() for g in groups]).annotate(bases, offsets, n_threads=NT)
    assert base.tobytes() == po.Oracle([g.as_tuple() for g in groups], policy=_abi.POLICY_DEFAULT).annotate(bases, offsets, n_threads=NT).tobytes()
    for pol in ALTERNATIVES["H7"] + ALTERNATIVES["H8"]:
        rows = po.Oracle([g.as_tuple() for g in groups], policy=pol).annotate(bases, offsets, n_threads=NT)
        assert len(rows) == len(base), pol
        assert 0 < sum(a.tobytes() != b.tobytes() for a, b in zip(rows, base)) < len(base) // 5, pol


def test_h8_formula_values():
    M, S, I, D = 0, 1, 2, 3
    assert po.lodhi([M, M, M]) == 0.5 ** 3
    with po.policy("lodhi=3:0.5:2211"):
        assert po.lodhi([M, M, M]) == 0.5 ** 6                      # span in the pattern + span in the text
        assert po.lodhi([M, I, M, M]) == 0.5 ** 7                   # an inserted text character adds 1, not 2
    with po.policy("lodhi=3:0.5:1110"):
        assert po.lodhi([M, D, M, M]) == po.lodhi([M, M, M])         # deleted pattern characters do not stretch the span
    with po.policy("lodhi=2:0.5:1111"):
        assert po.lodhi([M, M]) == 0.25 and po.lodhi([M, S, M]) == 0.125
    with po.policy("lodhi=3:0.7:1111"):
        assert abs(po.lodhi([M, M, M]) - 0.7 ** 3) < 1e-15


def test_reference_kats_hold_under_policies_that_do_not_touch_them():
    """cigar_parse.rs:104-176: no plateau ambiguity is pinned, no overhang (new_rc()), one pattern -> every H1 / H2 / H4 / H7 / H8
    alternative must reproduce all five."""
    from tests.test_oracle_kat import KATS  # (pattern, text, k, region, expected sub-cost / spans)

    for pol in ALTERNATIVES["H2"] + ALTERNATIVES["H4"] + ALTERNATIVES["H5"] + ALTERNATIVES["H7"] + ALTERNATIVES["H8"] + ["lm=left"]:
        with po.policy(pol):
            for pat, text, k, want in KATS:
                ms, h = po.search(pat, text, k, rc=True)
                got = po.map_pat_to_text_with_cost(h, 0, 5, 8)
                po.free_matches(h)
                assert got is not None and got[2] == want[2] and (want[1] is None or got[1] == want[1]), (pol, text)


# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("pol", GPU_POLICIES)
def test_hip_equals_checker_under_policy(pol):
    from barbell_amd import annotate as A
    from tests.test_gpu_parity import assert_same

    from tests.common import noisy_reads

    for cfg, n, lmin, lmax, rate in (("nbd96", 1200, 150, 2500, 0.08), ("dual", 500, 300, 2500, 0.06), ("nbd96", 400, 1, 300, 0.0)):
        groups, bases, offsets = noisy_reads(cfg, 777, n, lmin, lmax, rate)
        dm = A.Demuxer(policy=pol)
        for g in groups:
            dm.add_query_group(g)
        got = dm.demux_packed(bases, offsets)
        want = po.Oracle([g.as_tuple() for g in groups], policy=pol).annotate(bases, offsets, n_threads=NT)
        assert len(want) > n // 4
        assert_same(got, want)
        dm.close()


# k_barcode_lane's bound by row classes (lane_rows4: one of eS, eD is 0), with and without the NM masks of large flank budgets (rbk96x: k = 20)
@pytest.mark.gpu
@pytest.mark.parametrize("pol", ["lodhi=3:0.5:1011", "lodhi=3:0.5:2012,trace=DSIM", "lodhi=3:0.5:1110,tie=last,lm=left"])
def test_hip_equals_checker_row_class_bound(pol):
    from barbell_amd import annotate as A
    from tests.test_gpu_parity import assert_same

    from tests.common import noisy_reads

    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    from policy_sensitivity import mutate   # substitutions, insertions and deletions over the whole read: all three row classes occur

    for cfg, n, lmin, lmax, rate in (("nbd96", 1500, 150, 2500, 0.05), ("rbk96x", 600, 300, 2500, 0.04), ("dual", 400, 300, 2000, 0.03)):
        groups, bases, offsets = noisy_reads(cfg, 4242, n, lmin, lmax, 0.0)
        bases, offsets = mutate(bases, offsets, rate, 0.03, 0.04, 99)
        dm = A.Demuxer(policy=pol)
        for g in groups:
            dm.add_query_group(g)
        got = dm.demux_packed(bases, offsets)
        want = po.Oracle([g.as_tuple() for g in groups], policy=pol).annotate(bases, offsets, n_threads=NT)
        assert len(want) > n // 4
        assert_same(got, want)
        dm.close()


# one class of each kind of bb_prio.h (which three bits the planes are functions of: the order's last op) besides the default's
KIND_CLASSES = ["trace=MSID", "trace=MDSI", "trace=DSIM", "trace=MIDS,rcpath=mirror"]
VARIANT_KNOBS = [None, ("BARBELL_AMD_LANE", "0"), ("BARBELL_AMD_LANE", "2"), ("BARBELL_AMD_NO_FAST", "1"), ("BARBELL_AMD_FAST_MARGIN", "10"),
                 ("BARBELL_AMD_NO_PFX", "1"), ("BARBELL_AMD_GENERIC", "1"), ("BARBELL_AMD_NO_TAIL", "1")]


@pytest.mark.gpu
@pytest.mark.parametrize("knob", VARIANT_KNOBS, ids=lambda k: "default" if k is None else f"{k[0][12:]}={k[1]}")
@pytest.mark.parametrize("pol", KIND_CLASSES)
def test_every_barcode_kernel_under_other_trace_orders(monkeypatch, pol, knob):
    """The traceback order reaches every barcode kernel: k_barcode_lane and the fast k_barcode_pfx as a compile-time class
    (bb_tu_class.hip), the exact k_barcode_pfx, the prefix kernels, k_barcode_reg and k_barcode at run time — each forced in turn, on
    every BASELINE query set (rbk96x: k = 20, the 64-column kernels, the checkpointed flank traceback), noisy reads."""
    from barbell_amd import annotate as A
    from tests.test_gpu_parity import assert_same
    from tests.common import noisy_reads

    if knob:
        monkeypatch.setenv(*knob)
    for cfg, n, rate in (("nbd96", 900, 0.08), ("dual", 400, 0.06), ("rbk96x", 200, 0.04), ("rbk24", 300, 0.05)):
        groups, bases, offsets = noisy_reads(cfg, 4321, n, 200, 2500, rate)
        dm = A.Demuxer(policy=pol)
        for g in groups:
            dm.add_query_group(g)
        got = dm.demux_packed(bases, offsets)
        want = po.Oracle([g.as_tuple() for g in groups], policy=pol).annotate(bases, offsets, n_threads=NT)
        assert len(want) > n // 4
        assert_same(got, want)
        dm.close()


@pytest.mark.gpu
@pytest.mark.parametrize("pol", ["lodhi=3:0.5:1011", "lodhi=3:0.5:1101", "lodhi=3:0.5:0111", "lodhi=3:0.5:3302,trace=MDIS"])
def test_fast_path_with_zero_and_unequal_decay_exponents(pol):
    """The fast path's bound is built from the policy's exponents (lodhi_bound_table_entry: eM on Match columns, min(eS, eI) on the other
    text columns) and stays an upper bound when an exponent is 0 or the two differ — looser then (more hits go on to the exact kernel), never
    wrong.  Not among the bench's policy variants: no reading of a Lodhi kernel makes a substituted column free and an inserted one not."""
    from barbell_amd import annotate as A
    from tests.test_gpu_parity import assert_same
    from tests.common import noisy_reads

    for cfg, n, rate in (("nbd96", 800, 0.08), ("rbk96x", 150, 0.04)):
        groups, bases, offsets = noisy_reads(cfg, 99, n, 300, 2500, rate)
        dm = A.Demuxer(policy=pol)
        for g in groups:
            dm.add_query_group(g)
        got = dm.demux_packed(bases, offsets)
        want = po.Oracle([g.as_tuple() for g in groups], policy=pol).annotate(bases, offsets, n_threads=NT)
        assert len(want) > n // 4
        assert_same(got, want)
        dm.close()


@pytest.mark.gpu
def test_trace_orders_in_and_out_of_the_build():
    """bb_last_barcode_stats: under every traceback class THIS BUILD holds (bb_build_trace_classes: by default the five classes the
    reference's own KATs leave open, tests/golden/policy_feasible.json; all 18 with `make CLASSES=all`) the SQK-NBD114-96 hits are decided
    by k_barcode_lane.  A class outside the build gives the same rows as the checker through the kernels that read the order at run time,
    and the context says so (bb_last_error(ctx))."""
    import sys

    from barbell_amd import annotate as A
    from tests.common import policy_feasible

    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import policy_feasible as pf

    order = pf.build_class_order()
    built = [order[i] for i in A.build_trace_classes()]
    assert "MISD" in built and set(policy_feasible()["feasible"]["trace"]) <= set(built)    # every feasible class is fast in every build
    groups = config_groups("nbd96")
    bases, offsets = A.synth_reads_host(groups, 17, 1000, 3000, 0, 2000)
    for cls in TRACE_CLASSES:
        dm = A.Demuxer(policy="trace=" + cls)
        for g in groups:
            dm.add_query_group(g)
        got = dm.demux_packed(bases, offsets)
        st = [dm.barcode_stats(0, s) for s in (0, 1)]
        if cls in built:
            assert all(s["lane_kernel"] for s in st) and sum(s["hits"] for s in st) > 1000, (cls, st)
            assert dm.note() == ""
        else:
            assert "not in this build" in dm.note() and "trace=" + cls in dm.note()
            want = po.Oracle([g.as_tuple() for g in groups], policy="trace=" + cls).annotate(bases, offsets, n_threads=NT, fast=True)
            assert got.tobytes() == want.tobytes(), cls
        dm.close()


@pytest.mark.gpu
def test_policy_through_the_environment_and_getter(monkeypatch):
    import ctypes as C

    from barbell_amd import annotate as A
    from barbell_amd._lib import lib

    monkeypatch.setenv("BARBELL_AMD_POLICY", "lm=left,lodhi=3:0.5:2211")
    groups = config_groups("nbd96")
    dm = A.Demuxer()
    for g in groups:
        dm.add_query_group(g)
    p = _abi.Policy()
    assert lib().bb_get_policy(dm._ctx(), C.byref(p)) == 0
    assert _abi.policy_to_str(p) == "lm=left,rc=scan,trace=MISD,ovh=floor,tie=first,lodhi=3:0.5:2211,rcpath=fwd"
    bases, offsets = A.synth_reads_host(groups, 5, 300, 1500, 0, 300)
    got = dm.demux_packed(bases, offsets)
    want = po.Oracle([g.as_tuple() for g in groups], policy="lm=left,lodhi=3:0.5:2211").annotate(bases, offsets, n_threads=NT)
    assert got.tobytes() == want.tobytes()
    monkeypatch.setenv("BARBELL_AMD_POLICY", "lm=sideways")
    with pytest.raises(A.BarbellError):
        A.Demuxer().add_query_group(groups[0])._ctx()
