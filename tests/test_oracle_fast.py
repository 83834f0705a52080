"""The bit-parallel timing path of the CPU checker (bbo_annotate_batch_fast: 64-bit Myers / Hyyro words, what bench.py's
cpu_baseline reports) gives exactly the rows of the scalar restatement (bbo_annotate_batch, what every parity test compares the
GPU with) — every BASELINE query set, ragged and tiny reads, overhang settings, noisy reads, and the policies it honours."""
import os
import time

import numpy as np
import pytest

from oracle import pyoracle as po
from barbell_amd.parallel import effective_cpus  # noqa: E402
from tests.common import config_groups, noisy_reads

NT = effective_cpus()


@pytest.mark.parametrize("cfg,n,lmin,lmax,rate", [("nbd96", 500, 1, 1200, 0.0), ("nbd96", 400, 200, 900, 0.08), ("dual", 300, 50, 1500, 0.05),
                                                  ("rbk24", 200, 100, 2000, 0.0), ("rbk96x", 120, 300, 1500, 0.03)])
def test_fast_equals_scalar(cfg, n, lmin, lmax, rate):
    groups, bases, offsets = noisy_reads(cfg, 4242, n, lmin, lmax, rate)
    o = po.Oracle([g.as_tuple() for g in groups])
    a = o.annotate(bases, offsets, n_threads=NT)
    b = o.annotate(bases, offsets, n_threads=NT, fast=True)
    assert len(a) > n // 4 and a.tobytes() == b.tobytes()


@pytest.mark.parametrize("kw", [dict(alpha=0.0), dict(alpha=1.0), dict(policy="lm=left,tie=last,ovh=ceil,rc=fwd,lodhi=3:0.5:2211"),
                                dict(policy="lm=strict,ovh=near:f64"), dict(policy="trace=MSID"),
                                dict(policy="trace=DSIM"), dict(policy="trace=MIDS,rcpath=mirror"), dict(policy="trace=IDMS,lodhi=3:0.5:2131"), dict(policy="trace=MDSI")])
def test_fast_equals_scalar_variants(kw):
    groups, bases, offsets = noisy_reads("nbd96", 7, 400, 1, 700, 0.06)
    o = po.Oracle([g.as_tuple() for g in groups], **kw)
    assert o.annotate(bases, offsets, n_threads=NT).tobytes() == o.annotate(bases, offsets, n_threads=NT, fast=True).tobytes()


def test_fast_is_faster():
    groups, bases, offsets = noisy_reads("nbd96", 11, 300, 3000, 4000, 0.0)
    o = po.Oracle([g.as_tuple() for g in groups])
    t0 = time.perf_counter(); a = o.annotate(bases, offsets, n_threads=1); t1 = time.perf_counter()
    b = o.annotate(bases, offsets, n_threads=1, fast=True); t2 = time.perf_counter()
    assert a.tobytes() == b.tobytes() and (t2 - t1) * 3 < (t1 - t0)


@pytest.mark.parametrize("cfg,n,lmin,lmax,rate", [("nbd96", 160, 900, 6000, 0.0), ("nbd96", 120, 2000, 5000, 0.06), ("dual", 80, 3000, 4000, 0.04),
                                                  ("rbk96x", 50, 1800, 4000, 0.03), ("rbk24", 80, 1500, 4000, 0.05)])
@pytest.mark.parametrize("policy", [None, "lm=left", "lm=strict,tie=last", "trace=MSID,ovh=ceil,rc=fwd", "lodhi=3:0.5:2211", "lodhi=4:0.7:1110"])
def test_vector_forms_equal_scalar_on_long_reads(cfg, n, lmin, lmax, rate, policy):
    """The AVX-512 forms of the timing path (oracle/bb_oracle_simd.h: text-parallel flank scan with valleys replayed, pattern-parallel barcode
    pass, eight Lodhi scores per vector) take reads of at least 16 (m + k) bases: rows equal to the scalar restatement's on reads of
    1-6 kb, under every local-minimum rule, tie rule, traceback order, overhang rounding and Lodhi setting."""
    groups, bases, offsets = noisy_reads(cfg, 99, n, lmin, lmax, rate)
    kw = {"policy": policy} if policy else {}
    orc = po.Oracle([g.as_tuple() for g in groups], **kw)
    want = orc.annotate(bases, offsets, n_threads=NT)
    assert len(want) > n // 3
    got = orc.annotate(bases, offsets, n_threads=NT, fast=True)
    assert got.tobytes() == want.tobytes()


def test_flank_matches_at_the_reads_ends_and_in_runs():
    """What the replay of valleys must get right: matches that end in the overhang columns behind the read, matches at column 1, valleys
    hundreds of columns long (a low-complexity flank on a repeat), valleys a few columns apart, reads barely long enough for the vector scan."""
    from barbell_amd import _abi, kits

    groups = config_groups("nbd96")
    flank = bytes(groups[0].seqs[7])
    rng = np.random.default_rng(3)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    comp = bytes.maketrans(b"ACGT", b"TGCA")

    def body(k):
        return acgt[rng.integers(0, 4, k)].tobytes()

    reads = []
    for i in range(60):
        r = body(int(rng.integers(800, 3000)))
        kind = i % 6
        if kind == 0:
            r = flank[5:] + r                                   # truncated construct at column 0 (left overhang)
        elif kind == 1:
            r = r + flank[:-9]                                  # ... hanging off the end (right overhang columns)
        elif kind == 2:
            r = r + flank.translate(comp)[::-1][:30]            # rc construct cut by the read's end
        elif kind == 3:
            r = flank + flank[:20] + flank + r[:900] + flank    # constructs next to each other: valleys a few columns apart
        elif kind == 4:
            r = r[:800]                                         # just above 16 (m + k)
        else:
            r = r[:400] + flank + r[400:]
        reads.append(r)
    # a flank of (AC)n against (AC)n text: one valley over the whole repeat, a local minimum every other column
    lowc = [kits.QueryGroup([b"ACACACACACACACAC" + bc + b"ACACACAC" for bc in (b"GGTTGGTT", b"TTGGTTGG", b"GTGTGTGT")], ["a", "b", "c"], 0, 2)]
    rep = [body(300) + b"AC" * 700 + body(300) for _ in range(6)]
    for gs, rs in ((groups, reads), (lowc, rep)):
        bases, offsets = _abi.pack_reads(rs)
        for pol in (None, "lm=left", "lm=strict"):
            orc = po.Oracle([g.as_tuple() for g in gs], **({"policy": pol} if pol else {}))
            want = orc.annotate(bases, offsets, n_threads=NT)
            assert len(want) > 5
            assert orc.annotate(bases, offsets, n_threads=NT, fast=True).tobytes() == want.tobytes(), pol
