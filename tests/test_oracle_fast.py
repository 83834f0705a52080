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
