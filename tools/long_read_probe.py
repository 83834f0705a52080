#!/usr/bin/env python3
"""dev aid (GPU box): ultra-long reads (nanopore reads reach megabases) among ordinary ones — reads of 0.3 / 1 / 2.5 M nt glued from
synthetic 4 kb constructs (so flank hits lie all along them), every scan kind (filtered narrow / wide, full), against the oracle.
usage: long_read_probe.py [cfg ...]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from barbell_amd import annotate as A  # noqa: E402
from tests.common import config_groups, long_batch  # noqa: E402
from tests.test_gpu_parity import run_both, assert_same  # noqa: E402


if __name__ == "__main__":
    bad = 0
    for cfg in (sys.argv[1:] or ["nbd96", "dual", "rbk96x", "rbk24"]):
        groups = config_groups(cfg)
        b, o = long_batch(groups, 11)
        t0 = time.time()
        dm, got, want = run_both(groups, b, o)
        try:
            assert_same(got, want)
            print(cfg, "ok", len(got), "rows", "%.1f s" % (time.time() - t0), flush=True)
        except AssertionError as e:
            bad += 1
            print(cfg, "MISMATCH", str(e)[:600], flush=True)
    print("bad", bad)
