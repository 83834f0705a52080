#!/usr/bin/env python3
"""Extended differential run (not part of the test suite): GPU rows vs oracle rows on many seeds, read-length
mixes, junk characters and flank error budgets.  usage: long_fuzz.py [n_seeds] [reads_per_seed]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from barbell_amd import annotate as A, kits  # noqa: E402
from barbell_amd.parallel import effective_cpus  # noqa: E402
from oracle import pyoracle as po  # noqa: E402
from tests.common import config_groups  # noqa: E402

n_seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 40
per = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
rng = np.random.default_rng(20240928)
total = 0
for s in range(n_seeds):
    kind = s % 5
    if kind == 0:
        groups = kits.groups_from_kit("SQK-NBD114-96", flank_max_errors=int(rng.integers(1, 9)))
    elif kind == 1:
        groups = config_groups("dual")
    elif kind == 2:
        groups = kits.groups_from_kit("SQK-RBK114-24", flank_max_errors=int(rng.integers(3, 14)))
    elif kind == 3:
        groups = config_groups("nbd96")
    else:
        groups = config_groups("rbk96x")
    lo = int(rng.integers(20, 400))
    hi = lo + int(rng.integers(10, 6000))
    bases, offsets = A.synth_reads_host(groups, int(rng.integers(1, 1 << 30)), lo, hi, int(rng.integers(0, 1 << 20)), per)
    b = bases.copy()
    if s % 3 == 0:  # junk: IUPAC codes, lower case, non-letters
        idx = rng.integers(0, len(b), size=len(b) // 200)
        b[idx] = rng.choice(np.frombuffer(b"NRYKMSWacgtn-*X", dtype=np.uint8), size=len(idx))
    dm = A.Demuxer()
    for g in groups:
        dm.add_query_group(g)
    got = dm.demux_packed(b, offsets)
    want = po.Oracle([g.as_tuple() for g in groups]).annotate(b, offsets, n_threads=effective_cpus())
    ok = got.tobytes() == want.tobytes()
    total += len(got)
    print(f"seed {s} kind {kind} reads {per} len {lo}..{hi}: rows {len(got)} {'ok' if ok else 'MISMATCH'}", flush=True)
    if not ok:
        sys.exit(1)
    dm.close()
print(f"all {n_seeds} seeds identical, {total} rows")
