#!/usr/bin/env python3
"""dev aid (GPU box): replays one seed of tests/test_gpu_parity.py::test_fuzz_geometry and prints the rows of the reads on
which the GPU and the oracle differ, with the query geometry.  usage: fuzz_debug.py SEED"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from barbell_amd import annotate as A  # noqa: E402
from barbell_amd.kits import QueryGroup  # noqa: E402
from tests.test_gpu_parity import run_both  # noqa: E402

seed = int(sys.argv[1])
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
    k = int(rng.integers(0, 9)) if rng.random() < 0.7 else None
    groups.append(QueryGroup(seqs, [f"g{gi}_{i}" for i in range(n)], int(rng.integers(0, 2)), k))
    print(f"group {gi}: pre {len(pre)} barcode {blen} suf {len(suf)} n {n} k {k} type {groups[-1].match_type}")
bases, offsets = A.synth_reads_host(groups, 50 + seed, 80, 1500, 0, 300)
kw = dict(alpha=float(rng.choice([0.0, 0.4, 0.7])), min_score_frac=float(rng.choice([0.1, 0.2, 0.5])),
          min_score_diff_frac=float(rng.choice([0.0, 0.1, 0.2])))
print(kw)
_, got, want = run_both(groups, bases, offsets, **kw)
bad = sorted(set(got["read_idx"][[i for i in range(min(len(got), len(want))) if got[i].tobytes() != want[i].tobytes()]].tolist()))
print("reads that differ:", bad[:10])
for r in bad[:3]:
    print("read", r, "len", int(offsets[r + 1] - offsets[r]), "misalignment of the read's first byte", int(offsets[r]) % 128)
    print(" names:", got.dtype.names)
    for x in got[got["read_idx"] == r]:
        print("  got ", x)
    for x in want[want["read_idx"] == r]:
        print("  want", x)
