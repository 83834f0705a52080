#!/usr/bin/env python3
"""dev aid (GPU box): randomised soak of the annotate path against the oracle — kit / custom query sets with random flank
error budgets and overhang factors, read mixtures (synthetic constructs, constructs cut at either end, homopolymers,
repeats of flank pieces, tiny reads), random scan variants (BARBELL_AMD_SCAN_FILTER / _WIDE / _ENDS).
usage: soak.py FIRST_SEED N_SEEDS   (BARBELL_AMD_SEG_LINES_FIXED=1: keep the caller's BARBELL_AMD_SEG_LINES instead of drawing one per seed)"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from barbell_amd import _abi, annotate as A, kits  # noqa: E402
from tests.common import EX  # noqa: E402
from tests.test_gpu_parity import run_both  # noqa: E402

import itertools  # noqa: E402

TRACE_ORDERS = ["MISD"] * 6 + ["".join(p) for p in itertools.permutations("MSID")]   # every order, the default a fifth of the time
first, count = int(sys.argv[1]), int(sys.argv[2])
bad = 0
for seed in range(first, first + count):
    rng = np.random.default_rng(seed)
    cfg = rng.choice(["nbd96", "nbd24", "dual", "rbk24", "left", "rbk96x", "mixed", "mixed"])   # round 5: contexts of several groups of different widths and scan kinds (one launch per kind: bb_launch_scans)
    k = None if rng.random() < 0.2 else int(rng.integers(0, 9))
    if cfg == "nbd96": groups = kits.groups_from_kit("SQK-NBD114-96", flank_max_errors=k)
    elif cfg == "nbd24": groups = kits.groups_from_kit("SQK-NBD114-24", flank_max_errors=k)
    elif cfg == "rbk24": groups = kits.groups_from_kit("SQK-RBK114-24", flank_max_errors=k if k is None or rng.random() < 0.7 else None)
    elif cfg == "rbk96x": groups = kits.groups_from_kit("SQK-RBK114-96", use_extended=True, flank_max_errors=k if rng.random() < 0.5 else None)
    elif cfg == "mixed":   # two to five groups drawn from different kits and the examples: widths 2 and 3, filtered and full scans, narrow and wide windows side by side
        pool = [lambda: kits.groups_from_kit("SQK-NBD114-24", flank_max_errors=k)[0], lambda: kits.groups_from_kit("SQK-RBK114-24", flank_max_errors=None if rng.random() < 0.5 else k)[0],
                lambda: kits.group_from_fasta(os.path.join(EX, "native_left.fasta"), _abi.BB_FTAG, k), lambda: kits.group_from_fasta(os.path.join(EX, "native_right.fasta"), _abi.BB_RTAG, k),
                lambda: kits.groups_from_kit("SQK-PCB114-24", flank_max_errors=k)[0], lambda: kits.groups_from_kit("SQK-16S114-24", flank_max_errors=k)[-1]]
        groups = [pool[int(j)]() for j in rng.permutation(len(pool))[: int(rng.integers(2, 6))]]
    elif cfg == "left": groups = [kits.group_from_fasta(os.path.join(EX, "native_left.fasta"), _abi.BB_FTAG, k)]
    else: groups = [kits.group_from_fasta(os.path.join(EX, "native_left.fasta"), _abi.BB_FTAG, k),
                    kits.group_from_fasta(os.path.join(EX, "native_right.fasta"), _abi.BB_RTAG, k)]
    mode = rng.choice(["", "1", "wide", "ends", "0"])
    for v in ("BARBELL_AMD_SCAN_FILTER", "BARBELL_AMD_FILTER_WIDE", "BARBELL_AMD_FILTER_ENDS"):
        os.environ.pop(v, None)
    if mode in ("1", "wide", "ends"): os.environ["BARBELL_AMD_SCAN_FILTER"] = "1"
    if mode == "0": os.environ["BARBELL_AMD_SCAN_FILTER"] = "0"
    if mode == "wide": os.environ["BARBELL_AMD_FILTER_WIDE"] = "1"
    if mode == "ends": os.environ["BARBELL_AMD_FILTER_ENDS"] = "1"
    alpha = float(rng.choice([0.0, 0.3, 0.5, 0.7, 1.0]))
    # a random policy (include/barbell_amd_policy.h) for a third of the seeds, a forced per-batch scan choice for some
    policy = None
    if rng.random() < 0.35:
        policy = ",".join([f"lm={rng.choice(['right', 'left', 'strict'])}", f"rc={rng.choice(['scan', 'fwd'])}",
                           f"trace={rng.choice(TRACE_ORDERS)}", f"rcpath={rng.choice(['fwd', 'fwd', 'mirror'])}",
                           f"ovh={rng.choice(['floor', 'ceil', 'near', 'floor:f64', 'near:f64'])}", f"tie={rng.choice(['first', 'last'])}",
                           f"lodhi={rng.choice(['3:0.5:1111', '3:0.5:1111', '3:0.5:2211', '3:0.5:1121', '3:0.5:1110', '3:0.5:2131', '3:0.5:1011', '2:0.5:1111', '3:0.7:1111', '4:0.5:1212'])}"])
    # round 5: the scans' work items — segments of 512 bytes .. 2 KB (nearly every read cut, the full scan's valley rule at every cut), the
    # default (reads above 8 KB cut), or whole reads in file order; a setting given from outside (a forced soak) stays
    if "BARBELL_AMD_SEG_LINES_FIXED" not in os.environ:
        os.environ.pop("BARBELL_AMD_SEG_LINES", None)
        sl = rng.choice(["", "", "4", "8", "16", "0"])
        if sl: os.environ["BARBELL_AMD_SEG_LINES"] = str(sl)
    # round 6: how a batch is treated — as the library treats a small one by default (deferred, one lane per (hit, barcode)), deferred with one lane
    # per hit, or the classic way (a round trip per decision), as every 2 M-read step of the benchmark runs
    for v in ("BARBELL_AMD_DEFER_MAX", "BARBELL_AMD_SMALL_PFX_MAX"): os.environ.pop(v, None)
    treat = int(rng.integers(0, 3))
    if treat >= 1: os.environ["BARBELL_AMD_SMALL_PFX_MAX"] = "0"
    if treat == 2: os.environ["BARBELL_AMD_DEFER_MAX"] = "0"
    os.environ.pop("BARBELL_AMD_ADAPT_FRAC", None)
    if rng.random() < 0.3: os.environ["BARBELL_AMD_ADAPT_FRAC"] = str(rng.choice(["0", "1", "0.01"]))
    noise = float(rng.choice([0.0, 0.0, 0.03, 0.08]))
    n = int(rng.integers(100, 500))
    lmax = int(rng.integers(40, 3000))
    b1, o1 = A.synth_reads_host(groups, seed, max(1, lmax // 8), lmax, 0, n)
    reads = [b1[int(o1[i]):int(o1[i + 1])].tobytes() for i in range(n)]
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    for i in range(n // 2):
        g = groups[int(rng.integers(0, len(groups)))]
        q = bytes(g.seqs[int(rng.integers(0, len(g.seqs)))])
        body = bytes(rng.choice(acgt, int(rng.integers(0, 600))))
        kind = int(rng.integers(0, 7))
        if kind == 0: r = q[int(rng.integers(0, len(q))):] + body
        elif kind == 1: r = body + q[: len(q) - int(rng.integers(0, len(q)))]
        elif kind == 2: r = bytes([b"ACGT"[int(rng.integers(0, 4))]]) * int(rng.integers(1, 400)) + body[:50]
        elif kind == 3: r = q[: int(rng.integers(4, 20))] * int(rng.integers(1, 30)) + body
        elif kind == 4: r = body[:100] + q + body[100:] + q[::-1]
        elif kind == 5: r = q[int(rng.integers(0, len(q))): len(q) - int(rng.integers(0, 10))]
        else: r = body[: int(rng.integers(0, 30))]
        reads.append(r)
    order = rng.permutation(len(reads))
    reads = [reads[i] for i in order]
    bases = np.frombuffer(b"".join(reads), dtype=np.uint8).copy() if any(reads) else np.zeros(0, np.uint8)
    if noise and len(bases):
        pos = rng.random(len(bases)) < noise
        bases[pos] = rng.choice(acgt, int(pos.sum()))
    offsets = np.concatenate([[0], np.cumsum([len(r) for r in reads])]).astype(np.uint64)
    try:
        kw = dict(alpha=alpha) if policy is None else dict(alpha=alpha, policy=policy)
        dm_, got, want = run_both(groups, bases, offsets, **kw)
        if seed % 4 == 0 and got.tobytes() == want.tobytes():   # the two-bases-per-byte form of the boundary on the same reads
            got = dm_.demux_nibbles(bases, offsets)
        dm_.close()
    except A.BarbellError as e:
        print(f"seed {seed} {cfg} k={k} alpha={alpha} mode='{mode}' policy={policy}: {e}")
        if e.code != _abi.BB_E_UNSUPPORTED: bad += 1
        continue
    ok = got.tobytes() == want.tobytes()
    if not ok:
        bad += 1
        print(f"MISMATCH seed {seed} {cfg} k={k} alpha={alpha} mode='{mode}' policy={policy} adapt={os.environ.get('BARBELL_AMD_ADAPT_FRAC')} seg={os.environ.get('BARBELL_AMD_SEG_LINES')} noise={noise} rows {len(got)} vs {len(want)}")
print(f"{count} seeds, {bad} bad")
sys.exit(1 if bad else 0)
