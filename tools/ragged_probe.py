#!/usr/bin/env python3
"""dev aid (GPU box): does a heavy-tailed read-length distribution (real nanopore runs: most reads a few hundred to a few thousand nt, a
tail to 100 kb) cost throughput per BASE against the benchmark's fixed 4 kb reads?  The scans that give a lane one (read, strand) run a wave
at the pace of its longest read.  Prints bases/s and the stage times for both shapes.  usage: ragged_probe.py [cfg ...] [--reads N]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from barbell_amd import annotate as A  # noqa: E402
from tests.common import config_groups, heavy_tailed_batch as mixed  # noqa: E402


def run(groups, bases, offs, label, order=None):
    dev = torch.device("cuda:0")
    if order is not None:
        lens = (offs[1:] - offs[:-1]).astype(np.int64)
        idx = order(lens)
        no = np.zeros_like(offs); no[1:] = np.cumsum(lens[idx])
        nb = np.empty_like(bases)
        for j, i in enumerate(idx):
            nb[int(no[j]):int(no[j + 1])] = bases[int(offs[i]):int(offs[i + 1])]
        bases, offs = nb, no
    dm = A.Demuxer(device=0)
    for g in groups:
        dm.add_query_group(g)
    n = len(offs) - 1
    d_b = torch.from_numpy(bases).to(dev)
    d_o = torch.from_numpy(offs.astype(np.int64)).to(dev)
    d_rows = torch.empty(8 * n * 48, dtype=torch.uint8, device=dev)
    for _ in range(2):
        dm.demux_dev(d_b.data_ptr(), d_o.data_ptr(), n, d_rows.data_ptr(), 8 * n)
    dm.set_timing(True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    K = 4
    kms = {}
    for _ in range(K):
        rows = dm.demux_dev(d_b.data_ptr(), d_o.data_ptr(), n, d_rows.data_ptr(), 8 * n)
        for k, v in dm.kernel_ms().items():
            kms[k] = kms.get(k, 0.0) + v / K
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / K
    print(f"  {label:28s} reads {n} bases {len(bases)/1e9:.2f} G  {dt*1e3:8.2f} ms  {len(bases)/dt/1e9:7.1f} Gbases/s  {n/dt/1e6:6.1f} M reads/s  rows {rows}  " +
          " ".join(f"{k}={v:.2f}" for k, v in kms.items() if v > 0.05), flush=True)
    return dt


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    n = int(sys.argv[sys.argv.index("--reads") + 1]) if "--reads" in sys.argv else 300000
    if "--short" in sys.argv:   # many short reads (amplicons): what a read costs apart from its bases
        for cfg in (args or ["nbd96"]):
            groups = config_groups(cfg)
            print(cfg, flush=True)
            for lo, hi, k in ((4000, 4000, 300000), (300, 600, 2000000), (150, 250, 4000000)):
                b, o = A.synth_reads_host(groups, 13, lo, hi, 0, k)
                run(groups, b, o, f"{k} reads of {lo}..{hi} nt")
        sys.exit(0)
    for cfg in (args or ["nbd96", "dual", "rbk96x"]):
        groups = config_groups(cfg)
        print(cfg, flush=True)
        b, o = mixed(groups, n)
        mean = int(len(b) / (len(o) - 1))
        fb, fo = A.synth_reads_host(groups, 9, mean, mean, 0, len(o) - 1)
        run(groups, fb, fo, f"fixed {mean} nt")
        run(groups, b, o, "heavy-tailed, file order")
        run(groups, b, o, "heavy-tailed, sorted by length", order=lambda lens: np.argsort(lens, kind="stable"))
