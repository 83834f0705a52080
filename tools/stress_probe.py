#!/usr/bin/env python3
"""dev aid (GPU box): one stress mix of bench.py's `stress` object in detail — per-kernel time, hits and undecided hits per strand.
    python tools/stress_probe.py MODE [N]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from barbell_amd import annotate as A
from tests.common import config_groups

mode = int(sys.argv[1]); n = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
L = 4000
groups = config_groups("nbd96")
dm = A.Demuxer(device=0)
for g in groups: dm.add_query_group(g)
dev = torch.device("cuda:0")
seed = (mode << 56) | (0xBA7BE11 ^ 2)
d_off = torch.arange(0, n + 1, dtype=torch.int64, device=dev) * L
d_bases = torch.empty(n * L, dtype=torch.uint8, device=dev)
torch.cuda.synchronize()
dm.synth_dev(seed, L, L, 0, n, d_off.data_ptr(), d_bases.data_ptr())
cap = 6 * n
d_rows = torch.empty(cap * 48, dtype=torch.uint8, device=dev)
dm.set_timing(True)
for it in range(3):
    nr = dm.demux_dev(d_bases.data_ptr(), d_off.data_ptr(), n, d_rows.data_ptr(), cap)
    print(it, "rows", nr, {k: round(v, 2) for k, v in dm.kernel_ms().items()}, dm.scan_stats(0), [dm.barcode_stats(0, s) for s in (0, 1)], dm.dominant_kernel())
