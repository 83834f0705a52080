import sys, os
sys.path.insert(0, '/root/repo')
import torch, numpy as np
from barbell_amd import annotate as A
from tests.common import config_groups
for cfg in ("rbk96x", "dual", "nbd96"):
    groups = config_groups(cfg)
    dm = A.Demuxer()
    for g in groups: dm.add_query_group(g)
    n, L = 500000, 4000
    dev = torch.device("cuda", 0)
    d_off = torch.arange(0, n + 1, dtype=torch.int64, device=dev) * L
    d_bases = torch.empty(n * L, dtype=torch.uint8, device=dev)
    dm.synth_dev(0xBA7BE11 ^ 5, L, L, 0, n, d_off.data_ptr(), d_bases.data_ptr())
    d_rows = torch.empty(8 * n * 48, dtype=torch.uint8, device=dev)
    nr = dm.demux_dev(d_bases.data_ptr(), d_off.data_ptr(), n, d_rows.data_ptr(), 8 * n)
    print(cfg, 'rows', nr, [(g, s, dm.barcode_stats(g, s)) for g in range(len(groups)) for s in (0, 1)])
    dm.close()
