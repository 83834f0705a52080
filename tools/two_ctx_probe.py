#!/usr/bin/env python3
"""dev aid (GPU box): one 2 M-read step as ONE call on one context against the same batch cut into C parts on C contexts of the library,
called from C host threads at once (what the C++ host does with --streams C): the latency-bound kernels of one part (verification, flank
traceback, lists, collapse, emit) run under the VALU-bound kernels of another.  usage: two_ctx_probe.py [contexts ...]
  two_ctx_probe.py stagger FRAC DELAY_MS   two contexts, the first takes FRAC of the batch, the second starts DELAY_MS later"""
import sys, os, time, threading
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from barbell_amd import annotate as A
from tests.common import config_groups

L, n, steps = 4000, 2_000_000, 10
dev = torch.device("cuda", 0)
groups = config_groups("nbd96")
d_off = torch.arange(0, n + 1, dtype=torch.int64, device=dev) * L
d_bases = torch.empty(n * L, dtype=torch.uint8, device=dev)
dm0 = A.Demuxer()
for g in groups: dm0.add_query_group(g)
dm0.synth_dev(0xBA7BE11 ^ 2, L, L, 0, n, d_off.data_ptr(), d_bases.data_ptr())
if len(sys.argv) > 1 and sys.argv[1] == "stagger":
    frac, delay = float(sys.argv[2]), float(sys.argv[3]) * 1e-3
    dms = []
    for _ in range(2):
        dm = A.Demuxer()
        for g in groups: dm.add_query_group(g)
        dms.append(dm)
    n0 = int(n * frac); parts = [(0, n0), (n0, n - n0)]
    rows = [torch.empty(4 * p[1] * 48, dtype=torch.uint8, device=dev) for p in parts]
    out = [0, 0]
    def work(i):
        if i == 1 and delay > 0:
            t = time.perf_counter() + delay
            while time.perf_counter() < t: pass
        s0, cnt = parts[i]
        out[i] = dms[i].demux_dev(d_bases.data_ptr(), d_off.data_ptr() + 8 * s0, cnt, rows[i].data_ptr(), 4 * cnt)
    def step():
        ts = [threading.Thread(target=work, args=(i,)) for i in range(2)]
        for t in ts: t.start()
        for t in ts: t.join()
    step(); step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps): step()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / steps
    print(f"stagger frac {frac} delay {delay * 1e3:.1f} ms: {dt * 1e3:.2f} ms per 2 M-read step, {n / dt / 1e6:.1f} M reads/s, rows {sum(out)}", flush=True)
    sys.exit(0)
for C in [int(x) for x in sys.argv[1:]] or [1, 2, 3, 4]:
    dms = []
    for _ in range(C):
        dm = A.Demuxer()
        for g in groups: dm.add_query_group(g)
        dms.append(dm)
    part = n // C
    offs = d_off[: part + 1].contiguous()
    rows = [torch.empty(4 * part * 48, dtype=torch.uint8, device=dev) for _ in range(C)]
    out = [0] * C
    def work(i):
        out[i] = dms[i].demux_dev(d_bases.data_ptr() + i * part * L, offs.data_ptr(), part, rows[i].data_ptr(), 4 * part)
    def step():
        ts = [threading.Thread(target=work, args=(i,)) for i in range(C)]
        for t in ts: t.start()
        for t in ts: t.join()
    step(); step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps): step()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / steps
    print(f"contexts {C}: {dt * 1e3:.2f} ms per 2 M-read step, {n / dt / 1e6:.1f} M reads/s, rows {sum(out)}", flush=True)
    for dm in dms: dm.close()
