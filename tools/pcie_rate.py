#!/usr/bin/env python3
"""Measures the PCIe-inclusive rate of the host-pointer boundary (bb_annotate_batch: pageable host
buffers -> H2D -> pipeline -> rows D2H).  Reported in DESIGN.md only; never bench.py's `value`."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from barbell_amd import annotate as A
from tests.common import config_groups

n, L = int(sys.argv[1]) if len(sys.argv) > 1 else 500_000, 4000
groups = config_groups("nbd96")
bases, offsets = A.synth_reads_host(groups, 0xBA7BE11 ^ 2, L, L, 0, n)
dm = A.Demuxer()
for g in groups:
    dm.add_query_group(g)
dm.demux_packed(bases[: 1000 * L], offsets[:1001])  # warm up
def run(tag, b):
    for _ in range(3):
        t = time.perf_counter()
        rows = dm.demux_packed(b, offsets)
        dt = time.perf_counter() - t
        print(f"{tag}: {n} reads, {len(rows)} rows: {dt*1e3:.1f} ms -> {n/dt/1e6:.2f} M reads/s PCIe-inclusive ({n*L/dt/1e9:.1f} GB/s of bases)")


run("pageable", bases)
# the same reads in page-locked memory from bb_host_malloc
import ctypes as C
from barbell_amd._lib import lib
p = C.c_void_p()
assert lib().bb_host_malloc(dm._ctx(), len(bases), C.byref(p)) == 0
pinned = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), shape=(len(bases),))
pinned[:] = bases
run("page-locked", pinned)
lib().bb_host_free(dm._ctx(), p)
