#!/usr/bin/env python3
"""dev aid: a FASTQ file gzip-compressed as many members (what `cat run/*.fastq.gz > all.fastq.gz` makes) through the C++ host — the members
inflated side by side (ParallelInflater::inflate_regular) against one after the other (BARBELL_AMD_GZ_SERIAL=1).
usage: gz_members_probe.py [n_reads] [members]   (GPU box: annotate; add --stage for the host side alone, no GPU)"""
import os
import re
import subprocess
import sys
import time
import zlib
from multiprocessing import Pool

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from barbell_amd import annotate as A  # noqa: E402
from barbell_amd.parallel import effective_cpus  # noqa: E402
from tests.common import config_groups  # noqa: E402

CLI = os.path.join(ROOT, "barbell_amd", "bin", "barbell-amd")


def gz_member(text):
    c = zlib.compressobj(1, zlib.DEFLATED, 31)
    return c.compress(text) + c.flush()


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    n = int(args[0]) if args else 200000
    members = int(args[1]) if len(args) > 1 else 64
    groups = config_groups("nbd96")
    b, o = A.synth_reads_host(groups, 5, 4000, 4000, 0, n)
    per = (n + members - 1) // members
    texts = []
    for m in range(members):
        lo, hi = m * per, min(n, (m + 1) * per)
        texts.append(b"".join(b"@r%d\n" % i + b[int(o[i]):int(o[i + 1])].tobytes() + b"\n+\n" + b"I" * int(o[i + 1] - o[i]) + b"\n" for i in range(lo, hi)))
    with Pool(effective_cpus()) as pool:
        blobs = pool.map(gz_member, texts)
    d = "/dev/shm" if os.path.isdir("/dev/shm") else "/tmp"
    gz = f"{d}/members.fastq.gz"
    with open(gz, "wb") as f:
        for x in blobs:
            f.write(x)
    print("reads", n, "members", members, "text", sum(len(t) for t in texts), "compressed", os.path.getsize(gz), flush=True)
    stage = "--stage" in sys.argv
    outs = {}
    for name, env in (("members side by side", {}), ("one after the other", {"BARBELL_AMD_GZ_SERIAL": "1"})):
        cmd = ([CLI, "stage", "-i", gz, "-o", f"{d}/members.out", "-t", "32"] if stage else
               [CLI, "annotate", "-i", gz, "-o", f"{d}/members.tsv", "--kit", "SQK-NBD114-96", "--flank-max-errors", "3", "-t", "32"])
        t0 = time.time()
        r = subprocess.run(cmd, capture_output=True, text=True, env=dict(os.environ, BARBELL_AMD_PROFILE="1", BARBELL_AMD_NO_TORCH="1", **env))
        wall = time.time() - t0
        outs[name] = open(f"{d}/members.out" if stage else f"{d}/members.tsv", "rb").read() if r.returncode == 0 else None
        m = re.search(r"profile: gzip[^\n]*", r.stderr)
        print(f"{name:22s} rc {r.returncode} wall {wall:6.2f} s  {n / wall / 1e6:6.2f} M reads/s   {m.group(0)[:200] if m else r.stderr[-300:]}", flush=True)
    print("same output", outs["members side by side"] == outs["one after the other"] and outs["one after the other"] is not None)
    os.remove(gz)
