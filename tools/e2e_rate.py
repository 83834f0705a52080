#!/usr/bin/env python3
"""End-to-end wall clock of the CLI on a FASTQ file (file IO + GPU ingest + annotate [+ filter + trim]).
Run on the GPU box: python tools/e2e_rate.py [n_reads] [read_len].  Prints one JSON line."""
import json
import os
import subprocess
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from barbell_amd import annotate as A, kits  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 400_000
L = int(sys.argv[2]) if len(sys.argv) > 2 else 4000
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
cli = os.path.join(root, "barbell_amd", "bin", "barbell-amd")
tmp = os.environ.get("TMPDIR", "/tmp")
fq = os.path.join(tmp, "e2e.fastq")
groups = kits.groups_from_kit("SQK-NBD114-96")
t0 = time.time()
with open(fq, "wb") as f:
    step = 50_000
    for first in range(0, n, step):
        m = min(step, n - first)
        bases, off = A.synth_reads_host(groups, 1234, L, L, first, m)
        b = bases.reshape(m, L)
        hdr = np.tile(np.frombuffer(b"@r00000000 ch=0000 st=2024-01-01T00:00Z\n", dtype=np.uint8), (m, 1))
        idx = np.arange(first, first + m)
        for d in range(8):
            hdr[:, 9 - d] = 48 + (idx // 10 ** d) % 10
        q = np.full((m, L), 53, dtype=np.uint8)
        sep = np.tile(np.frombuffer(b"\n+\n", dtype=np.uint8), (m, 1))
        nl = np.full((m, 1), 10, dtype=np.uint8)
        f.write(np.concatenate([hdr, b, sep, q, nl], axis=1).tobytes())
gen_s = time.time() - t0
size = os.path.getsize(fq)
env = dict(os.environ, BARBELL_AMD_NO_TORCH="1", BARBELL_AMD_PROFILE="1")
out = {"n_reads": n, "read_len": L, "fastq_bytes": size, "gen_s": gen_s}
for name, cmd in (("annotate", [cli, "annotate", "-i", fq, "-o", os.path.join(tmp, "e2e_a.tsv"), "--kit", "SQK-NBD114-96", "--flank-max-errors", "3"]),
                  ("kit", [cli, "kit", "-k", "SQK-NBD114-96", "-i", fq, "-o", os.path.join(tmp, "e2e_kit"), "--flank-max-errors", "3", "--maximize"])):
    t0 = time.time()
    r = subprocess.run(cmd, capture_output=True, text=True, env=env)
    dt = time.time() - t0
    assert r.returncode == 0, r.stderr
    out[name] = {"wall_s": dt, "reads_per_s": n / dt, "fastq_gb_per_s": size / dt / 1e9, "profile": [l for l in r.stderr.splitlines() if l.startswith("profile:")]}
print(json.dumps(out))

# several gzip files: parallel inflate (-t 8) against one inflating thread (-t 1)
if len(sys.argv) > 3 and sys.argv[3] == "gz":
    parts = []
    per = (n // 8) * (size // n)
    with open(fq, "rb") as f:
        for k in range(8):
            p = os.path.join(tmp, f"e2e_part{k}.fastq")
            with open(p, "wb") as o:
                o.write(f.read(per))
            subprocess.run(["gzip", "-1", "-f", p], check=True)
            parts.append(p + ".gz")
    res = {}
    for t in (1, 8):
        t0 = time.time()
        r = subprocess.run([cli, "annotate", "-i"] + parts + ["-o", os.path.join(tmp, "e2e_gz.tsv"), "--kit", "SQK-NBD114-96", "--flank-max-errors", "3",
                            "-t", str(t)], capture_output=True, text=True, env=env)
        dt = time.time() - t0
        assert r.returncode == 0, r.stderr
        res[f"threads_{t}"] = {"wall_s": dt, "reads_per_s": 8 * (n // 8) / dt, "gz_bytes": sum(os.path.getsize(p) for p in parts)}
    print(json.dumps({"gz_8_files": res}))
