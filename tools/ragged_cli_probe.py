#!/usr/bin/env python3
"""dev aid (GPU box): the C++ host + FASTQ ingest on reads of differing lengths against equal reads of the same mean (same bytes of FASTQ):
does a 100 kb record cost the ingest kernels or the TSV path anything?  usage: ragged_cli_probe.py [n_reads]"""
import os
import re
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from barbell_amd import annotate as A  # noqa: E402
from tests.common import config_groups, heavy_tailed_batch  # noqa: E402

CLI = os.path.join(ROOT, "barbell_amd", "bin", "barbell-amd")


def write_fastq(path, bases, offs):
    with open(path, "wb") as f:
        chunks = []
        for i in range(len(offs) - 1):
            s = bases[int(offs[i]):int(offs[i + 1])].tobytes()
            chunks.append(b"@r%d\n" % i + s + b"\n+\n" + b"I" * len(s) + b"\n")
            if len(chunks) >= 4096:
                f.write(b"".join(chunks)); chunks = []
        f.write(b"".join(chunks))


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 300000
    groups = config_groups("nbd96")
    hb, ho = heavy_tailed_batch(groups, n)
    mean = int(len(hb) / (len(ho) - 1))
    fb, fo = A.synth_reads_host(groups, 9, mean, mean, 0, len(ho) - 1)
    d = "/dev/shm" if os.path.isdir("/dev/shm") else "/tmp"
    for name, (b, o) in (("equal", (fb, fo)), ("heavy_tailed", (hb, ho))):
        fq = f"{d}/ragged_{name}.fastq"
        write_fastq(fq, b, o)
        for rep in range(2):
            t0 = time.time()
            r = subprocess.run([CLI, "annotate", "-i", fq, "-o", f"{d}/ragged_{name}.tsv", "--kit", "SQK-NBD114-96", "--flank-max-errors", "3", "-t", "32"],
                               capture_output=True, text=True, env=dict(os.environ, BARBELL_AMD_PROFILE="1", BARBELL_AMD_NO_TORCH="1"))
            wall = time.time() - t0
            m = re.search(r"pipeline ([0-9.]+) s", r.stderr)
            print(name, "rc", r.returncode, "wall %.2f s" % wall, "pipeline", m.group(1) if m else "?", "bytes", os.path.getsize(fq), flush=True)
            if r.returncode:
                print(r.stderr[-600:])
        for l in r.stderr.splitlines():
            if "profile:" in l:
                print("   ", l[:200])
        # the whole kit pipeline (annotate -> filter -> trim, per-barcode FASTQ files written): the trim kernels on long records
        import shutil
        outd = f"{d}/ragged_kit_{name}"
        t0 = time.time()
        r = subprocess.run([CLI, "kit", "-k", "SQK-NBD114-96", "-i", fq, "-o", outd, "--maximize", "--flank-max-errors", "3", "-t", "32"],
                           capture_output=True, text=True, env=dict(os.environ, BARBELL_AMD_PROFILE="1", BARBELL_AMD_NO_TORCH="1"))
        m = re.search(r"pipeline ([0-9.]+) s", r.stderr)
        print(name, "kit rc", r.returncode, "wall %.2f s" % (time.time() - t0), "pipeline", m.group(1) if m else "?", flush=True)
        for l in r.stderr.splitlines():
            if "profile: pipeline" in l:
                print("   ", l[:260])
        shutil.rmtree(outd, ignore_errors=True)
        if "--gzip-out" in sys.argv:   # the per-label files gzip-compressed: libdeflate members against zlib's gzwrite
            for lib, e2 in (("libdeflate", {}), ("zlib", {"BARBELL_AMD_NO_LIBDEFLATE": "1"})):
                t0 = time.time()
                r = subprocess.run([CLI, "kit", "-k", "SQK-NBD114-96", "-i", fq, "-o", outd, "--maximize", "--flank-max-errors", "3", "--gzip"],
                                   capture_output=True, text=True, env=dict(os.environ, BARBELL_AMD_PROFILE="1", BARBELL_AMD_NO_TORCH="1", **e2))
                sz = sum(os.path.getsize(os.path.join(outd, x)) for x in os.listdir(outd)) if os.path.isdir(outd) else 0
                print(name, "kit --gzip", lib, "rc", r.returncode, "wall %.2f s" % (time.time() - t0), "output bytes", sz, flush=True)
                shutil.rmtree(outd, ignore_errors=True)
        os.remove(fq)
