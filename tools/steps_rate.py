#!/usr/bin/env python3
"""dev aid (GPU box): what the stand-alone steps cost — `barbell-amd kit` on N synthetic 4 kb reads, then `barbell-amd filter`, `inspect`,
`trim` and the Python twin's filter on the files it wrote.  usage: steps_rate.py [N_READS=200000]"""
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from barbell_amd import annotate as A, kits  # noqa: E402

CLI = os.path.join(ROOT, "barbell_amd", "bin", "barbell-amd")
KIT = "SQK-NBD114-96"


def timed(*args):
    t = time.perf_counter()
    r = subprocess.run(args, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-800:]
    return time.perf_counter() - t


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
    groups = kits.groups_from_kit(KIT, flank_max_errors=3)
    base = "/dev/shm" if os.path.isdir("/dev/shm") else None
    with tempfile.TemporaryDirectory(dir=base) as d:
        fq = os.path.join(d, "r.fastq")
        with open(fq, "wb") as f:
            done = 0
            while done < n:
                m = min(50000, n - done)
                bases, offsets = A.synth_reads_host(groups, 7, 4000, 4000, done, m)
                q = b"5" * 4000
                f.write(b"".join(b"@r%d ch=1\n" % (done + i) + bases[int(offsets[i]):int(offsets[i + 1])].tobytes() + b"\n+\n" + q + b"\n" for i in range(m)))
                done += m
        out = {"reads": n, "kit": KIT}
        out["kit_s"] = timed(CLI, "kit", "-k", KIT, "-i", fq, "-o", os.path.join(d, "kit"), "--maximize", "--flank-max-errors", "3")
        anno, filt = os.path.join(d, "kit", "annotation.tsv"), os.path.join(d, "kit", "filtered.tsv")
        out["rows"] = sum(1 for _ in open(anno)) - 1
        data = kits._data()
        pats = os.path.join(d, "p.txt")
        open(pats, "w").write("\n".join(data["pattern_sets"][data["kit_filter"][KIT]["maximize"]]) + "\n")
        out["filter_s"] = timed(CLI, "filter", "-i", anno, "-o", os.path.join(d, "f.tsv"), "-f", pats)
        out["filter_identical"] = open(os.path.join(d, "f.tsv"), "rb").read() == open(filt, "rb").read()
        out["inspect_s"] = timed(CLI, "inspect", "-i", anno, "-o", os.path.join(d, "ppr.tsv"))
        out["trim_s"] = timed(CLI, "trim", "-i", filt, "-r", fq, "-o", os.path.join(d, "t"), "--no-orientation", "--no-flanks", "--only-side", "left")
        out["python_filter_s"] = timed(sys.executable, "-m", "barbell_amd", "filter", "-i", anno, "-o", os.path.join(d, "pf.tsv"), "-f", pats)
        out["python_filter_identical"] = open(os.path.join(d, "pf.tsv"), "rb").read() == open(filt, "rb").read()
        for k in ("filter_s", "inspect_s", "python_filter_s"):
            out[k.replace("_s", "_rows_per_s")] = out["rows"] / out[k]
        out["trim_reads_per_s"] = n / out["trim_s"]
        print(json.dumps(out))


if __name__ == "__main__":
    main()
