#!/usr/bin/env python3
"""End-to-end rate of the C++ host (barbell-amd) on a FASTQ file: file reads (page cache) + upload + GPU parse + annotate
+ GPU-rendered TSV [+ filter + trim] + file writes.  Run on the GPU box:

    python tools/e2e_rate.py [n_reads=4000000] [read_len=4000] [--kit-run] [--json OUT]

The FASTQ is synthesised on the GPU (the bench generator) and written once; each CLI run then reports its own
steady-state figure — the time between the first block being requested and the last block being committed
(AnnotateStats::seconds_pipeline, printed on the "Done:" line), which leaves out process start, context creation and
teardown — next to the wall clock of the whole process.  Prints one JSON line."""
import argparse
import json
import os
import re
import subprocess
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from barbell_amd import annotate as A, kits  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("n_reads", nargs="?", type=int, default=4_000_000)
ap.add_argument("read_len", nargs="?", type=int, default=4000)
ap.add_argument("--kit-run", action="store_true", help="also time `barbell-amd kit` (annotate + inspect + filter + trim)")
ap.add_argument("--only-kit", action="store_true", help="skip the annotate runs")
ap.add_argument("--kit-env", action="append", default=[], help="NAME=VALUE[,NAME=VALUE..]: one more kit run (3 streams) under this environment")
ap.add_argument("--quick", action="store_true", help="only the default operating point (2 streams, 256 MiB blocks), three times, and its text-lines form")
ap.add_argument("--json")
ap.add_argument("--dir", default=os.environ.get("TMPDIR", "/tmp"))
a = ap.parse_args()
n, L = a.n_reads, a.read_len
cli = os.path.join(ROOT, "barbell_amd", "bin", "barbell-amd")
fq = os.path.join(a.dir, "e2e.fastq")
groups = kits.groups_from_kit("SQK-NBD114-96", flank_max_errors=3)
dm = A.Demuxer()
for g in groups:
    dm.add_query_group(g)
dev = torch.device("cuda", 0)
t0 = time.time()
with open(fq, "wb") as f:
    step = 250_000
    for first in range(0, n, step):
        m = min(step, n - first)
        d_off = torch.arange(0, m + 1, dtype=torch.int64, device=dev) * L
        d_bases = torch.empty(m * L, dtype=torch.uint8, device=dev)
        torch.cuda.synchronize()
        dm.synth_dev(0xBA7BE11 ^ 2, L, L, first, m, d_off.data_ptr(), d_bases.data_ptr())
        hdr = np.tile(np.frombuffer(b"@r00000000 ch=0000 st=2024-01-01T00:00Z\n", dtype=np.uint8), (m, 1))
        idx = np.arange(first, first + m)
        for d in range(8):
            hdr[:, 9 - d] = 48 + (idx // 10 ** d) % 10
        q = torch.full((m, L), 53, dtype=torch.uint8, device=dev)
        sep = torch.tensor(list(b"\n+\n"), dtype=torch.uint8, device=dev).repeat(m, 1)
        nl = torch.full((m, 1), 10, dtype=torch.uint8, device=dev)
        text = torch.cat([torch.from_numpy(hdr).to(dev), d_bases.view(m, L), sep, q, nl], dim=1).contiguous().view(-1)
        text.cpu().numpy().tofile(f)
gen_s = time.time() - t0
dm.close()
del dm
torch.cuda.empty_cache()
size = os.path.getsize(fq)
env = dict(os.environ, BARBELL_AMD_NO_TORCH="1", BARBELL_AMD_PROFILE="1")
out = {"n_reads": n, "read_len": L, "fastq_bytes": size, "gen_s": gen_s, "host_cores": os.cpu_count(), "runs": {}}


def run(name, cmd):
    t0 = time.time()
    r = subprocess.run(cmd, capture_output=True, text=True, env=env)
    dt = time.time() - t0
    assert r.returncode == 0, r.stderr[-2000:]
    m = re.search(r"\(([\d.]+) s in the pipeline, ([\d.]+) M reads/s", r.stderr)
    pipe_s = float(m.group(1)) if m else None
    out["runs"][name] = {"wall_s": dt, "wall_reads_per_s": n / dt, "pipeline_s": pipe_s, "steady_state_reads_per_s": n / pipe_s if pipe_s else None,
                         "fastq_gb_per_s_steady": size / pipe_s / 1e9 if pipe_s else None,
                         "profile": [l for l in r.stderr.splitlines() if l.startswith("profile:")]}


base = [cli, "annotate", "-i", fq, "-o", os.path.join(a.dir, "e2e_a.tsv"), "--kit", "SQK-NBD114-96", "--flank-max-errors", "3"]
if a.quick:
    for rep in range(3):
        run(f"annotate_streams2_block256Mi_t32_rep{rep}", base + ["--streams", "2", "--block-bytes", str(256 << 20), "-t", "32"])
    run("annotate_text_lines_streams2_block256Mi_t32", base + ["--streams", "2", "--block-bytes", str(256 << 20), "-t", "32", "--no-pack"])
    for thr in ("4", "10", "16"):   # -t: the reference's default is 10 (worker threads there, reader threads here)
        run(f"annotate_defaults_t{thr}", base + ["-t", thr])
    run("annotate_defaults", base)
    best = max(out["runs"], key=lambda k: out["runs"][k]["steady_state_reads_per_s"] or 0)
    out["best"] = {"run": best, **{k: out["runs"][best][k] for k in ("steady_state_reads_per_s", "wall_reads_per_s", "wall_s")}}
elif not a.only_kit:
    for streams in (1, 2, 3, 4):
        for bb in (128 << 20, 256 << 20, 512 << 20):
            run(f"annotate_streams{streams}_block{bb >> 20}Mi_t32", base + ["--streams", str(streams), "--block-bytes", str(bb), "-t", "32"])
    # the upload forms side by side (default: header lines + two bases per byte; --no-pack: sequence lines as text; --no-compact: whole records)
    for name, extra in (("text_lines", ["--no-pack"]), ("whole_records", ["--no-compact"])):
        run(f"annotate_{name}_streams2_block256Mi_t32", base + ["--streams", "2", "--block-bytes", str(256 << 20), "-t", "32"] + extra)
    run("annotate_streams2_block256Mi_t64", base + ["--streams", "2", "--block-bytes", str(256 << 20), "-t", "64"])
    env_keep = env
    env = dict(env_keep, BARBELL_AMD_PINNED_SLOTS="1")
    run("annotate_streams2_block256Mi_t32_pinned_slots", base + ["--streams", "2", "--block-bytes", str(256 << 20), "-t", "32"])
    env = env_keep
    out["tsv_bytes"] = os.path.getsize(os.path.join(a.dir, "e2e_a.tsv"))
    # the host side alone (BARBELL_AMD_FEED_ONLY=1: files -> reader threads -> blocks of whole records in the upload buffers, no GPU call):
    # what ONE host process can feed, i.e. the ceiling of a node's end-to-end rate however many GPUs take the blocks
    env_run = env
    for thr in (32, 64):
        env = dict(env_run, BARBELL_AMD_FEED_ONLY="1")
        t0 = time.time()
        r = subprocess.run(base + ["--streams", "4", "--block-bytes", str(128 << 20), "-t", str(thr)], capture_output=True, text=True, env=env)
        mm = re.search(r"feed-only: (\d+) bytes of staged text in ([\d.]+) s", r.stderr)
        if r.returncode == 0 and mm:
            out.setdefault("feed_only", {})[f"t{thr}"] = {"staged_bytes": int(mm.group(1)), "seconds": float(mm.group(2)), "reads_per_s": n / float(mm.group(2)),
                                                          "fastq_gb_per_s": size / float(mm.group(2)) / 1e9, "wall_s": time.time() - t0}
    env = env_run
    best = max(out["runs"], key=lambda k: out["runs"][k]["steady_state_reads_per_s"] or 0)
    bw = min(out["runs"], key=lambda k: out["runs"][k]["wall_s"])
    out["best_wall"] = {"run": bw, **{k: out["runs"][bw][k] for k in ("wall_s", "wall_reads_per_s", "steady_state_reads_per_s")}}
    out["best"] = {"run": best, **{k: out["runs"][best][k] for k in ("steady_state_reads_per_s", "wall_reads_per_s", "fastq_gb_per_s_steady")}}
if a.kit_run:
    import shutil

    for streams in (2, 3, 4):
        shutil.rmtree(os.path.join(a.dir, "e2e_kit"), ignore_errors=True)
        run(f"kit_streams{streams}", [cli, "kit", "-k", "SQK-NBD114-96", "-i", fq, "-o", os.path.join(a.dir, "e2e_kit"), "--flank-max-errors", "3", "--maximize",
                                      "--streams", str(streams), "-t", "32"])
    kit_cmd = lambda streams, extra=(): [cli, "kit", "-k", "SQK-NBD114-96", "-i", fq, "-o", os.path.join(a.dir, "e2e_kit"), "--flank-max-errors", "3", "--maximize",
                                         "--streams", str(streams), "-t", "32"] + list(extra)
    shutil.rmtree(os.path.join(a.dir, "e2e_kit"), ignore_errors=True)
    run("kit_streams3_gpu_render", kit_cmd(3, ["--gpu-render"]))
    shutil.rmtree(os.path.join(a.dir, "e2e_kit"), ignore_errors=True)
    run("kit_streams3_block256Mi", kit_cmd(3, ["--batch-reads", "65536"]))
    base_env = env
    for spec in a.kit_env:
        env = dict(base_env, **dict(kv.split("=", 1) for kv in spec.split(",")))
        shutil.rmtree(os.path.join(a.dir, "e2e_kit"), ignore_errors=True)
        run("kit_streams3_" + spec, kit_cmd(3))
    env = base_env
txt = json.dumps(out)
if a.json:
    open(a.json, "w").write(txt + "\n")
print(txt)
