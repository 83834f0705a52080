#!/usr/bin/env python3
"""The C ABI driven the way integration/annotator.patch drives it (bench.py's `boundary_step`).

`barbell_amd/bin/bb-boundary-bench` (csrc/host/boundary_bench.cpp) runs T worker threads — paraseq's `process_parallel(.., n_threads, ..)`,
/root/reference/src/annotate/annotator.rs:278-280 — each with its own context (one Demuxer per worker, annotator.rs:88-101), each calling
bb_annotate_batch on ordinary (pageable) host memory with B reads per call and taking the rows back.  This module runs it over a grid of
(B, T), fits the per-call fixed cost (intercept of a T = 1 call's time over B), runs the batch size and thread count INTEGRATION.md names
in the one-byte-per-base and the two-bases-per-byte form, and checks that the rows do not depend on B or T (an order-independent hash of
all rows with batch-global read indices).

    python tools/boundary_rate.py [--reads N] [--seconds S] [--out FILE]
"""
import argparse
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "barbell_amd", "bin", "bb-boundary-bench")
# what integration/annotator.patch and INTEGRATION.md ("Batch size") name: DemuxProcessor collects paraseq batches until it holds
# NAMED_BATCH reads, `barbell annotate -t` defaults to 10 worker threads (bin/main.rs:69-71)
NAMED_BATCH, NAMED_THREADS = 8192, 10


def harness(*flags, timeout=600):
    if not os.path.exists(BIN):
        raise FileNotFoundError(f"{BIN}: build it with barbell_amd/csrc/build.sh")
    out = subprocess.run([BIN, *map(str, flags)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=timeout, check=False)
    if out.returncode != 0:
        raise RuntimeError(f"bb-boundary-bench {' '.join(map(str, flags))}: rc {out.returncode}: {out.stderr.decode(errors='replace')[-300:]}")
    return json.loads(out.stdout.decode())


def _slim(run):
    keep = ("batch", "threads", "reads_per_s", "ms_per_call", "host_syncs_per_call")
    d = {k: (round(run[k], 4) if isinstance(run[k], float) else run[k]) for k in keep}
    if "phase_ms_per_call" in run and run["threads"] == 1 and run["batch"] <= 65536:   # (the phases of a lone caller: with more callers they mostly show the waiting)
        d["phase_ms_per_call"] = {k: v for k, v in run["phase_ms_per_call"].items() if k in ("upload", "pipeline", "rows_back")}
    return d


def measure(reads=524288, seconds=1.0, cpus=None, batches=(1024, 8192, 65536, 524288), check_reads=65536):
    from barbell_amd.parallel import effective_cpus

    cpus = cpus or effective_cpus()
    threads = sorted({1, NAMED_THREADS, cpus})
    grid = harness("--reads", reads, "--batch", ",".join(map(str, batches)), "--threads", ",".join(map(str, threads)), "--seconds", seconds, "--phases")
    runs = [_slim(r) for r in grid["runs"]]
    by = {(r["batch"], r["threads"]): r for r in runs}
    # per-call fixed cost: time of a one-thread call = a + b * B; from the two smallest batches (beyond them PCIe's share is no longer linear in B alone)
    b0, b1 = sorted(batches)[:2]
    t0, t1 = by[(b0, 1)]["ms_per_call"], by[(b1, 1)]["ms_per_call"]
    slope = (t1 - t0) / (b1 - b0)
    fixed_ms = t0 - slope * b0
    named = harness("--reads", min(reads, 262144), "--batch", NAMED_BATCH, "--threads", NAMED_THREADS, "--seconds", seconds)["runs"][0]
    named_packed = harness("--reads", min(reads, 262144), "--batch", NAMED_BATCH, "--threads", NAMED_THREADS, "--seconds", seconds, "--packed")["runs"][0]
    named_prepacked = harness("--reads", min(reads, 262144), "--batch", NAMED_BATCH, "--threads", NAMED_THREADS, "--seconds", seconds, "--prepacked")["runs"][0]
    # rows across batch sizes and thread counts
    chk = harness("--reads", check_reads, "--batch", f"1024,{NAMED_BATCH},{check_reads}", "--threads", f"1,{NAMED_THREADS}", "--check")
    chk_p = harness("--reads", check_reads, "--batch", f"1024,{check_reads}", "--threads", f"1,{NAMED_THREADS}", "--check", "--packed")
    hashes = {(r["rows_hash"], r["rows"]) for r in chk["runs"] + chk_p["runs"]}
    best_1k = max(r["reads_per_s"] for r in runs if r["batch"] == 1024)
    return {
        "what": "T worker threads x own context x bb_annotate_batch on pageable host memory, B reads per call (csrc/host/boundary_bench.cpp); "
                "reads/s of the whole process, wall clock, rows copied back; outside `value`",
        "workload": f"SQK-NBD114-96, flank-max-errors 3, {reads} synthetic 4000-nt reads in host memory",
        "host_cpus": cpus,
        "runs": runs,
        "per_call_fixed_ms_one_thread": round(fixed_ms, 4), "per_read_us_one_thread": round(slope * 1e3, 4),
        "host_waits_per_call": by[(b0, 1)]["host_syncs_per_call"],
        "named": {"batch": NAMED_BATCH, "threads": NAMED_THREADS, "where": "integration/annotator.patch (GPU_BATCH_READS), INTEGRATION.md 'Batch size'",
                  "reads_per_s": round(named["reads_per_s"], 1),
                  "reads_per_s_packed_by_the_worker": round(named_packed["reads_per_s"], 1),
                  "reads_per_s_packed_beforehand": round(named_prepacked["reads_per_s"], 1)},
        "calls_of_1024_reads": {"best_reads_per_s": round(best_1k, 1), "at_named_threads": by[(1024, NAMED_THREADS)]["reads_per_s"],
                                "round5_at_named_threads": 1.58e6, "round5_one_thread": 0.88e6},
        "rows_identical_across_batch_sizes_threads_and_forms": len(hashes) == 1,
        "rows_hash": sorted(h for h, _ in hashes),
    }


if __name__ == "__main__":
    sys.path.insert(0, ROOT)
    ap = argparse.ArgumentParser()
    ap.add_argument("--reads", type=int, default=524288)
    ap.add_argument("--seconds", type=float, default=1.0)
    ap.add_argument("--out")
    a = ap.parse_args()
    res = measure(a.reads, a.seconds)
    text = json.dumps(res, indent=1)
    if a.out:
        with open(a.out, "w") as f:
            f.write(text + "\n")
    print(text)
