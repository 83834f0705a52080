#!/usr/bin/env python3
"""Turns rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE counter_collection CSVs (collected in SEPARATE
passes, as /opt/skills/guides/MI355X_MICROARCH.md §HBM prescribes) into profiles/<name>.json:
per-kernel HBM bytes per launch.  gfx950 correction applied as the guide says: FETCH_SIZE counts
128-byte requests of wide (16 B/lane) loads at 64 B, so the read side is doubled; WRITE_SIZE is
taken as reported (uncalibrated).  Units of both counters: KiB.
usage: collect_traffic.py PMC_DIR OUT_JSON BATCH_READS READ_LEN "<bench command the counters were collected with>" """
import csv
import glob
import json
import os
import sys
from collections import defaultdict

pmc_dir, out, batch, read_len, cmd = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), sys.argv[5]
acc = defaultdict(lambda: defaultdict(float))
calls = defaultdict(lambda: defaultdict(int))
for f in glob.glob(os.path.join(pmc_dir, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] not in ("FETCH_SIZE", "WRITE_SIZE"):
            continue
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        calls[k][r["Counter_Name"]] += 1
res = {"command": cmd, "batch_reads": batch, "read_len": read_len, "note": "bytes per launch; hbm_bytes = 2*FETCH_SIZE*1024 + WRITE_SIZE*1024 (gfx950 FETCH_SIZE half-count correction)", "kernels": {}}
for k in acc:
    if not k.startswith("k_") or k.startswith("k_synth"):  # k_synth is bench setup, not the path
        continue
    f = acc[k]["FETCH_SIZE"] / max(1, calls[k]["FETCH_SIZE"])
    w = acc[k]["WRITE_SIZE"] / max(1, calls[k]["WRITE_SIZE"])
    # launches per step from the data: k_collapse runs exactly once per step (multi-group configs launch the scan kernels per group,
    # the barcode kernels per (group, strand))
    steps_seen = max(1, calls.get("k_collapse", {}).get("FETCH_SIZE", 0))
    per_step = max(1, round(calls[k]["FETCH_SIZE"] / steps_seen)) if "k_collapse" in calls else {"k_scan_block": 2, "k_scan_sums": 2, "k_scan_add": 2}.get(k, 1)
    # kernels of bench.py's filter / trim / ingest legs (outside the timed annotate step) are listed but not part of the step
    in_step = not (k.startswith(("k_filter", "k_rs_", "k_scan32", "k_scan64", "k_scan_sums_t", "k_trim", "k_fq_", "k_nl_", "k_fmt", "k_inspect", "k_iota")))
    res["kernels"][k] = {"launches_seen": calls[k]["FETCH_SIZE"], "launches_per_step": per_step, "FETCH_SIZE_KiB": f, "WRITE_SIZE_KiB": w,
                         "hbm_bytes": 2 * f * 1024 + w * 1024, "in_step": in_step}
json.dump(res, open(out, "w"), indent=1, sort_keys=True)
print(json.dumps(res, indent=1, sort_keys=True))
