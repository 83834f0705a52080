#!/usr/bin/env python3
"""Summarises rocprofv3 --pmc counter_collection CSVs: per kernel, sum of each counter over dispatches.
usage: pmc_summary.py dir_or_csv [...]"""
import csv
import glob
import os
import sys
from collections import defaultdict

acc = defaultdict(lambda: defaultdict(float))
calls = defaultdict(int)
for arg in sys.argv[1:]:
    files = glob.glob(os.path.join(arg, "**", "*counter_collection.csv"), recursive=True) if os.path.isdir(arg) else [arg]
    for f in files:
        seen = set()
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0][:60]
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            key = (k, r["Dispatch_Id"])
            if key not in seen:
                seen.add(key)
for k in sorted(acc, key=lambda k: -acc[k].get("SQ_BUSY_CYCLES", acc[k].get("FETCH_SIZE", 0))):
    if not any(x in k for x in ("k_", "bb_")):
        continue
    print(k)
    for c in sorted(acc[k]):
        print(f"    {c:28s} {acc[k][c]:20.0f}")
