#!/usr/bin/env python3
"""Turns the SQ counter passes of tools/profile_round.sh (rocprofv3 --pmc, counter_collection CSVs) into
profiles/valu_<config>.json: per kernel and launch, VALU / SALU / LDS instruction counts and the wave-cycle split
(active / issue-stalled / parked), next to the measured issue ceilings of profiles/valu_ceiling.json.
usage: collect_valu.py PMC_DIR... OUT_JSON BATCH_READS READ_LEN"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

*dirs, out, batch, read_len = sys.argv[1:]
acc = defaultdict(lambda: defaultdict(float))
calls = defaultdict(lambda: defaultdict(set))
for d in dirs:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "")
            if not k.startswith("k_") or k.startswith("k_synth"):
                continue
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            calls[k][r["Counter_Name"]].add(r["Dispatch_Id"])
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ceil = json.load(open(os.path.join(ROOT, "profiles", "valu_ceiling.json")))
peak = lambda k: max(x["G"] for x in ceil["classes"][k]["ind"].values())
res = {"batch_reads": int(batch), "read_len": int(read_len),
       "ceilings_G_wave_instr_per_s": {"full_rate (and/or/xor/add/sub/mov/lshr/bitop3)": peak("v_add_u32"),
                                       "half_rate (lshl/alignbit/bfe/brev/cmp/cndmask/add3/or3/mad24/f64/64-bit)": peak("v_lshl_or_b32")},
       "note": "counts per launch; SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_ANY are quad-cycles summed over waves", "kernels": {}}
for k in acc:
    e = {}
    for c, v in acc[k].items():
        e[c] = v / max(1, len(calls[k][c]))
    wc = e.get("SQ_WAVE_CYCLES")
    if wc and "SQ_WAIT_ANY" in e:
        e["frac_parked(s_waitcnt/barrier)"] = e["SQ_WAIT_ANY"] / wc
        e["frac_issue_stalled"] = e.get("SQ_WAIT_INST_ANY", 0.0) / wc
        e["frac_active"] = e.get("SQ_ACTIVE_INST_ANY", 0.0) / wc
    res["kernels"][k] = e
json.dump(res, open(out, "w"), indent=1, sort_keys=True)
print(out, len(res["kernels"]), "kernels")
