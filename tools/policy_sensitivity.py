#!/usr/bin/env python3
"""What the unpinned assumptions are worth (include/barbell_amd_policy.h): for every single-field alternative of the
policy — all 18 distinguishable traceback orders, the rc path convention, the local-minimum rules, rc order, overhang roundings, the tie
rule, the Lodhi variants — how many reads of a noisy synthetic sample get a different answer than under the default policy:

    reads_changed     any byte of any of the read's rows differs (coordinates, costs, label, strand, row count)
    label_changed     the read's multiset of (match type, barcode label, strand) differs: what demultiplexing acts on
    tag_flank_flip    a row that is a tag (Ftag / Rtag) under one policy is flank-only (Fflank / Rflank) under the other, or appears / vanishes

Reads: the BASELINE query sets' synthetic reads (bb_synth.h) with sequencing-like noise over the WHOLE read (substitutions, insertions
and deletions, default 4 % + 2 % + 2 %), so that scores land near the thresholds and equal-cost alignments appear — on clean synthetic
reads most alternatives change nothing.  Backends: the CPU checker's bit-parallel path (`--backend oracle`, any box) or the HIP path
(`--backend hip`, a GPU box); `--backend both` runs both and requires identical rows per policy (parity at sample size, per policy).

    python tools/policy_sensitivity.py --reads 200000 --backend both --out profiles/policy_sensitivity.json
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from barbell_amd import _abi  # noqa: E402
from barbell_amd.parallel import effective_cpus  # noqa: E402
from tests.common import ALTERNATIVES, config_groups, is_feasible  # noqa: E402

HAZARD_TEXT = {
    "H1": "which end positions <= k sassy's search reports (plateau end / strict minima)",
    "H2": "order of the rc matches in the returned Vec",
    "H3": "traceback preference among equal-cost alignments (18 distinguishable orders)",
    "H4": "rounding of the overhang cost alpha * o",
    "H5": "pattern indices of Match::to_path() for Strand::Rc matches (forward or mirrored)",
    "H7": "which of several equally cheap minima of one barcode pattern is kept",
    "H8": "Lodhi::compute's formula (decay exponents per op; p and lambda are pinned by searcher.rs:209 and listed only)",
}


def mutate(bases, offsets, sub, ins, dele, seed, return_map=False):
    """sequencing-like noise over whole reads: each base deleted with probability `dele`, else substituted with `sub`; a random base
    inserted after it with probability `ins`.  Vectorised; returns (bases, offsets) [, position map: old index -> new index]."""
    rng = np.random.default_rng(seed)
    n = len(bases)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    keep = rng.random(n) >= dele
    b = bases.copy()
    s = rng.random(n) < sub
    b[s] = acgt[rng.integers(0, 4, int(s.sum()))]
    extra = (rng.random(n) < ins) & keep
    cnt = keep.astype(np.int64) + extra
    out = np.repeat(b, cnt)
    # the second copy of a base with an insertion becomes the inserted base
    ends = np.cumsum(cnt)
    ins_pos = ends[extra] - 1
    out[ins_pos] = acgt[rng.integers(0, 4, len(ins_pos))]
    csum = np.concatenate([[0], ends])
    new_off = csum[offsets.astype(np.int64)].astype(np.uint64)
    if return_map:
        return out, new_off, csum
    return out, new_off


def per_read(rows, n_reads):
    """start index of every read's rows (rows are in read order)"""
    return np.searchsorted(rows["read_idx"], np.arange(n_reads + 1))


def compare(a, b, n_reads):
    ia, ib = per_read(a, n_reads), per_read(b, n_reads)
    changed = label = flip = 0
    same_len = (ia[1:] - ia[:-1]) == (ib[1:] - ib[:-1])
    tag = lambda r: r["barcode_idx"] >= 0
    for r in range(n_reads):
        ra, rb = a[ia[r]:ia[r + 1]], b[ib[r]:ib[r + 1]]
        if same_len[r] and ra.tobytes() == rb.tobytes():
            continue
        changed += 1
        ka = sorted(zip(ra["group_idx"].tolist(), ra["match_type"].tolist(), ra["barcode_idx"].tolist(), ra["strand"].tolist()))
        kb = sorted(zip(rb["group_idx"].tolist(), rb["match_type"].tolist(), rb["barcode_idx"].tolist(), rb["strand"].tolist()))
        if ka != kb:
            label += 1
            if int(tag(ra).sum()) != int(tag(rb).sum()) or len(ra) != len(rb):
                flip += 1
    return changed, label, flip


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reads", type=int, default=200_000)
    ap.add_argument("--configs", default="nbd96,dual,rbk96x")
    ap.add_argument("--backend", default="oracle", choices=["oracle", "hip", "both"])
    ap.add_argument("--sub", type=float, default=0.04)
    ap.add_argument("--ins", type=float, default=0.02)
    ap.add_argument("--del", dest="dele", type=float, default=0.02)
    ap.add_argument("--len-min", type=int, default=600)
    ap.add_argument("--len-max", type=int, default=4000)
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "policy_sensitivity.json"))
    args = ap.parse_args()
    nt = effective_cpus()
    from barbell_amd import annotate as A
    from oracle import pyoracle as po

    pols = [(hz, p) for hz, alts in ALTERNATIVES.items() for p in alts if p != "trace=SMID"]  # SMID is MSID's class
    res = {"reads_per_config": args.reads, "noise": {"substitution": args.sub, "insertion": args.ins, "deletion": args.dele, "where": "whole read"},
           "read_len": [args.len_min, args.len_max], "backend": args.backend, "hazards": HAZARD_TEXT, "configs": {}}
    for ci, cfg in enumerate(args.configs.split(",")):
        groups = config_groups(cfg)
        bases, offsets = A.synth_reads_host(groups, 0x5E451 + ci, args.len_min, args.len_max, 0, args.reads)
        bases, offsets = mutate(bases, offsets, args.sub, args.ins, args.dele, 17 + ci)
        n = args.reads

        def run(pol):
            out = {}
            if args.backend in ("oracle", "both"):
                out["oracle"] = po.Oracle([g.as_tuple() for g in groups], policy=pol).annotate(bases, offsets, n_threads=nt, fast=True)
            if args.backend in ("hip", "both"):
                dm = A.Demuxer(policy=pol)
                for g in groups:
                    dm.add_query_group(g)
                out["hip"] = dm.demux_packed(bases, offsets)
                dm.close()
            if args.backend == "both" and out["hip"].tobytes() != out["oracle"].tobytes():
                raise SystemExit(f"{cfg} {pol}: HIP rows differ from the checker's")
            return out["hip" if args.backend != "oracle" else "oracle"]

        t0 = time.time()
        base = run(None)
        ib = per_read(base, n)
        c = {"rows_default": int(len(base)), "reads_with_rows": int(((ib[1:] - ib[:-1]) > 0).sum()),
             "tag_rows_default": int((base["barcode_idx"] >= 0).sum()), "policies": {}}
        for hz, pol in pols:
            rows = run(pol)
            ch, lb, fl = compare(base, rows, n)
            e = {"hazard": hz, "rows": int(len(rows)), "reads_changed_pct": 100.0 * ch / n, "label_changed_pct": 100.0 * lb / n, "tag_flank_flip_pct": 100.0 * fl / n}
            p = _abi.policy_from_str(pol)
            if p.lodhi_p != 3 or p.lodhi_lambda != 0.5:
                e["pinned_by_barbell"] = "searcher.rs:209: Lodhi::new(3, 0.5)"
            if not is_feasible(pol):
                e["refuted_by_reference"] = "tests/golden/policy_feasible.json: the reference's own vectors / documented examples exclude this setting"
            c["policies"][pol] = e
            print(f"{cfg:7s} {pol:22s} reads changed {e['reads_changed_pct']:7.3f} %  label {e['label_changed_pct']:7.3f} %  tag<->flank {e['tag_flank_flip_pct']:7.3f} %", flush=True)
        # the worst cases range over what is really open: neither pinned by Barbell's own code nor refuted by its own vectors (tools/policy_feasible.py)
        open_ = {k: v for k, v in c["policies"].items() if "pinned_by_barbell" not in v and "refuted_by_reference" not in v}
        refuted = {k: v for k, v in c["policies"].items() if "refuted_by_reference" in v}
        c["worst_label_change_among_refuted"] = max(((v["label_changed_pct"], k) for k, v in refuted.items()), default=None)
        c["worst_label_change"] = max(((v["label_changed_pct"], k) for k, v in open_.items()))
        c["worst_per_hazard"] = {hz: max(((v["label_changed_pct"], k) for k, v in open_.items() if v["hazard"] == hz), default=None) for hz in HAZARD_TEXT}
        c["seconds"] = time.time() - t0
        if args.backend == "both":
            c["hip_equals_checker_under_every_policy"] = True
        res["configs"][cfg] = c
    res["worst_label_change_pct"] = max(c["worst_label_change"][0] for c in res["configs"].values())
    json.dump(res, open(args.out, "w"), indent=1)
    print(args.out, "worst label change", res["worst_label_change_pct"], "%")


if __name__ == "__main__":
    main()
