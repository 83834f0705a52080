#!/usr/bin/env python3
"""policy_feasible.py — which settings of the unpinned assumptions (include/barbell_amd_policy.h) the reference's OWN material still allows.

The policy space is what this repo does not know about sassy 0.2.1 / cigar-lodhi-rs 0.1.0 (hazards H1-H8).  Not all of it is
open: the reference holds vectors and invariants that exclude settings without any help from the real crates.  This tool computes that
set with the CPU checker (oracle/, test infrastructure) and writes it to tests/golden/policy_feasible.json; everything that ranges
"over every policy" (tools/policy_sensitivity.py, tools/ref_fit.py, bench.py's policy_variants, the default class list of the
kernel build, the parity statement in README / DESIGN §2) takes its list from that file.

Stage A — the five sassy known-answer tests of the reference (src/annotate/cigar_parse.rs:104-176), as written there: `matches.first()`
    of `Searcher::<Iupac>::new_rc().search(p, t, k)`, `map_pat_to_text_with_cost(m, 5, 8)` -> cost and text span, including the
    reverse-complemented half of :104-123.  Run under the FULL cross product lm x rc x trace (18 classes) x ovh x rcpath x tie.
Stage B — the reference's own no-panic invariants and its documented examples, on noisy reads with planted constructs of every shipped
    kit geometry (src/kits/kits.rs: nine contexts covering its twelve distinct flanks) and of the examples/*.fasta query sets (README of the reference:
    the custom dual-end experiment):
      * `expect("No barcode match region found")` (searcher.rs:388) must never fire, `read[ws..we]` (searcher.rs:456) must never panic:
        a setting under which real Barbell aborts on ordinary reads is not what the real crates do;
      * `get_matching_region` (cigar_parse.rs:71-82, searcher.rs:445-456) must put the barcode window ON the barcode: a setting under
        which real Barbell windows off the barcode of its own kits / its own documented example is not a candidate.  Measured: of the
        flank matches that lie on a planted construct (same group, same strand), the share whose window overlaps the planted barcode.
    Run over the joint settings of the fields that reach the window (lm x trace x ovh x rcpath) that survived stage A; `rc` (order of
    the returned Vec) and `tie` (which equally cheap barcode match is kept) do not reach it and are run field by field.
`lodhi` (H8) is not touched by any vector or invariant the reference holds: listed as unconstrained.

    python tools/policy_feasible.py                      # full run (~10 min on 8 cores) -> tests/golden/policy_feasible.json
    python tools/policy_feasible.py --reads 300 --geometries nbd96,dual --check    # what tests/test_policy_feasible.py does
"""
import argparse
import hashlib
import itertools
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from barbell_amd import _abi, kits  # noqa: E402
from barbell_amd.parallel import effective_cpus  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden", "policy_feasible.json")
EX = os.path.join(ROOT, "tests", "golden", "examples")

TRACE_CLASSES = ["MISD"] + sorted("".join(p) for p in itertools.permutations("MSID") if "SM" not in "".join(p) and "".join(p) != "MISD")
SPACE = {   # first value = default
    "lm": ["right", "left", "strict"],
    "rc": ["scan", "fwd"],
    "trace": TRACE_CLASSES,
    "ovh": ["floor", "ceil", "near", "floor:f64", "ceil:f64", "near:f64"],
    "rcpath": ["fwd", "mirror"],
    "tie": ["first", "last"],
}
KEYS = list(SPACE)
WINDOW_FIELDS = ["lm", "trace", "ovh", "rcpath"]      # the fields that reach get_matching_region's input
ORDER_FIELDS = ["rc", "tie"]                          # order of the Vec / which equal-cost barcode match: cannot move a window
MIN_WINDOW_ON_BARCODE = 0.99                          # stage B: share of on-target flank matches whose window overlaps the planted barcode

# cigar_parse.rs:104-176 as data: (pattern, text, k, expected cost, expected text span or None); `rc=True`: the vector's second half
P = b"AAAAACCCAAAA"
KATS = [
    ("no_edits", P, b"GGGGAAAAACCCAAAAGGGGG", 0, 0, None, False),
    ("no_edits_rc", P, b"GGGGAAAAACCCAAAAGGGGG", 0, 0, None, True),
    ("1_edits", P, b"GGGGAAAAACGCAAAA", 1, 1, None, False),
    ("overhang_left_flank", P, b"ACGCAAAAGGGGGGGGGGGG", 5, 1, (1, 4), False),
    ("overhang_right_flank", P, b"GAAAAACGC", 5, 1, (6, 9), False),
    ("overhang_including_bar", P, b"GCAAAAGGGGGGGGGGGG", 8, 2, (0, 2), False),
]


def revcomp_kat(s):   # cigar_parse.rs:90-102
    t = {65: 84, 84: 65, 67: 71, 71: 67}
    return bytes(t.get(c, 78) for c in reversed(s))


def text_of(setting):
    return ",".join(f"{k}={setting[k]}" for k in KEYS if k in setting)


def default_setting():
    return {k: SPACE[k][0] for k in KEYS}


# ---------------------------------------------------------------------------------------------------------------------
def stage_a():
    """every joint setting against the reference's KATs -> {setting text: [names of the KATs it breaks]} for the refuted ones, and the survivors"""
    from oracle import pyoracle as po

    refuted, survivors = {}, []
    for combo in itertools.product(*(SPACE[k] for k in KEYS)):
        s = dict(zip(KEYS, combo))
        broken = []
        with po.policy(text_of(s)):
            for name, pat, text, k, cost, span, rc in KATS:
                p, t = (revcomp_kat(pat), revcomp_kat(text)) if rc else (pat, text)
                ms, h = po.search(p, t, k, alpha=None, rc=True)
                got = po.map_pat_to_text_with_cost(h, 0, 5, 8) if ms else None
                po.free_matches(h)
                if got is None or got[2] != cost or (span is not None and got[1] != span):
                    broken.append(name)
        if broken:
            refuted[text_of(s)] = broken
        else:
            survivors.append(s)
    return refuted, survivors


def factorise(settings, keys):
    """the per-field value sets of a list of settings, and whether the list is exactly their product"""
    vals = {k: [v for v in SPACE[k] if any(s[k] == v for s in settings)] for k in keys}
    n = 1
    for k in keys:
        n *= len(vals[k])
    return vals, n == len({tuple(s[k] for k in keys) for s in settings})


# ---------------------------------------------------------------------------------------------------------------------
def geometries(which=None):
    """name -> list[QueryGroup]: one context per distinct flank geometry of the shipped kits (largest barcode set of each), the custom dual-end
    experiment of the reference's examples (native_left + native_right, --flank-max-errors 5 as BASELINE configs[3]) and the two single-file
    example query sets"""
    out, seen = {}, set()
    from oracle import pyoracle as po

    names = {"SQK-NBD114-96": "nbd96", "SQK-RBK114-96": "rbk96"}
    for kit in sorted(kits.supported_kits(), key=lambda k: (k not in names, k)):
        for ext in (False, True):
            gs = kits.groups_from_kit(kit, use_extended=ext)
            o = po.Oracle([g.as_tuple() for g in gs])
            key = tuple((o.flank(i), g.match_type) for i, g in enumerate(gs))
            o.close()
            if key in seen or (ext and len(gs) == len(kits.groups_from_kit(kit))):
                continue
            seen.add(key)
            out[names.get(kit, kit) + ("x" if ext else "")] = gs
    out["dual"] = [kits.group_from_fasta(os.path.join(EX, "native_left.fasta"), _abi.BB_FTAG, 5),
                   kits.group_from_fasta(os.path.join(EX, "native_right.fasta"), _abi.BB_RTAG, 5)]
    out["native_bars"] = [kits.group_from_fasta(os.path.join(EX, "native_bars.fasta"), _abi.BB_FTAG)]
    out["rapid_bars"] = [kits.group_from_fasta(os.path.join(EX, "rapid_bars.fasta"), _abi.BB_FTAG)]
    if which:
        out = {k: v for k, v in out.items() if k in which}
    return out


COMP = np.zeros(256, dtype=np.uint8)
for a, b in zip(b"ACGTRYSWKMBDHVN", b"TGCAYRSWMKVHDBN"):
    COMP[a] = b


def planted_reads(groups, n, seed, sub=0.04, ins=0.02, dele=0.02):
    """n reads with one construct of a group near each end (random barcode, random strand; a twentieth of the 5' ones cut by 1-20 nt so
    that the flank hangs over the read's start), sequencing-like noise over the whole read; -> bases, offsets, truth[n, 2, 7] in the
    coordinates of the noisy reads"""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from oracle import pyoracle as po
    from policy_sensitivity import mutate

    rng = np.random.default_rng(seed)
    o = po.Oracle([g.as_tuple() for g in groups])
    info = [o.info(i) for i in range(len(groups))]
    o.close()
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    parts, truth, pos = [], np.full((n, po.TRUTH_PER_READ, po.TRUTH_FIELDS), -1, dtype=np.int32), 0
    offsets = np.zeros(n + 1, dtype=np.uint64)
    for r in range(n):
        start = pos
        for slot in range(2):
            gi = (r + slot) % len(groups)
            g, inf = groups[gi], info[gi]
            idx = int(rng.integers(len(g.seqs)))
            strand = int(rng.integers(2))
            cons = np.frombuffer(g.seqs[idx], dtype=np.uint8)
            b_lo, b_hi = inf.prefix_len, inf.prefix_len + inf.mask_len
            if strand:
                cons = COMP[cons[::-1]]
                b_lo, b_hi = len(cons) - b_hi, len(cons) - b_lo
            cut = int(rng.integers(1, 21)) if slot == 0 and rng.random() < 0.05 else 0
            if slot == 0:
                lead = 0 if cut else int(rng.integers(0, 61))
            else:
                lead = int(rng.integers(150, 500))          # the read's body
            body = acgt[rng.integers(0, 4, lead)]
            parts.append(body)
            pos += lead
            c_lo = pos
            parts.append(cons[cut:])
            pos += len(cons) - cut
            truth[r, slot] = (gi, strand, c_lo, pos, max(c_lo, c_lo + b_lo - cut), c_lo + b_hi - cut, idx)
        tail = int(rng.integers(0, 61))
        parts.append(acgt[rng.integers(0, 4, tail)])
        pos += tail
        offsets[r] = start
    offsets[n] = pos
    bases = np.concatenate(parts)
    # noise; the truth coordinates go through the same position map as the offsets
    nb, noff, csum = mutate(bases, offsets, sub, ins, dele, seed + 1, return_map=True)
    for f in (2, 3, 4, 5):
        truth[:, :, f] = (csum[truth[:, :, f]] - noff[:-1, None].astype(np.int64)).astype(np.int32)
    return nb, noff, truth


def stage_b(settings, geos, n_reads, seed, nt, log=print):
    """counters of bbo_annotate_diag per geometry and setting"""
    from oracle import pyoracle as po

    res = {}
    for gname, groups in geos.items():
        bases, offsets, truth = planted_reads(groups, n_reads, seed)
        t0 = time.time()
        per = {}
        for s in settings:
            o = po.Oracle([g.as_tuple() for g in groups], policy=text_of(s))
            per[text_of(s)] = o.annotate_diag(bases, offsets, truth, n_threads=nt, fast=True)
            o.close()
        res[gname] = per
        d = per[text_of(settings[0])]
        log(f"  {gname:16s} {len(settings):4d} settings in {time.time() - t0:6.1f} s; first setting: {d['on_target']} on-target flank matches, "
            f"window on barcode {d['window_overlaps'] / max(1, d['on_target']):.4f}, tags correct {d['tag_rows_correct']}/{d['tag_rows_on_target']}")
    return res


def verdict_b(d):
    """why a setting's counters on one geometry refute it (None = they do not)"""
    if d["subpath_none"]:
        return f"searcher.rs:388 expect would fire on {d['subpath_none']} matches"
    if d["slice_panic"]:
        return f"searcher.rs:456 slice would panic on {d['slice_panic']} matches"
    if d["on_target"] and d["window_overlaps"] < MIN_WINDOW_ON_BARCODE * d["on_target"]:
        return f"barcode window off the planted barcode for {100.0 * (1 - d['window_overlaps'] / d['on_target']):.1f} % of the on-target flank matches"
    return None


def inputs_digest():
    """what the result depends on besides the checker: the vectors, the space, the kit tables and example queries, this file's parameters"""
    h = hashlib.sha256()
    h.update(repr((KATS, SPACE, WINDOW_FIELDS, MIN_WINDOW_ON_BARCODE)).encode())
    for p in [os.path.join(ROOT, "barbell_amd", "data", "kits.json")] + [os.path.join(EX, f) for f in sorted(os.listdir(EX))]:
        h.update(open(p, "rb").read())
    return h.hexdigest()[:16]


def compute(n_reads, which, seed, nt, log=print):
    t0 = time.time()
    refuted_a, surv_a = stage_a()
    vals_a, prod_a = factorise(surv_a, KEYS)
    log(f"stage A: {len(surv_a)} of {len(refuted_a) + len(surv_a)} joint settings reproduce cigar_parse.rs:104-176 ({time.time() - t0:.1f} s); per field: "
        + ", ".join(f"{k} {len(vals_a[k])}/{len(SPACE[k])}" for k in KEYS) + ("" if prod_a else "  [NOT a product of per-field sets]"))
    by_kat = {}
    for s, broken in refuted_a.items():
        for b in broken:
            by_kat[b] = by_kat.get(b, 0) + 1
    # which single fields, changed alone from the default, break a KAT
    single_a = {}
    for k in KEYS:
        for v in SPACE[k][1:]:
            s = dict(default_setting(), **{k: v})
            if text_of(s) in refuted_a:
                single_a[f"{k}={v}"] = refuted_a[text_of(s)]
    # stage B: joint over the window fields among stage A's survivors, rc / tie field by field
    geos = geometries(which)
    win_settings = []
    for combo in itertools.product(*(vals_a[k] for k in WINDOW_FIELDS)):
        s = dict(default_setting(), **dict(zip(WINDOW_FIELDS, combo)))
        if any(all(t[k] == s[k] for k in KEYS) for t in surv_a):
            win_settings.append(s)
    order_settings = [dict(default_setting(), **{k: v}) for k in ORDER_FIELDS for v in vals_a[k][1:]]
    log(f"stage B: {len(win_settings)} joint settings of {WINDOW_FIELDS} + {len(order_settings)} single-field settings of {ORDER_FIELDS} on {len(geos)} geometries x {n_reads} reads")
    res_b = stage_b(win_settings + order_settings, geos, n_reads, seed, nt, log)
    refuted_b = {}
    for s in win_settings + order_settings:
        why = {g: verdict_b(res_b[g][text_of(s)]) for g in geos}
        why = {g: w for g, w in why.items() if w}
        if why:
            refuted_b[text_of(s)] = why
    surv_win = [s for s in win_settings if text_of(s) not in refuted_b]
    surv_order = {k: [SPACE[k][0]] + [v for v in vals_a[k][1:] if text_of(dict(default_setting(), **{k: v})) not in refuted_b] for k in ORDER_FIELDS}
    vals_b, prod_b = factorise(surv_win, WINDOW_FIELDS)
    feasible = dict(vals_b, **surv_order)
    joint = [dict(s, **dict(zip(ORDER_FIELDS, o))) for s in surv_win for o in itertools.product(*(surv_order[k] for k in ORDER_FIELDS))]
    joint = [s for s in joint if any(all(t[k] == s[k] for k in KEYS) for t in surv_a)]
    n_total = 1
    for k in KEYS:
        n_total *= len(SPACE[k])
    # per (geometry, setting) summary kept in the file: the window share and the counters that decide
    table = {g: {s: {"on_target": d["on_target"], "window_on_barcode": round(d["window_overlaps"] / max(1, d["on_target"]), 5),
                     "window_covers_barcode": round(d["window_covers"] / max(1, d["on_target"]), 5),
                     "tags_correct": round(d["tag_rows_correct"] / max(1, d["tag_rows_on_target"]), 5), "tag_rows": d["tag_rows_on_target"],
                     "region_none": d["region_none"], "slice_panic": d["slice_panic"], "subpath_none": d["subpath_none"]}
                 for s, d in per.items()} for g, per in res_b.items()}
    worst = {}   # each value of a window field alone (the other fields at their defaults): lowest share over the geometries, and where
    for k in WINDOW_FIELDS:
        for v in vals_a[k]:
            t = text_of(dict(default_setting(), **{k: v}))
            worst[f"{k}={v}"] = min((table[g][t]["window_on_barcode"], g) for g in geos) if all(t in table[g] for g in geos) else None
    out = {
        "_what": "settings of include/barbell_amd_policy.h that the reference's own vectors (cigar_parse.rs:104-176) and invariants (searcher.rs:388, :445-456 on "
                 "its kits and examples) still allow; made by tools/policy_feasible.py with the CPU checker; consumed by tools/policy_sensitivity.py, "
                 "tools/ref_fit.py, bench.py (policy_variants), barbell_amd/csrc/Makefile (CLASSES) and tests/test_policy_feasible.py",
        "inputs_digest": inputs_digest(),
        "space": SPACE, "n_joint_total": n_total,
        "stage_a": {"vectors": [k[0] for k in KATS], "n_survive": len(surv_a), "n_refuted": len(refuted_a), "survivors_per_field": vals_a,
                    "survivors_are_product_of_fields": prod_a, "refuted_single_field": single_a, "joint_settings_breaking_each_vector": by_kat},
        "stage_b": {"reads_per_geometry": n_reads, "seed": seed, "geometries": {g: [len(x.seqs) for x in gs] for g, gs in geos.items()},
                    "noise": {"substitution": 0.04, "insertion": 0.02, "deletion": 0.02}, "min_window_on_barcode": MIN_WINDOW_ON_BARCODE,
                    "window_fields": WINDOW_FIELDS, "order_fields": ORDER_FIELDS, "n_settings": len(win_settings) + len(order_settings),
                    "refuted": refuted_b, "worst_window_on_barcode_per_value": worst, "survivors_are_product_of_fields": prod_b,
                    "default": {g: table[g][text_of(default_setting())] for g in geos}},
        "feasible": feasible,
        "feasible_trace_class_indices": sorted(build_class_order().index(c) for c in feasible["trace"]),   # barbell_amd/csrc/Makefile: default CLASSES
        "n_joint_feasible": len(joint),
        "feasible_is_product_of_fields": bool(prod_a and prod_b),
        "unconstrained": {"lodhi": "H8: no vector or invariant the reference holds reaches Lodhi::compute's formula (p and lambda are pinned by searcher.rs:209)"},
        "default": text_of(default_setting()),
        "default_feasible": any(all(s[k] == SPACE[k][0] for k in KEYS) for s in joint),
        # per geometry: the default and every setting that differs from it in ONE field (the joint settings' verdicts are in stage_b.refuted)
        "table": {g: {s: v for s, v in per.items() if sum(a != b for a, b in zip(s.split(","), text_of(default_setting()).split(","))) <= 1} for g, per in table.items()},
    }
    if not out["feasible_is_product_of_fields"]:
        out["feasible_joint"] = [text_of(s) for s in joint]
    log(f"feasible: {len(joint)} of {n_total} joint settings; per field: " + ", ".join(f"{k}: {' '.join(feasible[k])}" for k in KEYS))
    return out


# ---- what the consumers call -------------------------------------------------------------------------------------------
def load(path=OUT):
    with open(path) as f:
        return json.load(f)


def feasible_values(field, path=OUT):
    return load(path)["feasible"][field]


def is_feasible(policy_text, path=OUT):
    """a (possibly partial) policy text: every field it names holds a feasible value (trace by class; lodhi unconstrained)"""
    f = load(path)
    p = _abi.policy_from_str(policy_text)
    full = dict(tok.split("=", 1) for tok in _abi.policy_to_str(p).split(","))
    full["trace"] = full["trace"].replace("SM", "MS")
    if not f["feasible_is_product_of_fields"]:
        return ",".join(f"{k}={full[k]}" for k in KEYS) in f["feasible_joint"]
    return all(full[k] in f["feasible"][k] for k in KEYS)


def build_class_order():
    """the classes in the order of barbell_amd/csrc/bb_prio.h's BB_PRIO_TABLE (what -DBB_TU_CLASS=<i> and the Makefile's CLASSES index): the
    default first, then the canonical orders by ascending packed value (first choice in bits 0-1; M=0 S=1 I=2 D=3)"""
    pack = lambda o: sum("MSID".index(c) << (2 * i) for i, c in enumerate(o))
    return ["MISD"] + sorted((c for c in TRACE_CLASSES if c != "MISD"), key=pack)


def feasible_class_indices(path=OUT):
    """positions of the feasible traceback classes in the build's class list"""
    order = build_class_order()
    return sorted(order.index(c) for c in feasible_values("trace", path))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reads", type=int, default=10_000, help="planted reads per geometry (stage B)")
    ap.add_argument("--geometries", default="", help="comma-separated subset (default: every kit geometry + the examples)")
    ap.add_argument("--seed", type=int, default=0xFEA51B)
    ap.add_argument("--out", default=OUT)
    ap.add_argument("--check", action="store_true", help="compute and compare with --out instead of writing: stage A exactly, stage B's verdicts on the geometries run")
    ap.add_argument("--classes", action="store_true", help="print the feasible classes' indices (the Makefile's default CLASSES) and exit")
    args = ap.parse_args()
    if args.classes:
        print(" ".join(str(i) for i in feasible_class_indices(args.out)))
        return
    which = [g for g in args.geometries.split(",") if g] or None
    res = compute(args.reads, which, args.seed, effective_cpus())
    if args.check:
        old = load(args.out)
        bad = [k for k in ("inputs_digest", "space", "n_joint_total", "default", "default_feasible") if old[k] != res[k]]
        bad += ["stage_a." + k for k in ("n_survive", "survivors_per_field", "refuted_single_field") if old["stage_a"][k] != res["stage_a"][k]]
        # the verdicts of a smaller run must not contradict the committed ones: what the file calls feasible stays unrefuted here
        for s, why in res["stage_b"]["refuted"].items():
            if is_feasible(s, args.out):
                bad.append(f"stage_b: {s} refuted here ({why}) but feasible in {args.out}")
        if bad:
            sys.exit("policy_feasible.json is stale or wrong: " + "; ".join(bad))
        print("policy_feasible.json agrees")
        return
    with open(args.out, "w") as f:
        json.dump(res, f, indent=1, sort_keys=False)
    print(args.out)


if __name__ == "__main__":
    main()
