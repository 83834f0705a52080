#!/usr/bin/env python3
"""ref_fit.py — which policy (include/barbell_amd_policy.h) makes this repo reproduce the real crates / real Barbell?

The assumptions about sassy 0.2.1 and cigar-lodhi-rs 0.1.0 that Barbell's own code does not pin (hazards
H1-H4, H7, H8) are fields of `bb_policy`, honoured by the HIP kernels and the CPU checker alike.  Given answers of the
real thing this tool searches the policy space with the CPU checker (oracle/, test infrastructure — never the product)
and reports the policy that reproduces them, or how close the best one comes and what is left unexplained:

  tools/ref_fit.py kat  tests/golden/ref_kat.jsonl          answers of tools/ref_golden/kat.rs (Lodhi::compute, Searcher::search,
                                                            search_encoded_patterns on tools/ref_golden/kat_inputs.tsv)
  tools/ref_fit.py tsv  EXPORT_DIR/<config> [--reads N]     a ref.tsv made by real `barbell annotate` on a read set of
                                                            tools/ref_export.py (also: tools/ref_diff.py --fit)

The winning policy goes into BARBELL_AMD_POLICY / `barbell-amd --policy` / bb_create_policy(), and — once it is backed by
committed vectors — into bb_policy_default() and tests/golden/policy.txt, which the golden-vector tests apply.
"""
import argparse
import itertools
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools", "ref_golden"))
from barbell_amd.parallel import effective_cpus  # noqa: E402

OPS = {"=": 0, "X": 1, "I": 2, "D": 3}

# the space per field of the text form (first value = default) ...
FULL_SPACE = {
    "lm": ["right", "left", "strict"],
    "rc": ["scan", "fwd"],
    "trace": ["MISD"] + ["".join(p) for p in itertools.permutations("MSID") if "".join(p) != "MISD"],
    "ovh": ["floor", "ceil", "near", "floor:f64", "ceil:f64", "near:f64"],
    "rcpath": ["fwd", "mirror"],
    "tie": ["first", "last"],
    "lodhi": ["3:0.5:1111"] + [f"{p}:0.5:{a}{b}{c}{d}" for p in (3, 2, 4) for a in (1, 2, 0) for b in (1, 2, 0) for c in (1, 2, 0) for d in (1, 2, 0)
                               if (p, a, b, c, d) != (3, 1, 1, 1, 1)],
}


def _feasible_space():
    """... cut to what the reference's own vectors and invariants leave open (tests/golden/policy_feasible.json, tools/policy_feasible.py:
    cigar_parse.rs:163-176 refutes 13 of the 18 traceback classes, the documented dual-end example refutes rcpath=mirror): there is no point
    in asking whether the real crates do what the reference's own tests say they do not.  `--all` searches the refuted values too."""
    with open(os.path.join(ROOT, "tests", "golden", "policy_feasible.json")) as f:
        feas = json.load(f)["feasible"]
    space, refuted = {}, {}
    for k, vals in FULL_SPACE.items():
        if k not in feas:
            space[k] = list(vals)
            continue
        ok = lambda v: (v.replace("SM", "MS") if k == "trace" else v) in feas[k]
        space[k] = [v for v in vals if ok(v)]
        refuted[k] = [v for v in vals if not ok(v)]
    return space, refuted


SPACE, REFUTED = _feasible_space()
KEYS = list(SPACE)


def to_text(pol):
    return ",".join(f"{k}={pol[k]}" for k in KEYS)


def default_policy():
    return {k: SPACE[k][0] for k in KEYS}


# ---- answers of the CPU checker in kat.rs's output format ----------------------------------------------------------
def checker_answer(inp, pol_text):
    """one input of kat_inputs.tsv -> the dict kat.rs would print for it, computed by the CPU checker under a policy"""
    from oracle import pyoracle as po

    with po.policy(pol_text):
        if inp[0] == "lodhi":
            import numpy as np

            v = po.lodhi([OPS[c] for c in inp[1]])
            return {"kind": "lodhi", "ops": inp[1], "score": v, "bits": "%#018x" % int(np.float64(v).view(np.uint64))}
        if inp[0] == "search":
            _, alpha, k, pat, text = inp
            ms, h = po.search(pat.encode(), text.encode(), k, alpha=None if alpha == -1 else alpha, rc=True)
            po.free_matches(h)
            return {"kind": "search", "searcher": "rc" if alpha == -1 else "rc_overhang", "alpha": alpha, "pattern": pat, "text": text, "k": k,
                    "matches": [{"text_start": m.text_start, "text_end": m.text_end, "pattern_start": m.pattern_start, "pattern_end": m.pattern_end,
                                 "cost": m.cost, "strand": "Rc" if m.strand else "Fwd", "pattern_idx": 0, "ops": m.cigar,
                                 "path": [list(q) for q in m.path]} for m in ms]}
        _, k, text, pats = inp
        allm = []
        for idx, p in enumerate(pats):
            ms, h = po.search(p.encode(), text.encode(), k, alpha=None, rc=False)
            po.free_matches(h)
            allm += [{"text_start": m.text_start, "text_end": m.text_end, "pattern_start": m.pattern_start, "pattern_end": m.pattern_end, "cost": m.cost,
                      "strand": "Fwd", "pattern_idx": idx, "ops": m.cigar, "path": [list(q) for q in m.path]} for m in ms]
        return {"kind": "search_set", "patterns": pats, "text": text, "k": k, "matches": allm}


def input_of(v):
    if v["kind"] == "lodhi":
        return ("lodhi", v["ops"])
    if v["kind"] == "search":
        return ("search", -1 if v["alpha"] < 0 else float(v["alpha"]), v["k"], v["pattern"], v["text"])
    return ("search_set", v["k"], v["text"], v["patterns"])


def kept_per_pattern(matches, last=False):
    """what collect_candidates_for_region keeps of a match list (searcher.rs:294-300): per pattern the first strictly lowest
    cost (policy tie=last: our list is in ascending position, the crate's is taken to be descending)"""
    best = {}
    for m in (reversed(matches) if last else matches):
        b = best.get(m["pattern_idx"])
        if b is None or m["cost"] < b["cost"]:
            best[m["pattern_idx"]] = m
    return {i: (m["text_start"], m["text_end"], m["cost"], m["ops"]) for i, m in best.items()}


def same_answer(ref, ours, pol):
    """does the checker's answer explain the crate's?  lodhi: the f64 bits; search: the full match list incl. paths;
    search_set: what Barbell keeps of it"""
    if ref["kind"] == "lodhi":
        return int(ref["bits"], 16) == int(ours["bits"], 16)
    if ref["kind"] == "search":
        key = lambda m: (m["text_start"], m["text_end"], m["pattern_start"], m["pattern_end"], m["cost"], m["strand"], m["ops"], [list(p) for p in m["path"]])
        return [key(m) for m in ref["matches"]] == [key(m) for m in ours["matches"]]
    return kept_per_pattern(ref["matches"]) == kept_per_pattern(ours["matches"], last=pol["tie"] == "last")


def kat_score(vectors, pol, kinds=None):
    txt = to_text(pol)
    ok = []
    for v in vectors:
        if kinds and v["kind"] not in kinds:
            continue
        ok.append(same_answer(v, checker_answer(input_of(v), txt), pol))
    return ok


# which fields can influence which kind of vector: the three kinds are fitted independently, each exhaustively where that
# is cheap (lodhi: 242 settings) and by coordinate descent over its fields otherwise
FIELDS_OF = {"lodhi": ["lodhi"], "search": ["lm", "rc", "trace", "ovh", "rcpath"], "search_set": ["lm", "trace", "tie"]}


def descend(score_fn, pol, fields, log=None):
    """coordinate descent: change one field at a time while the score improves; returns (policy, score)"""
    best = score_fn(pol)
    improved = True
    while improved:
        improved = False
        for f in fields:
            for val in SPACE[f]:
                if val == pol[f]:
                    continue
                cand = dict(pol, **{f: val})
                sc = score_fn(cand)
                if sc > best:
                    best, pol, improved = sc, cand, True
                    if log:
                        log(f"  {f}={val}: {sc}")
    return pol, best


def fit_kat(vectors, log=None):
    pol = default_policy()
    report = {}
    for kind in ("lodhi", "search", "search_set"):
        vs = [v for v in vectors if v["kind"] == kind]
        if not vs:
            continue
        sf = lambda p, vs=vs: sum(kat_score(vs, p))
        if log:
            log(f"{kind}: {len(vs)} vectors, default explains {sf(pol)}")
        pol, sc = descend(sf, pol, FIELDS_OF[kind], log)
        # equally good alternatives of each field (the vectors do not tell them apart)
        ties = {f: [val for val in SPACE[f] if val != pol[f] and sf(dict(pol, **{f: val})) == sc] for f in FIELDS_OF[kind]}
        bad = [input_of(v) for v, ok in zip(vs, kat_score(vs, pol)) if not ok]
        report[kind] = {"vectors": len(vs), "explained": sc, "not_told_apart": {f: t for f, t in ties.items() if t}, "unexplained": bad[:5]}
    return pol, report


# ---- whole-path fit on a ref.tsv ----------------------------------------------------------------------------------------
def fit_tsv(export_dir, n_reads=2000, log=None, threads=None):
    import ref_diff
    import ref_export
    from barbell_amd import annotate as A
    from oracle import pyoracle as po

    man = json.load(open(os.path.join(export_dir, "manifest.json")))
    groups = ref_export.config_groups(man["config"])
    ids, seqs = [], []
    for rid, s in A.read_fastq(os.path.join(export_dir, "reads.fastq")):
        ids.append(rid)
        seqs.append(s)
        if len(ids) >= n_reads:
            break
    keep = set(ids)
    ref_rows = [r for r in ref_diff.parse_tsv(os.path.join(export_dir, "ref.tsv")) if r["read_id"] in keep]
    threads = threads or effective_cpus()

    def rows_under(pol):
        rows = po.Oracle([g.as_tuple() for g in groups], policy=to_text(pol)).annotate_reads(seqs, n_threads=threads)
        lines = A.format_rows(rows, ids, groups)
        return [dict(zip(ref_diff.COLS, l.split("\t"))) for l in lines]

    def score(pol):
        return ref_diff.diff_rows(ref_rows, rows_under(pol), ids, max_examples=0)["reads_equal"]

    pol = default_policy()
    if log:
        log(f"{len(ids)} reads, {len(ref_rows)} reference rows; default policy: {score(pol)} reads identical")
    pol, sc = descend(score, pol, KEYS, log)
    rep = ref_diff.diff_rows(ref_rows, rows_under(pol), ids)
    return pol, {"reads": len(ids), "reads_identical": sc, "buckets": rep["buckets"], "hazards": rep["hazards"], "examples": rep["examples"][:3]}


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("mode", choices=["kat", "tsv"])
    ap.add_argument("path")
    ap.add_argument("--reads", type=int, default=2000)
    ap.add_argument("--all", action="store_true", help="search the values the reference's own vectors refute as well (tests/golden/policy_feasible.json)")
    a = ap.parse_args()
    if a.all:
        SPACE.update({k: list(v) for k, v in FULL_SPACE.items()})
    log = lambda s: print(s, file=sys.stderr)
    if a.mode == "kat":
        pol, rep = fit_kat([json.loads(l) for l in open(a.path) if l.strip()], log)
        done = all(r["explained"] == r["vectors"] for r in rep.values())
    else:
        pol, rep = fit_tsv(a.path, a.reads, log)
        done = rep["reads_identical"] == rep["reads"]
    print(json.dumps({"policy": to_text(pol), "explains_everything": done, "report": rep, "searched": {k: len(v) for k, v in SPACE.items()},
                      "not_searched_refuted_by_the_reference": {} if a.all else {k: v for k, v in REFUTED.items() if v}}, indent=1))
    return 0 if done else 1


if __name__ == "__main__":
    sys.exit(main())
