"""The reference-parity hook: tools/ref_export.py + tools/ref_diff.py exercised against a
fake `barbell` that replays the oracle's TSV (in worker-batch order, like the real one writes it), and the ingestion
of golden vectors produced off-box by tools/ref_golden/kat.rs / real Barbell — skipped with
"reference parity unpinned beyond KATs" while those files are absent."""
import json
import os
import stat
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, os.path.join(ROOT, "tools", "ref_golden"))

import ref_diff  # noqa: E402
import ref_export  # noqa: E402

from barbell_amd import annotate as A  # noqa: E402
from barbell_amd.parallel import effective_cpus  # noqa: E402
from oracle import pyoracle as po  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
UNPINNED = "reference parity unpinned beyond KATs"


def golden_policy():
    """tests/golden/policy.txt: the policy (include/barbell_amd_policy.h) tools/ref_fit.py found to reproduce the committed golden
    vectors; absent = the default.  Goes in together with ref_kat.jsonl / ref_tsv/."""
    p = os.path.join(GOLD, "policy.txt")
    return open(p).read().strip() if os.path.exists(p) else None


def oracle_tsv(export_dir, cfg, path, policy=None):
    """the oracle's annotation.tsv of an exported read set (rows in input order)"""
    groups = ref_export.config_groups(cfg)
    ids, seqs = [], []
    for rid, s in A.read_fastq(os.path.join(export_dir, "reads.fastq")):
        ids.append(rid)
        seqs.append(s)
    rows = po.Oracle([g.as_tuple() for g in groups], policy=policy).annotate_reads(seqs, n_threads=effective_cpus())
    lines = A.format_rows(rows, ids, groups)
    with open(path, "w") as f:
        if lines:
            f.write(A.TSV_HEADER + "\n" + "\n".join(lines) + "\n")
    return ids, lines


FAKE = """#!/usr/bin/env python3
# stands in for real `barbell annotate`: replays a prepared TSV, rows of one read together, reads in a scrambled
# (worker-batch-like) order
import sys
a = sys.argv[1:]
assert a[0] == "annotate", a
out = a[a.index("-o") + 1]
assert a[a.index("-i") + 1] == "reads.fastq" and "-t" in a
want = %r
assert a[1:1 + len(want)] == want and len(a) == 1 + len(want) + 6, a
lines = open(%r).read().split("\\n")
head, body = lines[0], [l for l in lines[1:] if l]
groups = {}
for l in body:
    groups.setdefault(l.split("\\t")[0], []).append(l)
keys = list(groups)
keys = keys[1::2] + keys[0::2]
with open(out, "w") as f:
    if body:
        f.write(head + "\\n")
        for k in keys:
            f.write("\\n".join(groups[k]) + "\\n")
"""


@pytest.mark.parametrize("cfg", ["nbd96", "dual"])
def test_export_and_diff_with_fake_barbell(tmp_path, cfg):
    d = ref_export.export_config(cfg, str(tmp_path), 150)
    man = json.load(open(os.path.join(d, "manifest.json")))
    assert man["n_reads"] == 150 and man["barbell_args"][0] in ("--kit", "-q")
    replay = str(tmp_path / "replay.tsv")
    ids, lines = oracle_tsv(d, cfg, replay)
    assert len(ids) == 150 and len(lines) > 100
    fake = tmp_path / "barbell"
    fake.write_text(FAKE % (man["barbell_args"], replay))
    fake.chmod(fake.stat().st_mode | stat.S_IXUSR)
    # no binary, no cached ref.tsv -> the unpinned verdict
    rep, secs = ref_diff.reference_check(d, barbell=str(tmp_path / "nope"), ours_tsv=replay)
    assert rep["reference_parity"] == ref_diff.UNPINNED and secs is None
    # the fake reference replays the same rows in scrambled read order: identical after re-sorting
    rep, secs = ref_diff.reference_check(d, barbell=str(fake), ours_tsv=replay, threads=3)
    assert rep["identical"] and rep["reference_parity"] == "identical" and rep["reads_differ"] == 0
    assert secs is not None and rep["reference_threads"] == 3 and rep["rows_ref"] == len(lines)
    # perturbed "ours": one label, one flank coordinate, one dropped row, one swapped pair -> the right buckets
    rows = [l.split("\t") for l in lines]
    by_read = {}
    for i, r in enumerate(rows):
        by_read.setdefault(r[0], []).append(i)
    tag_rows = [i for i, r in enumerate(rows) if r[9] in ("Ftag", "Rtag")]
    multi = [v for v in by_read.values() if len(v) >= 2]
    used = set()

    def pick(cands):
        for i in cands:
            if rows[i][0] not in used:
                used.add(rows[i][0])
                return i
        raise AssertionError("no candidate")

    a = pick(tag_rows); rows[a][12] = "NOT_A_BARCODE"
    b = pick(tag_rows); rows[b][5] = str(int(rows[b][5]) + 1)
    c = pick(tag_rows); rows[c][11] = str(int(rows[c][11]) + 1)
    drop = pick(range(len(rows)))
    sw = next(v for v in multi if rows[v[0]][0] not in used and rows[v[0]] != rows[v[1]])
    rows[sw[0]], rows[sw[1]] = rows[sw[1]], rows[sw[0]]
    bad = str(tmp_path / "bad.tsv")
    with open(bad, "w") as f:
        f.write(A.TSV_HEADER + "\n" + "\n".join("\t".join(r) for i, r in enumerate(rows) if i != drop) + "\n")
    rep, _ = ref_diff.reference_check(d, barbell=str(fake), ours_tsv=bad)
    assert rep["reference_parity"] == "differs" and rep["reads_differ"] == 5
    assert rep["buckets"] == {"row_count": 1, "flank": 1, "label": 1, "bar": 1, "strand": 0, "order": 1}
    assert rep["hazards"]["label"] == "H7,H8" and len(rep["examples"]) == 5
    # command line entry point: exit code 1 on differences, 0 when identical, 2 without any reference
    import subprocess

    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "ref_diff.py"), d, "--ours", bad], capture_output=True, text=True)
    assert r.returncode == 1 and json.loads(r.stdout)["reads_differ"] == 5
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "ref_diff.py"), d, "--ours", replay], capture_output=True, text=True)
    assert r.returncode == 0
    os.remove(os.path.join(d, "ref.tsv"))
    env = dict(os.environ, PATH="/usr/bin:/bin")
    env.pop("BARBELL_BIN", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "ref_diff.py"), d, "--ours", replay], capture_output=True, text=True, env=env)
    assert r.returncode == 2 and ref_diff.UNPINNED in r.stderr


def test_export_is_deterministic(tmp_path):
    a = ref_export.export_config("rbk24", str(tmp_path / "a"), 40)
    b = ref_export.export_config("rbk24", str(tmp_path / "b"), 40)
    fa, fb = open(os.path.join(a, "reads.fastq"), "rb").read(), open(os.path.join(b, "reads.fastq"), "rb").read()
    assert fa == fb and fa.count(b"\n") == 160
    lens = [len(l) for l in fa.split(b"\n")[1::4]]
    assert min(lens) >= 600 and max(lens) < 4000  # C1: U[600, 4000)


# ---- golden vectors produced off-box (absent until someone with a Rust toolchain runs tools/ref_golden) ----------
OPS = {"=": 0, "X": 1, "I": 2, "D": 3}


def _match_tuple(m):
    return (m.text_start, m.text_end, m.pattern_start, m.pattern_end, m.cost, m.strand, m.cigar, [list(p) for p in m.path])


def test_ref_kat_vectors():
    path = os.path.join(GOLD, "ref_kat.jsonl")
    if not os.path.exists(path):
        pytest.skip(UNPINNED + " (tests/golden/ref_kat.jsonl absent: see tools/ref_golden/README.md)")
    assert check_kat_file(path, golden_policy()) > 0


def test_kat_inputs_discriminate_every_alternative():
    """tools/ref_golden/kat_inputs.tsv (what kat.rs feeds the real crates) tells every setting of every policy field apart:
    under the CPU checker no two settings of a field give the same answers on all inputs — except traceback orders that
    differ only by swapping Match and Sub while adjacent, which can never differ (Match needs a zero diagonal step, Sub a
    non-zero one).  So the crates' answers single out one policy."""
    import gen_kat_inputs as G
    import ref_fit

    inputs = G.load()
    assert [G.format_line(t) for t in G.all_inputs()] == [G.format_line(t) for t in inputs], "kat_inputs.tsv is stale: run gen_kat_inputs.py"
    d = ref_fit.default_policy()

    def sig(pol, kinds):
        out = []
        for inp in inputs:
            if inp[0] in kinds:
                a = ref_fit.checker_answer(inp, ref_fit.to_text(pol))
                out.append(sorted(ref_fit.kept_per_pattern(a["matches"], last=pol["tie"] == "last").items()) if a["kind"] == "search_set"
                           else a["bits"] if a["kind"] == "lodhi" else a["matches"])
        return json.dumps(out)

    # over the FULL space, refuted values included (tests/golden/policy_feasible.json cuts ref_fit.SPACE to what the reference's own vectors allow)
    for field, kinds in (("lm", ("search", "search_set")), ("rc", ("search",)), ("ovh", ("search",)), ("tie", ("search_set",)), ("lodhi", ("lodhi",)),
                         ("rcpath", ("search",))):
        sigs = {sig(dict(d, **{field: v}), kinds) for v in ref_fit.FULL_SPACE[field]}
        assert len(sigs) == len(ref_fit.FULL_SPACE[field]), field
    groups = {}
    for v in ref_fit.FULL_SPACE["trace"]:
        groups.setdefault(sig(dict(d, trace=v), ("search", "search_set")), []).append(v)
    for g in groups.values():
        assert len(g) == 1 or (len(g) == 2 and g[0].replace("MS", "SM") == g[1].replace("MS", "SM")), g
    assert len(groups) == 18


def test_fit_recovers_a_planted_policy_from_kat_vectors(tmp_path):
    """answers written by the checker under a non-default policy stand in for the crates': tools/ref_fit.py names a policy that
    explains all of them, equal to the planted one in every field"""
    import gen_kat_inputs as G
    import ref_fit

    planted = "lm=left,rc=fwd,trace=MSDI,ovh=near:f64,rcpath=fwd,tie=last,lodhi=3:0.5:2211"   # a feasible one (tests/golden/policy_feasible.json)
    vectors = [ref_fit.checker_answer(inp, planted) for inp in G.load()]
    for v in vectors:  # tie=last models a crate whose Vec comes in descending position order (Barbell keeps its first strictly lowest)
        if v["kind"] == "search_set":
            v["matches"].reverse()
    f = tmp_path / "kat.jsonl"
    f.write_text("\n".join(json.dumps(v) for v in vectors) + "\n")
    assert check_kat_file(str(f), planted) == len(vectors)
    with pytest.raises(AssertionError):
        check_kat_file(str(f))                    # the default policy does not explain them
    pol, rep = ref_fit.fit_kat(vectors)
    assert all(r["explained"] == r["vectors"] for r in rep.values()), rep
    assert ref_fit.to_text(pol) == planted
    # nothing else explains them as well, except the other spelling of the planted order's class (adjacent M / S swapped: never distinguishable)
    for r in rep.values():
        assert all(f == "trace" and all(v.replace("SM", "MS") == "MSDI" for v in vals) for f, vals in r["not_told_apart"].items()), rep


def test_fit_searches_refuted_values_only_on_request(monkeypatch):
    """a planted policy the reference's own vectors refute (trace=MDSI breaks cigar_parse.rs:163-176, rcpath=mirror mis-windows the rapid kits:
    tests/golden/policy_feasible.json) is outside the default search space and found with the full one (ref_fit.py --all)"""
    import gen_kat_inputs as G
    import ref_fit

    planted = "lm=right,rc=scan,trace=MDSI,ovh=floor,rcpath=mirror,tie=first,lodhi=3:0.5:1111"
    vectors = [ref_fit.checker_answer(inp, planted) for inp in G.load() if inp[0] == "search"]
    assert "MDSI" in ref_fit.REFUTED["trace"] and "MDSI" not in ref_fit.SPACE["trace"] and ref_fit.SPACE["rcpath"] == ["fwd"]
    pol, rep = ref_fit.fit_kat(vectors)
    assert rep["search"]["explained"] < rep["search"]["vectors"]
    for k, v in ref_fit.FULL_SPACE.items():
        monkeypatch.setitem(ref_fit.SPACE, k, list(v))
    pol, rep = ref_fit.fit_kat(vectors)
    assert rep["search"]["explained"] == rep["search"]["vectors"] and pol["trace"] == "MDSI" and pol["rcpath"] == "mirror"


def test_fit_recovers_a_planted_policy_from_a_tsv(tmp_path):
    """the same through the whole path: a ref.tsv replayed from the checker under a non-default policy on the noisy export
    config; ref_diff --fit's search ends on a policy under which every read's rows are identical"""
    import ref_fit

    planted = "lm=left,trace=MSID,tie=last,lodhi=3:0.5:1121"
    d = ref_export.export_config("nbd96n", str(tmp_path), 400)
    oracle_tsv(d, "nbd96n", os.path.join(d, "ref.tsv"), policy=planted)
    ours = str(tmp_path / "default.tsv")
    ids, _ = oracle_tsv(d, "nbd96n", ours)
    rep = ref_diff.diff_rows(ref_diff.parse_tsv(os.path.join(d, "ref.tsv")), ref_diff.parse_tsv(ours), ids)
    assert rep["reads_differ"] > 5                                  # the default does not reproduce it
    pol, rep = ref_fit.fit_tsv(d, 400)
    assert rep["reads_identical"] == 400, rep
    assert pol["lm"] == "left" and pol["trace"] in ("MSID", "SMID")


def test_kat_ingest_selfcheck(tmp_path):
    """the ingestion code itself, on vectors in kat.rs's format written from the oracle's own answers"""
    inv = {v: k for k, v in OPS.items()}
    lines = []
    for ops in ("=" * 44, "==X==I=D====", "=" * 10 + "ID" + "=" * 30):
        v = po.lodhi([OPS[c] for c in ops])
        lines.append(json.dumps({"kind": "lodhi", "ops": ops, "score": v, "bits": "%#018x" % int(np.float64(v).view(np.uint64))}))
    flank, text = "ATTGCTAAGGTTAANNNNNNNNNNNNNNNNNNNNNNNNCAGCACCT", "GGACTTGA" + "ATTGCTAAGGTTAA" + "CACAAAGACACCGACAACTTTCTT" + "CAGCTCCT" + "GGGGACGTACGTTGCATGCATTAGC"
    for alpha in (-1.0, 0.4):
        ms, h = po.search(flank.encode(), text.encode(), 5, alpha=None if alpha < 0 else alpha, rc=True)
        po.free_matches(h)
        assert ms
        lines.append(json.dumps({"kind": "search", "searcher": "rc", "alpha": alpha, "pattern": flank, "text": text, "k": 5, "matches": [
            {"text_start": m.text_start, "text_end": m.text_end, "pattern_start": m.pattern_start, "pattern_end": m.pattern_end, "cost": m.cost,
             "strand": "Rc" if m.strand else "Fwd", "pattern_idx": 0, "ops": m.cigar, "path": [list(q) for q in m.path]} for m in ms]}))
    pats, win = ["ACGTACGTAC", "ACGTTCGTAC", "TTTTTTTTTT"], "GGACGTACGTACGG"
    allm = []
    for idx, pp in enumerate(pats):
        ms, h = po.search(pp.encode(), win.encode(), 4, alpha=None, rc=False)
        po.free_matches(h)
        allm += [{"text_start": m.text_start, "text_end": m.text_end, "pattern_start": 0, "pattern_end": len(pp), "cost": m.cost, "strand": "Fwd",
                  "pattern_idx": idx, "ops": m.cigar, "path": []} for m in ms]
    assert allm
    lines.append(json.dumps({"kind": "search_set", "patterns": pats, "text": win, "k": 4, "matches": allm}))
    f = tmp_path / "kat.jsonl"
    f.write_text("\n".join(lines) + "\n")
    assert check_kat_file(str(f)) == len(lines)
    bad = json.loads(lines[0]); bad["bits"] = "%#018x" % (int(bad["bits"], 16) + 1)
    f.write_text(json.dumps(bad) + "\n")
    with pytest.raises(AssertionError):
        check_kat_file(str(f))


def check_kat_file(path, policy=None):
    with po.policy(policy):
        return _check_kat_file(path, policy)


def _check_kat_file(path, policy):
    n = 0
    tie_last = "tie=last" in (policy or "")
    for line in open(path):
        v = json.loads(line)
        if v["kind"] == "lodhi":
            got = po.lodhi([OPS[c] for c in v["ops"]])
            assert np.float64(got).view(np.uint64) == int(v["bits"], 16), (v["ops"], got, v["score"])
        elif v["kind"] == "search":
            alpha = None if v["alpha"] < 0 else v["alpha"]
            ms, h = po.search(v["pattern"].encode(), v["text"].encode(), v["k"], alpha=alpha, rc=True)
            po.free_matches(h)
            want = [(w["text_start"], w["text_end"], w["pattern_start"], w["pattern_end"], w["cost"], 0 if w["strand"] == "Fwd" else 1,
                     w["ops"], w["path"]) for w in v["matches"]]
            assert [_match_tuple(m) for m in ms] == want, v
        elif v["kind"] == "search_set":
            # what collect_candidates_for_region keeps per pattern (searcher.rs:294-300: the first strictly lowest of the Vec)
            import ref_fit

            ours = ref_fit.checker_answer(ref_fit.input_of(v), policy)
            assert ref_fit.kept_per_pattern(v["matches"]) == ref_fit.kept_per_pattern(ours["matches"], last=tie_last), v
        n += 1
    return n


def _golden_tsvs():
    d = os.path.join(GOLD, "ref_tsv")
    return sorted(f[:-4] for f in os.listdir(d) if f.endswith(".tsv")) if os.path.isdir(d) else []


def test_golden_tsv_oracle(tmp_path):
    cfgs = _golden_tsvs()
    if not cfgs:
        pytest.skip(UNPINNED + " (no tests/golden/ref_tsv/<config>.tsv: see tools/ref_golden/README.md)")
    for cfg in cfgs:
        n = int(json.load(open(os.path.join(GOLD, "ref_tsv", cfg + ".json")))["n_reads"])
        d = ref_export.export_config(cfg, str(tmp_path), n)
        ours = str(tmp_path / (cfg + ".oracle.tsv"))
        ids, _ = oracle_tsv(d, cfg, ours, policy=golden_policy())
        rep = ref_diff.diff_rows(ref_diff.parse_tsv(os.path.join(GOLD, "ref_tsv", cfg + ".tsv")), ref_diff.parse_tsv(ours), ids)
        assert rep["identical"], json.dumps({k: rep[k] for k in ("buckets", "hazards", "examples")}, indent=1)


@pytest.mark.gpu
def test_golden_tsv_hip(tmp_path):
    cfgs = _golden_tsvs()
    if not cfgs:
        pytest.skip(UNPINNED + " (no tests/golden/ref_tsv/<config>.tsv)")
    for cfg in cfgs:
        n = int(json.load(open(os.path.join(GOLD, "ref_tsv", cfg + ".json")))["n_reads"])
        d = ref_export.export_config(cfg, str(tmp_path), n)
        ours = str(tmp_path / (cfg + ".hip.tsv"))
        A.annotate([os.path.join(d, "reads.fastq")], ours, ref_export.config_groups(cfg), policy=golden_policy())
        rep = ref_diff.diff_rows(ref_diff.parse_tsv(os.path.join(GOLD, "ref_tsv", cfg + ".tsv")), ref_diff.parse_tsv(ours),
                                 ref_diff.fastq_ids(os.path.join(d, "reads.fastq")))
        assert rep["identical"], json.dumps({k: rep[k] for k in ("buckets", "hazards", "examples")}, indent=1)
