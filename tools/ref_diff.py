#!/usr/bin/env python3
"""ref_diff.py — diff this repo's annotation.tsv against real Barbell's on a read set exported by ref_export.py.

SURVEY §8c's hook: "if a barbell binary is discoverable at run time the harness additionally diffs against real
Barbell and reports the mismatch rate, otherwise it reports 'reference parity unpinned beyond KATs'".  The
reference writes rows per worker-thread batch (annotator.rs:137-143), so its TSV is first stable-sorted back into
input read order (rows of one read stay in the order `collapse_overlapping_matches` returned them); then every read's
rows are compared and mismatches are bucketed by the column class that differs, which points at the unpinned
third-party hazard behind it (oracle/README.md):

  row_count   different number of rows for a read             -> H1 (local-minimum rule), H4 (overhang), H6
  flank       read_start_flank / read_end_flank / flank_cost /
              rel_dist_to_end                                  -> H1, H3 (traceback start), H4
  label       match_type or label (which barcode was called)   -> H7 (search_encoded_patterns), H8 (Lodhi)
  bar         read_start_bar / read_end_bar / bar_start /
              bar_end / barcode_cost with the same label       -> H3 (traceback preference)
  strand      strand column                                    -> H2, H5
  order       same rows, different order within the read       -> H2 (match order / stable-sort ties)

  tools/ref_diff.py EXPORT_DIR/<config> [--barbell BIN] [--ours TSV | --ours-bin barbell-amd] [-t THREADS] [--json OUT]
  tools/ref_diff.py EXPORT_DIR/<config> --fit     which policy (include/barbell_amd_policy.h) reproduces ref.tsv: the hazards
                                                  above are switchable in kernels and checker alike, tools/ref_fit.py picks

`--barbell` defaults to $BARBELL_BIN or `barbell` on PATH; its output is cached as EXPORT_DIR/<config>/ref.tsv.
`--ours` is a TSV made by this repo (barbell-amd annotate / barbell_amd.annotate.annotate); without it the product CLI
is run on the GPU with the manifest's flags.  Exit code 0 = identical, 1 = differences, 2 = no reference available.
"""
import argparse
import csv
import json
import os
import shutil
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
from barbell_amd.parallel import effective_cpus  # noqa: E402

COLS = ["read_id", "read_len", "rel_dist_to_end", "read_start_bar", "read_end_bar", "read_start_flank", "read_end_flank", "bar_start",
        "bar_end", "match_type", "flank_cost", "barcode_cost", "label", "strand", "cuts"]
CLASSES = {
    "flank": ["read_start_flank", "read_end_flank", "flank_cost", "rel_dist_to_end"],
    "label": ["match_type", "label"],
    "bar": ["read_start_bar", "read_end_bar", "bar_start", "bar_end", "barcode_cost"],
    "strand": ["strand"],
}
HAZARDS = {"row_count": "H1,H4,H6", "flank": "H1,H3,H4", "label": "H7,H8", "bar": "H3", "strand": "H2,H5", "order": "H2"}
UNPINNED = "unpinned beyond KATs"


def find_barbell(explicit=None):
    """path of a real `barbell` binary: explicit argument, $BARBELL_BIN, or PATH; None if there is none"""
    for cand in (explicit, os.environ.get("BARBELL_BIN"), shutil.which("barbell")):
        if cand and os.path.isfile(cand) and os.access(cand, os.X_OK):
            return cand
    return None


def parse_tsv(path):
    """annotation.tsv -> list of row dicts (csv crate quoting = RFC 4180, tab-delimited); an empty file = no rows"""
    if not os.path.exists(path) or os.path.getsize(path) == 0:
        return []
    with open(path, newline="") as f:
        rd = csv.reader(f, delimiter="\t")
        head = next(rd)
        if head != COLS:
            raise ValueError(f"{path}: unexpected header {head}")
        return [dict(zip(COLS, r)) for r in rd]


def fastq_ids(path):
    ids = []
    with open(path, "rb") as f:
        for i, line in enumerate(f):
            if i % 4 == 0:
                ids.append(line[1:].split(None, 1)[0].decode() if line[1:].strip() else "")
    return ids


def group_by_read(rows, order):
    """rows -> {read index: [rows in file order]}; stable with respect to the file's row order"""
    per = {}
    for r in rows:
        per.setdefault(order[r["read_id"]], []).append(r)
    return per


def diff_rows(ref_rows, our_rows, ids, max_examples=10):
    """per-read comparison; reads named in neither file count as equal"""
    order = {rid: i for i, rid in enumerate(ids)}
    ref, ours = group_by_read(ref_rows, order), group_by_read(our_rows, order)
    rep = {"reads_total": len(ids), "reads_with_rows_ref": len(ref), "reads_with_rows_ours": len(ours), "rows_ref": len(ref_rows),
           "rows_ours": len(our_rows), "reads_equal": 0, "reads_differ": 0,
           "buckets": {k: 0 for k in ("row_count", "flank", "label", "bar", "strand", "order")}, "examples": []}
    key = lambda r: tuple(r[c] for c in COLS)
    for i in range(len(ids)):
        a, b = ref.get(i, []), ours.get(i, [])
        if [key(r) for r in a] == [key(r) for r in b]:
            rep["reads_equal"] += 1
            continue
        rep["reads_differ"] += 1
        hit = set()
        if len(a) != len(b):
            hit.add("row_count")
        elif sorted(key(r) for r in a) == sorted(key(r) for r in b):
            hit.add("order")
        else:
            for ra, rb in zip(a, b):
                label_same = all(ra[c] == rb[c] for c in CLASSES["label"])
                for cls, cols in CLASSES.items():
                    if any(ra[c] != rb[c] for c in cols):
                        if cls == "bar" and not label_same:
                            continue  # a different barcode call moves the bar columns with it: counted under label
                        hit.add(cls)
        for h in hit:
            rep["buckets"][h] += 1
        if len(rep["examples"]) < max_examples:
            rep["examples"].append({"read": ids[i], "classes": sorted(hit), "ref": ["\t".join(key(r)) for r in a],
                                    "ours": ["\t".join(key(r)) for r in b]})
    n = max(1, len(ids))
    rep["mismatch_rate"] = rep["reads_differ"] / n
    rep["bucket_rates"] = {k: v / n for k, v in rep["buckets"].items()}
    rep["hazards"] = {k: HAZARDS[k] for k, v in rep["buckets"].items() if v}
    rep["identical"] = rep["reads_differ"] == 0
    return rep


def run_annotate(binary, args, fastq, out_tsv, threads, cwd):
    """`<binary> annotate <args> -i fastq -o out -t threads`; returns wall seconds.  Raises on a non-zero exit."""
    cmd = [binary, "annotate"] + list(args) + ["-i", fastq, "-o", out_tsv, "-t", str(threads)]
    t = time.perf_counter()
    p = subprocess.run(cmd, cwd=cwd, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    dt = time.perf_counter() - t
    if p.returncode != 0:
        raise RuntimeError(f"{' '.join(cmd)} failed ({p.returncode}): {p.stderr.decode(errors='replace')[-800:]}")
    return dt


def reference_check(export_dir, barbell=None, ours_tsv=None, ours_bin=None, threads=None, rerun=False):
    """The whole hook: returns (report dict, reference seconds or None).  report["reference_parity"] is either the
    string 'unpinned beyond KATs' (no binary, no cached ref.tsv) or 'identical' / 'differs'."""
    man = json.load(open(os.path.join(export_dir, "manifest.json")))
    fastq = os.path.join(export_dir, "reads.fastq")
    threads = threads or effective_cpus()
    ref_tsv = os.path.join(export_dir, "ref.tsv")
    secs = None
    bin_ = find_barbell(barbell)
    if bin_ and (rerun or not os.path.exists(ref_tsv)):
        secs = run_annotate(bin_, man["barbell_args"], "reads.fastq", "ref.tsv", threads, export_dir)
    if not os.path.exists(ref_tsv):
        return {"reference_parity": UNPINNED, "config": man["config"]}, None
    if ours_tsv is None:
        ours_bin = ours_bin or os.path.join(ROOT, "barbell_amd", "bin", "barbell-amd")
        ours_tsv = os.path.join(export_dir, "ours.tsv")
        run_annotate(ours_bin, man["barbell_args"], "reads.fastq", "ours.tsv", threads, export_dir)
    rep = diff_rows(parse_tsv(ref_tsv), parse_tsv(ours_tsv), fastq_ids(fastq))
    rep["config"] = man["config"]
    rep["reference_parity"] = "identical" if rep["identical"] else "differs"
    rep["reference_binary"] = bin_
    if secs is not None:
        rep["reference_seconds"] = secs
        rep["reference_reads_per_s"] = man["n_reads"] / secs
        rep["reference_threads"] = threads
    return rep, secs


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("export_dir")
    ap.add_argument("--barbell")
    ap.add_argument("--ours")
    ap.add_argument("--ours-bin")
    ap.add_argument("-t", "--threads", type=int)
    ap.add_argument("--rerun", action="store_true", help="run the reference even if ref.tsv is cached")
    ap.add_argument("--json")
    ap.add_argument("--fit", action="store_true",
                    help="search the policy space (include/barbell_amd_policy.h) with the CPU checker for the setting that reproduces ref.tsv "
                         "(tools/ref_fit.py); prints it and how many reads it explains")
    ap.add_argument("--fit-reads", type=int, default=2000)
    a = ap.parse_args()
    if a.fit:
        import ref_fit

        if find_barbell(a.barbell) and (a.rerun or not os.path.exists(os.path.join(a.export_dir, "ref.tsv"))):
            man = json.load(open(os.path.join(a.export_dir, "manifest.json")))
            run_annotate(find_barbell(a.barbell), man["barbell_args"], "reads.fastq", "ref.tsv", a.threads or effective_cpus(), a.export_dir)
        if not os.path.exists(os.path.join(a.export_dir, "ref.tsv")):
            print("no `barbell` binary (BARBELL_BIN / PATH) and no cached ref.tsv: reference parity " + UNPINNED, file=sys.stderr)
            return 2
        pol, rep = ref_fit.fit_tsv(a.export_dir, a.fit_reads, log=lambda m: print(m, file=sys.stderr))
        print(json.dumps({"policy": ref_fit.to_text(pol), "report": rep}, indent=1))
        return 0 if rep["reads_identical"] == rep["reads"] else 1
    rep, _ = reference_check(a.export_dir, a.barbell, a.ours, a.ours_bin, a.threads, a.rerun)
    txt = json.dumps(rep, indent=1)
    if a.json:
        open(a.json, "w").write(txt + "\n")
    print(txt)
    if rep["reference_parity"] == UNPINNED:
        print("no `barbell` binary (BARBELL_BIN / PATH) and no cached ref.tsv: reference parity " + UNPINNED, file=sys.stderr)
        return 2
    return 0 if rep["identical"] else 1


if __name__ == "__main__":
    sys.exit(main())
