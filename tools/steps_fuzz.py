#!/usr/bin/env python3
"""dev aid (GPU box): the stand-alone steps on random annotation files — the C++ host (barbell-amd filter / inspect / trim,
host/bb_steps.cpp) against the Python twin (barbell_amd/steps.py) against the CPU checker's filter on the same rows.

Per seed: random reads x rows (all four match types, labels with spaces / tabs / quotes / unicode, read ids that need quoting, rows of one
read in one run or — for trim — scattered), random patterns out of a small grammar (exact / ~substring / * labels, ?N placeholders, every
position tag, cuts with group ids), a FASTQ holding some of the reads.  Checked: filtered / dropped / pattern_per_read files of the two hosts
are the same bytes; the kept set equals the checker's verdicts (oracle/pyoracle.filter_rows on the parsed rows); every trimmed file is the
same bytes.
usage: steps_fuzz.py FIRST_SEED N_SEEDS"""
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from barbell_amd import _abi, annotate as A, filter as F, steps  # noqa: E402
from barbell_amd.trim import TrimConfig  # noqa: E402

CLI = os.path.join(ROOT, "barbell_amd", "bin", "barbell-amd")
MT = _abi.MATCH_TYPE_STR
ODD_LABELS = ["NB01", "NB02", "NB10", "BC 7", "bar\tcode", 'q"uote', "étiquette", "flanker", "x", "NB01_alt"]


def random_case(rng, d):
    n_reads = int(rng.integers(1, 60))
    labels = [ODD_LABELS[i] for i in rng.permutation(len(ODD_LABELS))[: int(rng.integers(1, len(ODD_LABELS) + 1))]]
    ids, lines, fq = [], [A.TSV_HEADER], []
    for r in range(n_reads):
        rid = ["read%d" % r, "r-%d/x" % r, 'odd "%d"' % r, "tab\t%d" % r][int(rng.choice(4, p=[0.7, 0.2, 0.05, 0.05]))]
        ids.append(rid)
        read_len = int(rng.integers(30, 3000))
        pos = 0
        for _ in range(int(rng.integers(1, 5))):
            mt = int(rng.integers(0, 4))
            start = int(min(read_len - 1, pos + rng.integers(0, max(1, read_len // 3))))
            end = int(min(read_len, start + rng.integers(1, 60)))
            pos = end
            lab = "flank" if mt >= 2 else str(rng.choice(labels))
            f = [A._csv_field(rid), read_len, int(min(start, read_len - end)) * (1 if rng.random() < 0.5 else -1), start, end, max(0, start - 5), min(read_len, end + 5),
                 int(rng.integers(0, 20)), int(rng.integers(20, 60)), MT[mt], int(rng.integers(0, 6)), int(rng.integers(0, 12)), A._csv_field(lab),
                 "Fwd" if rng.random() < 0.6 else "Rc", ""]
            lines.append("\t".join(map(str, f)))
        if rng.random() < 0.85:
            seq = "".join(rng.choice(list("ACGT"), read_len))
            qual = "".join(chr(33 + int(q)) for q in rng.integers(0, 40, read_len))
            hdr = rid if "\t" not in rid and " " not in rid else "plain%d" % r   # FASTQ ids end at the first whitespace
            if hdr != rid:   # the annotation's id must be the FASTQ's: rename in the TSV lines of this read
                k = len(lines) - 1
                while k > 0 and lines[k].startswith(A._csv_field(rid) + "\t"):
                    lines[k] = hdr + lines[k][len(A._csv_field(rid)):]
                    k -= 1
                ids[-1] = hdr
            fq.append("@%s%s\n%s\n+\n%s\n" % (hdr, ["", " desc x=1", "\tch=3"][int(rng.integers(0, 3))], seq, qual))
    open(os.path.join(d, "a.tsv"), "w", encoding="utf-8").write("\n".join(lines) + "\n")
    open(os.path.join(d, "r.fastq"), "w", encoding="utf-8").write("".join(fq))
    pats = []
    by_read = {}
    for l in lines[1:]:
        f = next(__import__("csv").reader([l], delimiter="\t"))
        by_read.setdefault(f[0], []).append(f)
    reads = list(by_read.values())
    for _ in range(int(rng.integers(1, 5))):
        el = []
        model = reads[int(rng.integers(0, len(reads)))] if rng.random() < 0.7 else None   # a pattern written after one of the reads: it passes
        for e in range(len(model) if model else int(rng.integers(1, 4))):
            if model:
                f = model[e]
                mt, ori = f[9], "fw" if f[13] == "Fwd" else "rc"
                lab = str(rng.choice(["*", f[12], "~" + f[12][:2], "?%d" % rng.integers(1, 3)])) if not mt.endswith("flank") else "*"
                if any(ch in lab for ch in ",[]\t\"") or lab.startswith("~") and len(lab) < 2:
                    lab = "*"
                tag = ["@left(0..5000)", "@right(0..5000)", "@prev_left(0..5000)"][int(rng.integers(0, 3 if e else 2))]
            else:
                mt, ori = MT[int(rng.integers(0, 4))], str(rng.choice(["fw", "rc"]))
                lab = str(rng.choice(["*", "*", "?1", "?2", "~NB", "~0", "NB01", "NB02", "x"]))
                lab = "*" if mt.endswith("flank") and lab.startswith("?") else lab
                tag = ["@left(0..%d)" % rng.integers(50, 3000), "@right(0..%d)" % rng.integers(50, 3000), "@prev_left(0..%d)" % rng.integers(50, 3000)][int(rng.integers(0, 3 if e else 2))]
            cut = ["", "", ">>", "<<", ">>%d" % rng.integers(1, 4), "<<%d" % rng.integers(1, 4)][int(rng.integers(0, 6))]
            parts = [ori, lab, tag] + ([cut] if cut else [])
            el.append("%s[%s]" % (mt, ", ".join(parts)))
        pats.append("__".join(el))
    open(os.path.join(d, "p.txt"), "w").write("\n".join(pats) + "\n")
    return pats


def run_cli(*args):
    r = subprocess.run([CLI, *args], capture_output=True, text=True, timeout=300)
    if r.returncode != 0:
        raise RuntimeError("barbell-amd %s: rc %d: %s" % (" ".join(args), r.returncode, r.stderr[-500:]))
    return r.stdout


def files(d, suffix):
    return {n: open(os.path.join(d, n), "rb").read() for n in sorted(os.listdir(d)) if n.endswith(suffix)} if os.path.isdir(d) else {}


def main():
    first, count = int(sys.argv[1]), int(sys.argv[2])
    from oracle import pyoracle

    bad = skipped = n_kept = n_dropped = n_files = n_failed = 0
    for seed in range(first, first + count):
        rng = np.random.default_rng(seed)
        with tempfile.TemporaryDirectory() as d:
            try:
                pats = random_case(rng, d)
                P = lambda n: os.path.join(d, n)  # noqa: E731
                try:
                    patterns = F.patterns_from_files([P("p.txt")])
                except ValueError:
                    skipped += 1
                    continue   # the grammar drew something the parser refuses (both hosts share the rule: tested elsewhere)
                run_cli("filter", "-i", P("a.tsv"), "-o", P("c_f.tsv"), "-f", P("p.txt"), "--dropped", P("c_d.tsv"))
                steps.filter_file(P("a.tsv"), P("p_f.tsv"), patterns, P("p_d.tsv"), batch_rows=int(rng.integers(1, 40)), log=lambda s: None)
                for a, b in (("c_f.tsv", "p_f.tsv"), ("c_d.tsv", "p_d.tsv")):
                    if open(P(a), "rb").read() != open(P(b), "rb").read():
                        raise AssertionError("filter: %s differs between the hosts" % a)
                # the checker on the parsed rows
                space = steps.LabelSpace.from_labels(steps.scan_labels(P("a.tsv")))
                bs = list(steps.read_annotation_tsv(P("a.tsv"), space))
                if bs:
                    orc = pyoracle.Oracle([g.as_tuple() for g in space.groups])
                    want = orc.filter_rows(patterns, space.groups, bs[0].rows)
                    kept_want = {bs[0].read_ids[int(r)] for r, p in zip(bs[0].rows["read_idx"], want["pass"]) if p}
                    kept_got = set()
                    if os.path.getsize(P("p_f.tsv")):
                        sp2 = steps.LabelSpace.from_labels(steps.scan_labels(P("p_f.tsv")))
                        for b in steps.read_annotation_tsv(P("p_f.tsv"), sp2):
                            kept_got |= set(b.read_ids)
                    n_kept += len(kept_got)
                    n_dropped += len(set(bs[0].read_ids)) - len(kept_got)
                    if kept_want != kept_got:
                        raise AssertionError("filter: kept reads differ from the checker's (%d vs %d)" % (len(kept_got), len(kept_want)))
                for src in ("a.tsv", "p_f.tsv"):   # inspect without and with cuts
                    run_cli("inspect", "-i", P(src), "-o", P("c_ppr.tsv"), "-s", "100")
                    steps.inspect_file(P(src), 10, P("p_ppr.tsv"), 100, batch_rows=int(rng.integers(1, 40)), log=lambda s: None)
                    if open(P("c_ppr.tsv"), "rb").read() != open(P("p_ppr.tsv"), "rb").read():
                        raise AssertionError("inspect: pattern_per_read differs between the hosts (%s)" % src)
                flags, cfg = [], TrimConfig()
                if rng.random() < 0.4: flags.append("--no-orientation"); cfg.add_orientation = False  # noqa: E701
                if rng.random() < 0.4: flags.append("--no-flanks"); cfg.add_flank = False  # noqa: E701
                if rng.random() < 0.2: flags.append("--no-label"); cfg.add_labels = False  # noqa: E701
                if rng.random() < 0.3: flags.append("--sort-labels"); cfg.sort_labels = True  # noqa: E701
                elif rng.random() < 0.4:
                    side = str(rng.choice(["left", "right"])); flags += ["--only-side", side]; cfg.only_side = side  # noqa: E702
                if rng.random() < 0.2: flags.append("--skip-trim"); cfg.skip_trim = True  # noqa: E701
                if rng.random() < 0.3: flags.append("--flip"); cfg.flip = True  # noqa: E701
                cfg.failed_trimmed_writer = P("p_failed.txt")
                run_cli("trim", "-i", P("p_f.tsv"), "-r", P("r.fastq"), "-o", P("c_t"), "--failed-out", P("c_failed.txt"), *flags)
                steps.trim_file(P("p_f.tsv"), [P("r.fastq")], P("p_t"), cfg, batch_reads=int(rng.integers(1, 30)), log=lambda s: None)
                a, b = files(P("c_t"), ".trimmed.fastq"), files(P("p_t"), ".trimmed.fastq")
                n_files += len(b)
                n_failed += len(open(P("p_failed.txt")).read().splitlines())
                if a != b:
                    raise AssertionError("trim: files differ between the hosts: %s vs %s" % (sorted(a), sorted(b)))
                if open(P("c_failed.txt"), "rb").read() != open(P("p_failed.txt"), "rb").read():
                    raise AssertionError("trim: failed ids differ between the hosts")
            except Exception as e:  # noqa: BLE001
                bad += 1
                print("seed %d: %s: %s" % (seed, type(e).__name__, str(e)[:400]), flush=True)
                if bad <= 3:
                    keep = "/tmp/steps_fuzz_seed%d" % seed
                    subprocess.run(["cp", "-r", d, keep])
                    print("  kept in", keep, "patterns:", pats if "pats" in dir() else "?")
    print("%d seeds (%d skipped: pattern refused), %d bad; reads kept %d / dropped %d, trimmed files %d, failed ids %d" % (count, skipped, bad, n_kept, n_dropped, n_files, n_failed))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
