"""The reference's stand-alone steps on files (barbell_amd/steps.py, `python -m barbell_amd filter|inspect|trim`): bin/main.rs:340-420,
filter.rs:10-119, inspect.rs:119-208, trim.rs:317-480.  CPU: the TSV comes back as the rows it was written from, the stand-in label
groups are well-formed, the command line takes the reference's flags.  GPU: the steps run one by one on the files of a fused run give
the fused run's files, byte for byte."""
import os

import numpy as np
import pytest

from barbell_amd import _abi, kits
from barbell_amd import annotate as A
from barbell_amd import filter as F

KIT = "SQK-NBD114-96"


def _oracle_rows(groups, n=300, seed=5):
    """rows of noisy reads (tag rows and flank-only rows) from the CPU checker"""
    from barbell_amd.parallel import effective_cpus
    from oracle import pyoracle
    from tests.common import noisy_reads

    _, bases, offsets = noisy_reads("nbd96", seed, n, 300, 1500, rate=0.10)
    rows = pyoracle.Oracle([g.as_tuple() for g in groups]).annotate(bases, offsets, n_threads=effective_cpus(), fast=True)
    return bases, offsets, rows


def test_cuts_column_round_trip():
    from barbell_amd import steps

    v = np.zeros(1, dtype=F.VERDICT_DTYPE)[0]
    v["n_cuts"], v["match_idx"] = 2, 3
    v["cuts"][0]["direction"], v["cuts"][0]["group_id"] = 1, 0
    v["cuts"][1]["direction"], v["cuts"][1]["group_id"] = 0, 12
    s = F.format_cuts(v)
    assert s == "After(0):3,Before(12):3"
    assert steps.parse_cuts(s) == ([(1, 0), (0, 12)], 3)
    assert steps.parse_cuts("") == ([], None)
    for bad in ("After(0)", "Around(0):1", "After(x):1", "After(0):1,Before(0):2", "After(0):1,After(1):1,After(2):1,After(3):1"):
        with pytest.raises(steps.TsvError):
            steps.parse_cuts(bad)


def test_tsv_comes_back_as_the_rows_it_was_written_from(tmp_path):
    from barbell_amd import steps

    groups = kits.groups_from_kit(KIT, flank_max_errors=3)
    _, _, rows = _oracle_rows(groups)
    assert len(rows) > 200 and (rows["barcode_idx"] < 0).any() and (rows["barcode_idx"] >= 0).any()
    n_reads = int(rows["read_idx"].max()) + 1
    ids = ["r%d" % i for i in range(n_reads)]
    ids[int(rows["read_idx"][3])] = 'odd\t"id"'   # quoted by the csv writer (annotator.rs:246-251), unquoted by the reader
    text = A.TSV_HEADER + "\n" + "\n".join(A.format_rows(rows, ids, groups)) + "\n"
    p = tmp_path / "a.tsv"
    p.write_text(text)
    # (a) with the run's groups: the very rows (read_idx renumbered over the reads that have rows)
    space = steps.LabelSpace(groups)
    for batch_rows in (1 << 18, 7):
        got, got_ids = [], []
        for b in steps.read_annotation_tsv(str(p), space, batch_rows):
            first = np.r_[True, b.rows["read_idx"][1:] != b.rows["read_idx"][:-1]]
            assert first.sum() == len(b.read_ids)          # a read never straddles two batches
            r = b.rows.copy()
            r["read_idx"] += len(got_ids)
            got.append(r)
            got_ids += b.read_ids
            assert not b.has_cuts and (b.verdicts["n_cuts"] == 0).all()
        got = np.concatenate(got)
        assert batch_rows > 7 or len(got_ids) > 20
        want = rows.copy()
        _, want["read_idx"] = np.unique(rows["read_idx"], return_inverse=True)
        assert got.tobytes() == want.tobytes()
        assert got_ids == [ids[i] for i in np.unique(rows["read_idx"])]
    # (b) without them: stand-in groups that carry the file's labels; the text written back is the file
    space2 = steps.LabelSpace.from_labels(steps.scan_labels(str(p)))
    assert space2.standin and 1 <= len(space2.groups) <= 2
    bs = list(steps.read_annotation_tsv(str(p), space2))
    assert len(bs) == 1
    assert steps.rows_to_tsv_text(bs[0].rows, bs[0].read_ids, space2) == text
    # an unknown label under explicit groups is an error, not a silent slot
    with pytest.raises(steps.TsvError):
        list(steps.read_annotation_tsv(str(p), steps.LabelSpace(kits.groups_from_kit("SQK-RBK114-24"))))
    # a field that is not an integer, a strand that is none
    for col, val in ((1, "12x"), (3, ""), (13, "Forward"), (9, "Xtag")):
        f = text.splitlines()[5].split("\t")
        f[col] = val
        (tmp_path / "broken.tsv").write_text("\n".join(text.splitlines()[:5] + ["\t".join(f)]) + "\n")
        with pytest.raises(steps.TsvError):
            list(steps.read_annotation_tsv(str(tmp_path / "broken.tsv"), space))
    # trim's view: one read per distinct id wherever its lines lie
    lines = text.splitlines()
    (tmp_path / "shuffled.tsv").write_text("\n".join([lines[0]] + lines[1:][::2] + lines[1:][1::2]) + "\n")
    b = list(steps.read_annotation_tsv(str(tmp_path / "shuffled.tsv"), space, group_consecutive=False))[0]
    assert len(b.read_ids) == len(np.unique(rows["read_idx"])) and (np.diff(b.rows["read_idx"].astype(np.int64)) >= 0).all()


def test_stand_in_groups_are_well_formed_query_groups():
    """bb_create must take them: the checker builds the same geometry from them (barcodes.rs:105-197) — 14 / 24 / 8 like SQK-NBD114-96"""
    from barbell_amd import steps
    from oracle import pyoracle

    pairs = [("Ftag", "NB%02d" % i) for i in range(1, 30)] + [("Fflank", "flank"), ("Rflank", "flank"), ("Rtag", "x y"), ("Ftag", "NB01")]
    space = steps.LabelSpace.from_labels(pairs)
    assert [g.match_type for g in space.groups] == [_abi.BB_FTAG, _abi.BB_RTAG]
    assert space.lookup("Ftag", "NB07") == (0, 6) and space.lookup("Rtag", "x y") == (1, 0)
    assert space.lookup("Fflank", "flank") == (0, -1) and space.lookup("Rflank", "flank") == (1, -1)
    orc = pyoracle.Oracle([g.as_tuple() for g in space.groups])
    geo = orc.geometry() if hasattr(orc, "geometry") else None
    if geo is not None:
        assert all(int(g["bar_hi"]) - int(g["bar_lo"]) == 23 for g in geo)
    for g in space.groups:
        assert len(set(g.seqs)) == len(g.seqs) and len({len(s) for s in g.seqs}) == 1
    many = steps.LabelSpace.from_labels(("Ftag", "L%d" % i) for i in range(2500))
    assert len(many.groups) == 3 and many.lookup("Ftag", "L2499") == (2, 2499 - 2 * steps.GROUP_MAX_LABELS)
    assert steps.LabelSpace.from_labels([]).groups == []


def test_command_line_takes_the_reference_flags():
    from barbell_amd.__main__ import parser

    ap = parser()
    a = ap.parse_args("filter -i a.tsv -o f.tsv -f p1.txt p2.txt --dropped d.tsv --verbose".split())
    assert (a.command, a.file, a.dropped) == ("filter", ["p1.txt", "p2.txt"], "d.tsv")
    a = ap.parse_args("trim -i f.tsv -r a.fastq b.fastq -o out --no-label --no-orientation --no-flanks --only-side left --failed-out x --skip-trim --flip --gzip".split())
    assert a.reads == ["a.fastq", "b.fastq"] and a.only_side == "left" and a.skip_trim and a.flip and a.gzip and not a.sort_labels
    a = ap.parse_args("inspect -i a.tsv".split())
    assert (a.top_n, a.bucket_size, a.read_pattern_out) == (10, 250, None)
    a = ap.parse_args("annotate -i r.fastq --kit SQK-RBK114-24".split())
    assert (a.output, a.threads, a.min_score, a.min_score_diff, a.alpha, a.barcode_types) == ("output.tsv", 10, 0.2, 0.1, 0.4, ["Ftag"])
    a = ap.parse_args("kit -k SQK-NBD114-96 -i r.fastq -o out --maximize".split())
    assert a.maximize and a.failed_out is None
    with pytest.raises(SystemExit):
        ap.parse_args("trim -i f.tsv -o out".split())   # -r is required (bin/main.rs:143)


def test_command_line_reports_errors_like_the_reference(tmp_path, capsys):
    from barbell_amd.__main__ import main

    (tmp_path / "f.tsv").write_bytes(b"")
    (tmp_path / "r.fastq").write_bytes(b"@a\nACGT\n+\nIIII\n")
    base = ["trim", "-i", str(tmp_path / "f.tsv"), "-r", str(tmp_path / "r.fastq"), "-o", str(tmp_path / "t")]
    assert main(base) == 0
    assert main(base + ["--sort-labels", "--only-side", "left"]) == 1          # trim.rs:331-335
    assert "ambiguous" in capsys.readouterr().err
    (tmp_path / "p.txt").write_text("Xtag[fw, *, @left(0..250)]\n")
    assert main(["filter", "-i", str(tmp_path / "f.tsv"), "-o", str(tmp_path / "o.tsv"), "-f", str(tmp_path / "p.txt")]) == 1
    assert main(["inspect", "-i", str(tmp_path / "missing.tsv")]) == 1
    assert capsys.readouterr().err.count("Error during processing:") == 2


def test_empty_annotation_files_need_no_device(tmp_path):
    """the csv writer emits the header with the first record only: a run without rows leaves an EMPTY file, and the steps take it"""
    from barbell_amd import steps

    (tmp_path / "a.tsv").write_bytes(b"")
    logs = []
    assert steps.filter_file(str(tmp_path / "a.tsv"), str(tmp_path / "f.tsv"), [F.pattern_from_str("Ftag[fw, *, @left(0..250), >>]")],
                             str(tmp_path / "d.tsv"), log=logs.append) == (0, 0, 0)
    assert (tmp_path / "f.tsv").read_bytes() == b"" and (tmp_path / "d.tsv").read_bytes() == b""
    insp = steps.inspect_file(str(tmp_path / "a.tsv"), log=logs.append)
    assert insp.summary(10)[0] == "Found 0 unique patterns"
    (tmp_path / "r.fastq").write_bytes(b"@a\nACGT\n+\nIIII\n@b x\nAC\n+\nII\n")
    assert steps.trim_file(str(tmp_path / "f.tsv"), [str(tmp_path / "r.fastq")], str(tmp_path / "t"), log=logs.append) == (2, 0, 0, 0)
    assert list((tmp_path / "t").iterdir()) == []
    (tmp_path / "bad.tsv").write_text("read_id\tread_len\n")
    with pytest.raises(steps.TsvError):
        steps.scan_labels(str(tmp_path / "bad.tsv"))
    # --verbose leaves '{step}.{unix ms}.log' next to the output (progress.rs:96-144)
    steps.filter_file(str(tmp_path / "a.tsv"), str(tmp_path / "f.tsv"), [], verbose=True, log=logs.append)
    (log,) = [p for p in tmp_path.iterdir() if p.name.startswith("filter.") and p.name.endswith(".log")]
    assert log.read_text() == "step\tmetric\tcount\nfilter\tTotal:\t0\nfilter\tKept:\t0\nfilter\tDropped:\t0\n"
    assert log.name.split(".")[1].isdigit() and len(log.name.split(".")[1]) == 13


def _write_fastq(path, groups, n, seed):
    bases, offsets = A.synth_reads_host(groups, seed, 300, 2500, 0, n)
    with open(path, "wb") as f:
        for i in range(n):
            s = bases[int(offsets[i]):int(offsets[i + 1])].tobytes()
            f.write(b"@q%d ch=%d\n" % (i, i % 7) + s + b"\n+\n" + bytes(33 + (j * 7 + i) % 40 for j in range(len(s))) + b"\n")


def _dir_bytes(d, suffix):
    return {p.name: p.read_bytes() for p in d.iterdir() if p.name.endswith(suffix)}


@pytest.mark.gpu
@pytest.mark.parametrize("with_groups", [False, True])
def test_steps_on_files_give_the_fused_runs_files(tmp_path, with_groups):
    from barbell_amd import steps, trim as T
    from barbell_amd.inspect_rows import Inspector

    groups = kits.groups_from_kit(KIT)
    fq = tmp_path / "r.fastq"
    _write_fastq(fq, groups, 1500, 11)
    pats = F.kit_patterns(KIT, True)
    fused = tmp_path / "fused"
    fused.mkdir()
    holder = {}

    def make_inspector(dm):
        holder["i"] = Inspector(dm, str(fused / "pattern_per_read.tsv"), 250)
        return holder["i"]

    cfg = T.TrimConfig.for_kit(str(fused / "failed.txt"))
    total, found = A.annotate([str(fq)], str(fused / "annotation.tsv"), kits.groups_from_kit(KIT), batch_reads=400, filter_patterns=pats,
                              filtered_file=str(fused / "filtered.tsv"), dropped_file=str(fused / "dropped.tsv"), trim_folder=str(fused),
                              trim_config=cfg, inspector=make_inspector)
    holder["i"].close()
    assert total == 1500 and found > 900
    g = kits.groups_from_kit(KIT) if with_groups else None
    alone = tmp_path / "alone"
    alone.mkdir()
    logs = []
    # filter: small batches, so that reads sit at batch borders
    t, kept, dropped = steps.filter_file(str(fused / "annotation.tsv"), str(alone / "filtered.tsv"), pats, str(alone / "dropped.tsv"), groups=g,
                                         batch_rows=333, log=logs.append)
    assert t == found and kept > 500 and dropped > 20
    assert (alone / "filtered.tsv").read_bytes() == (fused / "filtered.tsv").read_bytes()
    assert (alone / "dropped.tsv").read_bytes() == (fused / "dropped.tsv").read_bytes()
    # inspect
    insp = steps.inspect_file(str(fused / "annotation.tsv"), 10, str(alone / "pattern_per_read.tsv"), 250, groups=g, batch_rows=500, log=logs.append)
    assert (alone / "pattern_per_read.tsv").read_bytes() == (fused / "pattern_per_read.tsv").read_bytes()
    assert insp.summary(10) == holder["i"].summary(10)
    # trim, from the filtered file the stand-alone filter wrote
    cfg2 = T.TrimConfig.for_kit(str(alone / "failed.txt"))
    tt, n_trim, n_fail, _ = steps.trim_file(str(alone / "filtered.tsv"), [str(fq)], str(alone), cfg2, groups=g, batch_reads=257, log=logs.append)
    assert tt == 1500 and n_trim + n_fail == kept
    a, b = _dir_bytes(alone, ".trimmed.fastq"), _dir_bytes(fused, ".trimmed.fastq")
    assert len(b) > 20 and a.keys() == b.keys()
    for k in b:
        assert a[k] == b[k], k
    assert (alone / "failed.txt").read_bytes() == (fused / "failed.txt").read_bytes()
    assert any(l.startswith("filter:") for l in logs) and any(l.startswith("trim:") for l in logs)


@pytest.mark.gpu
def test_command_line_steps_with_other_label_flags(tmp_path):
    """the four commands one after the other through `python -m barbell_amd`'s main(), custom patterns from a file, label flags other than
    the kit preset's; checked against the in-HBM pipeline with the same configuration"""
    from barbell_amd import trim as T
    from barbell_amd.__main__ import main

    groups = kits.groups_from_kit("SQK-RBK114-24")
    fq = tmp_path / "r.fastq"
    _write_fastq(fq, groups, 600, 23)
    pat_file = tmp_path / "pats.txt"
    pat_file.write_text("Ftag[fw, *, @left(0..250), >>]\nFtag[fw, *, @left(0..250), >>]__Ftag[<<, rc, *, @right(0..250)]\n")
    out = tmp_path / "cli"
    out.mkdir()
    assert main(["annotate", "-i", str(fq), "-o", str(out / "a.tsv"), "--kit", "SQK-RBK114-24"]) == 0
    assert main(["filter", "-i", str(out / "a.tsv"), "-o", str(out / "f.tsv"), "-f", str(pat_file), "--dropped", str(out / "d.tsv")]) == 0
    assert main(["inspect", "-i", str(out / "f.tsv"), "-n", "5", "-o", str(out / "ppr.tsv")]) == 0
    assert main(["trim", "-i", str(out / "f.tsv"), "-r", str(fq), "-o", str(out / "t"), "--sort-labels", "--no-flanks", "--gzip"]) == 0
    ref = tmp_path / "ref"
    ref.mkdir()
    cfg = T.TrimConfig(True, True, False, True, None, None, True, False, False, False, True)
    A.annotate([str(fq)], str(ref / "a.tsv"), kits.groups_from_kit("SQK-RBK114-24"), filter_patterns=F.patterns_from_files([str(pat_file)]),
               filtered_file=str(ref / "f.tsv"), dropped_file=str(ref / "d.tsv"), trim_folder=str(ref / "t"), trim_config=cfg)
    for name in ("a.tsv", "f.tsv", "d.tsv"):
        assert (out / name).read_bytes() == (ref / name).read_bytes(), name
    import gzip

    a = {p.name: gzip.open(p).read() for p in (out / "t").iterdir()}
    b = {p.name: gzip.open(p).read() for p in (ref / "t").iterdir()}
    assert len(b) >= 10 and a == b
    # the filtered file carries cuts: inspect shows the cut markers
    assert ">>" in (out / "ppr.tsv").read_text()
    # `kit` through the same front end: the annotate step's file is the one written above
    assert main(["kit", "-k", "SQK-RBK114-24", "-i", str(fq), "-o", str(out / "kit"), "--maximize", "--verbose"]) == 0
    logs = {p.name.split(".")[0]: p.read_text().splitlines() for p in (out / "kit").iterdir() if p.name.endswith(".log")}
    assert set(logs) == {"annotate", "filter", "trim"} and logs["annotate"][1] == "annotate\tTotal:\t600" and logs["trim"][1] == "trim\tTotal:\t600"
    n = {k: [int(l.split("\t")[2]) for l in v[1:]] for k, v in logs.items()}
    assert n["filter"][0] == n["filter"][1] + n["filter"][2] == n["annotate"][1] and n["trim"][1] + n["trim"][3] == n["filter"][1] > 100
    r = _cli("kit", "-k", "SQK-RBK114-24", "-i", fq, "-o", out / "ckit", "--maximize", "--verbose")   # the C++ host counts the same
    assert r.returncode == 0, r.stderr
    clogs = {p.name.split(".")[0]: p.read_text().splitlines() for p in (out / "ckit").iterdir() if p.name.endswith(".log")}
    assert clogs == logs
    assert (out / "kit" / "annotation.tsv").read_bytes() == (out / "a.tsv").read_bytes()
    assert (out / "kit" / "filtered.tsv").exists() and (out / "kit" / "pattern_per_read.tsv").exists()
    assert any(p.name.endswith(".trimmed.fastq") for p in (out / "kit").iterdir())


@pytest.mark.gpu
def test_dual_end_experiment_as_the_reference_readme_runs_it(tmp_path):
    """README "Custom experiment": annotate -q left.fasta,right.fasta -b Ftag,Rtag, then two filters on the same annotation file (exact label,
    `~substring` label, Ftag + Rtag elements) and a trim per filtered file — against the one-pass pipeline run once per filter"""
    from barbell_amd import trim as T
    from barbell_amd.__main__ import main
    from tests.common import EX, config_groups

    fq = tmp_path / "r.fastq"
    _write_fastq(fq, config_groups("dual"), 800, 31)
    q = os.path.join(EX, "native_left.fasta") + "," + os.path.join(EX, "native_right.fasta")
    out = tmp_path / "cli"
    out.mkdir()
    assert main(["annotate", "-i", str(fq), "-q", q, "-b", "Ftag,Rtag", "--flank-max-errors", "5", "-o", str(out / "anno.tsv")]) == 0
    labels = sorted({l.split("\t")[12] for l in (out / "anno.tsv").read_text().splitlines()[1:]} - {"flank"})
    assert len(labels) > 20
    filters = {"g1": "Ftag[fw, *, @left(0..250), >>]__Rtag[<<, fw, *, @right(0..250)]\nFtag[fw, *, @left(0..250), >>]\n",
               "g2": "Ftag[fw, ~%s, @left(0..250), >>]\nRtag[<<, fw, %s, @right(0..250)]\n" % (labels[0][-2:], labels[-1])}
    for name, text in filters.items():
        (tmp_path / (name + ".txt")).write_text(text)
        assert main(["filter", "-i", str(out / "anno.tsv"), "-f", str(tmp_path / (name + ".txt")), "-o", str(out / (name + ".tsv"))]) == 0
        assert main(["trim", "-i", str(out / (name + ".tsv")), "-r", str(fq), "-o", str(out / name), "--no-orientation"]) == 0
        ref = tmp_path / ("ref_" + name)
        ref.mkdir()
        A.annotate([str(fq)], str(ref / "anno.tsv"), config_groups("dual"), filter_patterns=F.patterns_from_files([str(tmp_path / (name + ".txt"))]),
                   filtered_file=str(ref / "f.tsv"), trim_folder=str(ref / "t"), trim_config=T.TrimConfig(True, False, True, False, None, None, True, False, False, False, False))
        assert (out / "anno.tsv").read_bytes() == (ref / "anno.tsv").read_bytes()
        assert (out / (name + ".tsv")).read_bytes() == (ref / "f.tsv").read_bytes() and len((ref / "f.tsv").read_bytes()) > 1000, name
        a, b = _dir_bytes(out / name, ".trimmed.fastq"), _dir_bytes(ref / "t", ".trimmed.fastq")
        assert len(b) >= 1 and a == b, name


CLI = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "barbell_amd", "bin", "barbell-amd")


def _cli(*args):
    import subprocess

    return subprocess.run([CLI, *map(str, args)], capture_output=True, text=True, timeout=600)


def test_cpp_host_steps_without_rows_need_no_device(tmp_path):
    """barbell-amd filter / inspect / trim (host/bb_steps.cpp) on the EMPTY annotation file a run without rows leaves; malformed files fail loudly"""
    (tmp_path / "a.tsv").write_bytes(b"")
    (tmp_path / "p.txt").write_text("Ftag[fw, *, @left(0..250), >>]\n")
    r = _cli("filter", "-i", tmp_path / "a.tsv", "-o", tmp_path / "f.tsv", "-f", tmp_path / "p.txt", "--dropped", tmp_path / "d.tsv")
    assert r.returncode == 0 and "0 reads, 0 kept, 0 dropped" in r.stdout, r.stderr
    assert (tmp_path / "f.tsv").read_bytes() == b"" and (tmp_path / "d.tsv").read_bytes() == b""
    r = _cli("inspect", "-i", tmp_path / "a.tsv")
    assert r.returncode == 0 and "Found 0 unique patterns" in r.stdout
    (tmp_path / "r.fastq").write_bytes(b"@a\nACGT\n+\nIIII\n@b x\nAC\n+\nII\n")
    r = _cli("trim", "-i", tmp_path / "f.tsv", "-r", tmp_path / "r.fastq", "-o", tmp_path / "t")
    assert r.returncode == 0 and "2 reads, 0 trimmed" in r.stdout, r.stderr
    assert list((tmp_path / "t").iterdir()) == []
    r = _cli("trim", "-i", tmp_path / "f.tsv", "-r", tmp_path / "r.fastq", "-o", tmp_path / "t", "--verbose")
    (log,) = [p for p in (tmp_path / "t").iterdir() if p.name.startswith("trim.") and p.name.endswith(".log")]
    assert log.read_text() == "step\tmetric\tcount\ntrim\tTotal:\t2\ntrim\tKept:\t0\ntrim\tKept split:\t0\ntrim\tFailed:\t0\n"   # progress.rs:96-127, :51-70
    log.unlink()
    (tmp_path / "bad.tsv").write_text("read_id\tread_len\nx\t5\n")
    r = _cli("filter", "-i", tmp_path / "bad.tsv", "-o", tmp_path / "f2.tsv", "-f", tmp_path / "p.txt")
    assert r.returncode == 1 and "column missing" in r.stderr
    assert _cli("trim", "-i", tmp_path / "f.tsv", "-o", tmp_path / "t").returncode == 2      # -r is required
    assert _cli("trim", "-i", tmp_path / "f.tsv", "-r", tmp_path / "r.fastq", "-o", tmp_path / "t", "--sort-labels", "--only-side", "left").returncode == 2


@pytest.mark.gpu
def test_cpp_host_steps_on_files_give_the_fused_runs_files(tmp_path):
    """the C++ host's filter / inspect / trim on the files of a one-pass `barbell-amd kit` run: the same filtered.tsv, pattern_per_read.tsv,
    trimmed FASTQ files and failed ids, byte for byte; and a read id that needs quoting survives the round trip"""
    from barbell_amd import kits as K

    groups = K.groups_from_kit(KIT)
    fq = tmp_path / "r.fastq"
    _write_fastq(fq, groups, 1500, 17)
    fused = tmp_path / "fused"
    r = _cli("kit", "-k", KIT, "-i", fq, "-o", fused, "--maximize", "--failed-out", fused / "failed.txt", "--batch-reads", 400, "--verbose")
    assert r.returncode == 0, r.stderr

    def log_of(folder, step):   # what --verbose leaves behind (progress.rs:96-144)
        (p,) = [p for p in folder.iterdir() if p.name.startswith(step + ".") and p.name.endswith(".log")]
        return p.read_text()

    assert log_of(fused, "annotate").startswith("step\tmetric\tcount\nannotate\tTotal:\t1500\nannotate\tKept:\t")
    d = K._data()
    pats = tmp_path / "pats.txt"
    pats.write_text("\n".join(d["pattern_sets"][d["kit_filter"][KIT]["maximize"]]) + "\n")
    alone = tmp_path / "alone"
    alone.mkdir()
    r = _cli("filter", "-i", fused / "annotation.tsv", "-o", alone / "filtered.tsv", "-f", pats, "--dropped", alone / "dropped.tsv", "--verbose")
    assert r.returncode == 0, r.stderr
    assert log_of(alone, "filter") == log_of(fused, "filter") and "filter\tKept:\t" in log_of(alone, "filter")
    assert (alone / "filtered.tsv").read_bytes() == (fused / "filtered.tsv").read_bytes() and len((fused / "filtered.tsv").read_bytes()) > 10000
    kept = {l.split("\t")[0] for l in (alone / "filtered.tsv").read_text().splitlines()[1:]}
    drop = {l.split("\t")[0] for l in (alone / "dropped.tsv").read_text().splitlines()[1:]}
    every = {l.split("\t")[0] for l in (fused / "annotation.tsv").read_text().splitlines()[1:]}
    assert kept and drop and not (kept & drop) and (kept | drop) == every
    r = _cli("inspect", "-i", fused / "annotation.tsv", "-o", alone / "pattern_per_read.tsv")
    assert r.returncode == 0, r.stderr
    assert (alone / "pattern_per_read.tsv").read_bytes() == (fused / "pattern_per_read.tsv").read_bytes()
    assert "Found " in r.stdout and "Pattern 1:" in r.stdout
    r = _cli("trim", "-i", alone / "filtered.tsv", "-r", fq, "-o", alone, "--no-orientation", "--no-flanks", "--only-side", "left", "--failed-out", alone / "failed.txt",
             "--verbose")
    assert r.returncode == 0, r.stderr   # the label flags of the kit preset (use_kit.rs:87-99)
    assert log_of(alone, "trim") == log_of(fused, "trim") and "trim\tKept split:\t" in log_of(alone, "trim")
    a, b = _dir_bytes(alone, ".trimmed.fastq"), _dir_bytes(fused, ".trimmed.fastq")
    assert len(b) > 20 and a.keys() == b.keys()
    for k in b:
        assert a[k] == b[k], k
    assert (alone / "failed.txt").read_bytes() == (fused / "failed.txt").read_bytes()
    # the Python twin on the same files
    from barbell_amd import filter as FF, steps

    steps.filter_file(str(fused / "annotation.tsv"), str(alone / "py_filtered.tsv"), FF.patterns_from_files([str(pats)]), log=lambda s: None)
    assert (alone / "py_filtered.tsv").read_bytes() == (alone / "filtered.tsv").read_bytes()
    # a read id that the csv writer quotes
    lines = (fused / "annotation.tsv").read_text().splitlines()
    first_id = lines[1].split("\t")[0]
    odd = [lines[0]] + [l.replace(first_id + "\t", '"odd ""id""\tx"\t', 1) if l.startswith(first_id + "\t") else l for l in lines[1:]]
    (alone / "odd.tsv").write_text("\n".join(odd) + "\n")
    r = _cli("filter", "-i", alone / "odd.tsv", "-o", alone / "odd_f.tsv", "-f", pats, "--dropped", alone / "odd_d.tsv")
    assert r.returncode == 0, r.stderr
    both = (alone / "odd_f.tsv").read_text() + (alone / "odd_d.tsv").read_text()
    assert '"odd ""id""\tx"\t' in both


@pytest.mark.gpu
def test_filtering_a_filtered_file_keeps_its_cuts_and_trim_refuses_rows_of_other_reads(tmp_path):
    """ADVICE r5.  (a) `filter` on a filtered.tsv with a second, sub-selecting pattern WITHOUT cut markers: the rows keep the cuts they came
    with (filter.rs:204-209 pushes onto the existing cuts), so `trim` on the result still cuts every read — both hosts, same bytes.
    (b) `trim` with a FASTQ whose records are not the annotated ones (same ids, shorter sequences): an error naming the read, in both
    hosts, instead of neighbours' bases in the output (the reference panics on seq[start..end])."""
    from barbell_amd import steps, trim as T

    groups = kits.groups_from_kit(KIT)
    fq = tmp_path / "r.fastq"
    _write_fastq(fq, groups, 600, 23)
    pats = F.kit_patterns(KIT, True)
    one = tmp_path / "one"
    one.mkdir()
    A.annotate([str(fq)], str(one / "annotation.tsv"), kits.groups_from_kit(KIT), batch_reads=300, filter_patterns=pats, filtered_file=str(one / "filtered.tsv"),
               dropped_file=str(one / "dropped.tsv"))
    first = (one / "filtered.tsv").read_text().splitlines()
    assert sum(1 for l in first[1:] if l.split("\t")[-1]) > 200
    # (a) a pattern set that matches every read the first filter kept, and marks no cut
    plain = tmp_path / "plain.txt"
    import re

    plain.write_text("\n".join(re.sub(r"(, )?(>>|<<)[0-9]*(, )?", lambda m: ", " if m.group(1) and m.group(3) else "", p)
                               for p in kits._data()["pattern_sets"][kits._data()["kit_filter"][KIT]["maximize"]]) + "\n")
    assert ">>" not in plain.read_text() and "<<" not in plain.read_text()
    two_py, two_cc = tmp_path / "two_py.tsv", tmp_path / "two_cc.tsv"
    t, kept, dropped = steps.filter_file(str(one / "filtered.tsv"), str(two_py), F.patterns_from_files([str(plain)]), log=lambda s: None)
    assert kept == t > 300 and dropped == 0
    r = _cli("filter", "-i", one / "filtered.tsv", "-o", two_cc, "-f", plain)
    assert r.returncode == 0, r.stderr
    assert two_py.read_bytes() == two_cc.read_bytes() == (one / "filtered.tsv").read_bytes()      # nothing new to add: the file as it was, cuts included
    # ... and with cut markers again: old cuts first, the new ones behind them
    again = tmp_path / "again.tsv"
    steps.filter_file(str(one / "filtered.tsv"), str(again), pats, log=lambda s: None)
    both = [(a.split("\t")[-1], b.split("\t")[-1]) for a, b in zip(first[1:], again.read_text().splitlines()[1:])]
    assert all(b == (a + "," + a if a else "") for a, b in both) and any(a for a, _ in both)
    kit_pats = tmp_path / "kit_pats.txt"
    kit_pats.write_text("\n".join(kits._data()["pattern_sets"][kits._data()["kit_filter"][KIT]["maximize"]]) + "\n")
    r = _cli("filter", "-i", one / "filtered.tsv", "-o", tmp_path / "again_cc.tsv", "-f", kit_pats)
    assert r.returncode == 0 and (tmp_path / "again_cc.tsv").read_bytes() == again.read_bytes(), r.stderr
    cfg = T.TrimConfig.for_kit(None)
    out_a, out_b = tmp_path / "ta", tmp_path / "tb"
    n_a = steps.trim_file(str(one / "filtered.tsv"), [str(fq)], str(out_a), cfg, log=lambda s: None)
    n_b = steps.trim_file(str(two_py), [str(fq)], str(out_b), cfg, log=lambda s: None)
    assert n_a == n_b and n_a[1] > 300 and _dir_bytes(out_a, ".trimmed.fastq") == _dir_bytes(out_b, ".trimmed.fastq")
    # (b) the same ids, every sequence cut to its first 120 bases
    short = tmp_path / "short.fastq"
    recs = fq.read_bytes().split(b"\n")
    with open(short, "wb") as f:
        for i in range(0, len(recs) - 1, 4):
            f.write(recs[i] + b"\n" + recs[i + 1][:120] + b"\n+\n" + recs[i + 3][:120] + b"\n")
    with pytest.raises(steps.TsvError, match="FASTQ record has 120 bases"):
        steps.trim_file(str(one / "filtered.tsv"), [str(short)], str(tmp_path / "tc"), cfg, log=lambda s: None)
    r = _cli("trim", "-i", one / "filtered.tsv", "-r", short, "-o", tmp_path / "td")
    assert r.returncode == 1 and "FASTQ record has 120 bases" in r.stderr and "read 'q" in r.stderr, r.stderr
    assert not any(p.stat().st_size for p in (tmp_path / "td").glob("*.trimmed.fastq"))
    # the kernel's own guard, for callers of the C ABI that skip the hosts' check: rows beyond the read's end are BB_E_INVALID, not a copy
    dm = A.Demuxer()
    for g in kits.groups_from_kit(KIT):
        dm.add_query_group(g)
    space = steps.LabelSpace(kits.groups_from_kit(KIT))
    b = next(steps.read_annotation_tsv(str(one / "filtered.tsv"), space, group_consecutive=False))
    k = int(np.nonzero(b.verdicts["n_cuts"])[0][0])
    rd = int(b.rows["read_idx"][k])
    sel = b.rows["read_idx"] == rd
    rows, ver = b.rows[sel].copy(), b.verdicts[sel].copy()
    rows["read_idx"] = 0
    L = 10                                      # a record far shorter than the rows say
    assert int(rows["read_end_flank"].max()) > L
    F.Filter(dm, pats)                          # (installs the label ids the trim step keys its labels by)
    tr = T.Trimmer(dm, cfg)
    with pytest.raises(A.BarbellError, match="beyond the read's end"):
        tr.trim_batch(rows, ver, np.full(L, ord("A"), dtype=np.uint8), np.full(L, ord("I"), dtype=np.uint8), np.array([0, L], dtype=np.uint64), [b"x"])
    dm.close()
