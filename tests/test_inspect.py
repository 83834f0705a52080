"""Inspect step: get_group_structure (src/inspect/inspect.rs:15-117).  The reference has no
tests for it, so the CPU cases are worked by hand from the function's text; GPU: k_inspect == oracle on the
rows of synthetic reads, and the pattern strings parse back with the filter's pattern parser."""
import numpy as np
import pytest

from barbell_amd import _abi, filter as F, inspect_rows as I
from oracle import pyoracle as po


def row(read, start, end, mtype, strand, read_len, rel):
    r = np.zeros(1, dtype=_abi.ROW_DTYPE)[0]
    r["read_idx"], r["read_len"], r["rel_dist_to_end"] = read, read_len, rel
    r["read_start_bar"], r["read_end_bar"], r["match_type"], r["strand"] = start, end, mtype, strand
    return r


def pats(rows, ver=None, bs=250):
    rows = np.array(rows, dtype=_abi.ROW_DTYPE)
    return [p for _, p in I.patterns(po.inspect_rows(rows, ver, bs), rows)]


def test_bucket_and_tags():
    # single match on the left half: @left(bucket(start)..+bs); bucket is inclusive of its upper bound (inspect.rs:9-13)
    assert pats([row(0, 0, 24, 0, 0, 4000, 1)]) == ["Ftag[fw, *, @left(0..250)]"]
    assert pats([row(0, 250, 274, 0, 0, 4000, 250)]) == ["Ftag[fw, *, @left(0..250)]"]
    assert pats([row(0, 251, 275, 0, 1, 4000, 251)]) == ["Ftag[rc, *, @left(250..500)]"]
    # single match on the right half (rel_dist_to_end <= 0): @right(bucket(len-end)..bucket(len-start)+bs)
    assert pats([row(0, 3900, 3930, 1, 0, 4000, -70)]) == ["Rtag[fw, *, @right(0..250)]"]
    assert pats([row(0, 3400, 3740, 3, 1, 4000, -260)]) == ["Rflank[rc, *, @right(250..750)]"]
    # second element: prev_left when at least as close to the previous end as to the right end, else right
    two = [row(0, 10, 40, 0, 0, 4000, 10), row(0, 60, 90, 2, 0, 4000, 60)]
    assert pats(two) == ["Ftag[fw, *, @left(0..250)]__Fflank[fw, *, @prev_left(0..250)]"]
    two = [row(0, 10, 40, 0, 0, 4000, 10), row(0, 3900, 3960, 0, 1, 4000, -40)]
    assert pats(two) == ["Ftag[fw, *, @left(0..250)]__Ftag[rc, *, @right(0..250)]"]
    two = [row(0, 10, 40, 0, 0, 1000, 10), row(0, 520, 540, 0, 1, 1000, -460)]      # 480 vs 460 -> right
    assert pats(two) == ["Ftag[fw, *, @left(0..250)]__Ftag[rc, *, @right(250..500)]"]
    two = [row(0, 10, 40, 0, 0, 1000, 10), row(0, 500, 540, 0, 1, 1000, -460)]      # 460 vs 460 -> prev_left wins ties
    assert pats(two) == ["Ftag[fw, *, @left(0..250)]__Ftag[rc, *, @prev_left(250..500)]"]
    # overlapping previous element: distance saturates at 0
    two = [row(0, 10, 40, 0, 0, 1000, 10), row(0, 30, 60, 0, 0, 1000, 30)]
    assert pats(two)[0].endswith("@prev_left(0..250)]")
    # bucket size, several reads
    rows = [row(0, 120, 150, 0, 0, 4000, 120), row(1, 120, 150, 0, 0, 4000, 120), row(1, 400, 430, 1, 0, 4000, 400)]
    assert pats(rows, bs=100) == ["Ftag[fw, *, @left(100..200)]", "Ftag[fw, *, @left(100..200)]__Rtag[fw, *, @prev_left(200..300)]"]


def test_cut_marker_and_roundtrip():
    rows = [row(0, 10, 40, 0, 0, 4000, 10), row(0, 3900, 3960, 0, 1, 4000, -40)]
    ver = np.zeros(2, dtype=F.VERDICT_DTYPE)
    ver["n_cuts"] = [1, 2]
    p = pats(rows, ver)[0]
    assert p == "Ftag[fw, *, <<, @left(0..250)]__Ftag[rc, *, >>, @right(0..250)]"       # inspect.rs:72-85
    e = F.pattern_from_str(p).elements                                                     # the output is a valid filter pattern
    assert [(x.match_type, x.orientation, x.range, x.relative_to) for x in e] == [(0, 0, (0, 250), 1), (0, 1, (0, 250), 2)]
    assert [str(c) for c in e[0].cuts] == ["Before(0)"] and [str(c) for c in e[1].cuts] == ["After(0)"]


@pytest.mark.gpu
@pytest.mark.parametrize("config", ["nbd96", "dual"])
def test_gpu_inspect_matches_oracle(config, tmp_path):
    from barbell_amd import annotate as A
    from tests.common import config_groups

    groups = config_groups(config)
    bases, offsets = A.synth_reads_host(groups, 31, 150, 3000, 0, 4000)
    dm = A.Demuxer()
    for g in groups:
        dm.add_query_group(g)
    rows = dm.demux_packed(bases, offsets)
    ver = F.Filter(dm, [F.pattern_from_str("Ftag[fw, *, @left(0..250), >>]"), F.pattern_from_str("Ftag[fw, *, >>]__Ftag[<<, rc, *]")]).verdicts(rows)
    for v, bs in ((None, 250), (ver, 250), (ver, 64), (None, 1)):
        got, want = I.elements(dm, rows, v, bs), po.inspect_rows(rows, v, bs)
        assert got.tobytes() == want.tobytes()
    ids = [f"r{i}" for i in range(4000)]
    ins = I.Inspector(dm, str(tmp_path / "ppr.tsv"))
    ins.add(rows[: len(rows) // 2], ids)   # NB: a real driver splits batches at read boundaries; so does this cut
    ins.close()
    lines = open(tmp_path / "ppr.tsv").read().splitlines()
    assert len(lines) == len(np.unique(rows[: len(rows) // 2]["read_idx"])) and sum(ins.counts.values()) == len(lines)
    for l in lines[:200]:
        rid, p = l.split("\t")
        F.pattern_from_str(p)
    s = ins.summary(3)
    assert s[0].startswith("Found ") and s[-1].startswith("Showed 3 / ")


@pytest.mark.gpu
def test_kit_driver_writes_all_outputs(tmp_path):
    """demux_using_kit (use_kit.rs:11-109): annotation.tsv, pattern_per_read.tsv, filtered.tsv and the
    trimmed FASTQ files from one pass; cross-checked against the separately tested pieces."""
    from barbell_amd import annotate as A, kits, trim as T
    from barbell_amd.use_kit import demux_using_kit

    kit = "SQK-NBD114-96"
    groups = kits.groups_from_kit(kit)
    n = 1200
    bases, offsets = A.synth_reads_host(groups, 8, 300, 2500, 0, n)
    fq = tmp_path / "r.fastq"
    with open(fq, "wb") as f:
        for i in range(n):
            s = bases[int(offsets[i]):int(offsets[i + 1])].tobytes()
            f.write(b"@q%d ch=%d\n" % (i, i % 7) + s + b"\n+\n" + b"5" * len(s) + b"\n")
    out = tmp_path / "kit"
    logs = []
    total, found, insp = demux_using_kit([str(fq)], kit, str(out), maximize=True, batch_reads=500, log=logs.append)
    assert total == n and found > n // 2
    names = set(p.name for p in out.iterdir())
    assert {"annotation.tsv", "pattern_per_read.tsv", "filtered.tsv"} <= names
    trimmed = [x for x in names if x.endswith(".trimmed.fastq")]
    assert len(trimmed) > 20 and all(x.startswith("NB") for x in trimmed)
    a = (out / "annotation.tsv").read_text().splitlines()
    ppr = (out / "pattern_per_read.tsv").read_text().splitlines()
    assert len(ppr) == found == len({l.split("\t")[0] for l in a[1:]})
    kept = {l.split("\t")[0] for l in (out / "filtered.tsv").read_text().splitlines()[1:]}
    n_rec = 0
    for x in trimmed:
        lines = (out / x).read_text().splitlines()
        assert len(lines) % 4 == 0
        for h in lines[0::4]:
            rid = h[1:].split(" ")[0].split("_")[0]
            assert rid in kept and " ch=" in h
        n_rec += len(lines) // 4
    assert n_rec >= len(kept) - 5
    assert any(l.startswith("Found ") for l in logs)
