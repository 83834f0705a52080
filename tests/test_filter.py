"""Filter step.  CPU: the reference's pattern tests (src/filter/pattern.rs:389-937)
transcribed as known-answer tests for the pattern parser and the oracle's match_pattern /
check_filter_pass restatement.  GPU: k_filter verdicts bit-identical to the oracle's on rows of
synthetic reads with the kit preset pattern sets and with random patterns."""
import os

import numpy as np
import pytest

from barbell_amd import _abi, filter as F
from barbell_amd.kits import QueryGroup
from oracle import pyoracle as po


def groups_with_labels(labels):
    """a query group whose barcode labels are `labels` (sequences irrelevant for the filter tests)"""
    pre, suf = b"ACGTACGTAC", b"TTGCATGCAA"
    alphabet = [b"AAAA", b"CCCC", b"GGGG", b"TTTT", b"ACAC", b"GTGT", b"AGAG", b"CTCT"]
    return [QueryGroup([pre + alphabet[i] + suf for i in range(len(labels))], labels, _abi.BB_FTAG, 2)]


def mk(read_idx, start, end, mtype, label_idx, strand=0, read_len=500):
    r = np.zeros(1, dtype=_abi.ROW_DTYPE)[0]
    r["read_idx"], r["read_len"] = read_idx, read_len
    r["read_start_bar"], r["read_end_bar"] = start, end
    r["read_start_flank"], r["read_end_flank"] = start, end
    r["match_type"], r["barcode_idx"], r["strand"] = mtype, label_idx, strand
    return r


GROUPS = groups_with_labels(["XXX", "YYY", "yyyy", "BC14"])
ORC = None


def verdicts(patterns, rows):
    global ORC
    if ORC is None:
        ORC = po.Oracle([g.as_tuple() for g in GROUPS])
    pats = [F.pattern_from_str(p) if isinstance(p, str) else p for p in patterns]
    return ORC.filter_rows(pats, GROUPS, np.array(rows, dtype=_abi.ROW_DTYPE))


def is_match(pattern, rows):  # match_pattern alone: a single pattern matches iff pass with len == rows... use pass when lengths agree
    v = verdicts([pattern], rows)
    n_el = len(F.pattern_from_str(pattern).elements)
    return bool(v[0]["pass"]) if n_el == len(rows) else None


# ---- parser (pattern.rs:389-430, 242-383) --------------------------------------------------------
def test_pattern_macro():
    p = F.pattern_from_str("Ftag[fw, *, @left(0..250)]__Fflank[fw, @prev_left(5..100)]__Rtag[?1, fw, @right(0..20)]")
    e = p.elements
    assert len(e) == 3
    assert (e[0].match_type, e[0].orientation, e[0].label, e[0].placeholder, e[0].range, e[0].relative_to, e[0].cuts) == (0, 0, None, -1, (0, 250), 1, [])
    assert (e[1].match_type, e[1].orientation, e[1].range, e[1].relative_to) == (2, 0, (5, 100), 3)
    assert (e[2].match_type, e[2].placeholder, e[2].range, e[2].relative_to) == (1, 1, (0, 20), 2)


def test_pattern_parse_labels_cuts_and_errors():
    p = F.pattern_from_str('Ftag[rc, "BC14", <<2, >>, @right(10..20)]')
    e = p.elements[0]
    assert e.label == "BC14" and e.orientation == 1 and [str(c) for c in e.cuts] == ["Before(2)", "After(0)"]
    assert F.pattern_from_str("Rflank[~BC, ?7]").elements[0].label == "~BC"
    with pytest.raises(ValueError):
        F.pattern_from_str("Ftag[fw]__Bogus[fw]")          # basic_verify: element count mismatch
    with pytest.raises(ValueError):
        F.pattern_from_str("Flank[fw]")                      # pattern.rs:301
    assert str(F.Cut(3, "After")) == "After(3)"


def test_kit_pattern_sets():
    assert len(F.kit_patterns("SQK-NBD114-96")) == 3 and len(F.kit_patterns("SQK-NBD114-96", True)) == 9
    assert len(F.kit_patterns("SQK-RBK114-24")) == 2 and len(F.kit_patterns("SQK-RBK114-24", True)) == 5


# ---- match_pattern KATs (pattern.rs:432-935) -----------------------------------------------------
def test_distance_to_left_end():
    for start, want in ((0, True), (100, True), (250, True), (251, False)):
        assert is_match("Ftag[fw, *, @left(0..250)]", [mk(0, start, 100, 0, 0)]) is want


def test_distance_to_right_end():
    for end, want in ((500, True), (450, True), (250, True), (249, False)):
        assert is_match("Ftag[fw, *, @right(0..250)]", [mk(0, 0, end, 0, 0)]) is want


def test_distance_to_prev_left():
    pat = "Ftag[fw, *, @left(0..250)]__Fflank[fw, @prev_left(5..100)]"
    for start, want in ((50, False), (100, False), (105, True), (200, True), (201, False)):
        assert is_match(pat, [mk(0, 0, 100, 0, 0), mk(0, start, 200, 2, -1)]) is want


def test_placeholder():
    pat = "Ftag[fw, ?1, @left(0..250)]__Rtag[fw, ?1, @right(0..250)]"
    assert is_match(pat, [mk(0, 0, 100, 0, 0, read_len=250), mk(0, 100, 200, 1, 0, read_len=250)]) is True
    assert is_match(pat, [mk(0, 0, 100, 0, 0, read_len=250), mk(0, 100, 200, 1, 2, read_len=250)]) is False   # label "yyyy"
    pat2 = "Ftag[fw, ?1, @left(0..250)]__Rtag[fw, ?2, @right(0..250)]"                                       # mixed labels
    assert is_match(pat2, [mk(0, 0, 100, 0, 0, read_len=250), mk(0, 100, 200, 1, 0, read_len=250)]) is True


def test_placeholder_not_ordered():
    pat = "Ftag[fw, ?1, @left(0..250)]__Ftag[fw, ?2, @prev_left(0..250)]__Ftag[fw, ?1, @left(0..250)]"
    rows = [mk(0, 0, 100, 0, 0, read_len=600), mk(0, 100, 200, 0, 1, read_len=600), mk(0, 100, 200, 0, 0, read_len=600)]
    assert is_match(pat, rows) is True


def test_pattern_with_cuts():
    rows = [mk(0, 0, 10, 0, 0, read_len=250), mk(0, 15, 20, 2, -1, read_len=250)]
    v = verdicts(["Ftag[fw, *, >>, @left(0..250)]__Fflank[fw, <<, @prev_left(5..100)]"], rows)
    assert v["pass"].all() and [F.format_cuts(x) for x in v] == ["After(0):0", "Before(0):1"]
    v = verdicts(["Ftag[fw, *, >>1, @left(0..250)]__Fflank[fw, <<1, @prev_left(5..100)]"], rows)
    assert [F.format_cuts(x) for x in v] == ["After(1):0", "Before(1):1"]
    v = verdicts(["Ftag[fw, *, >>1, <<2, @left(0..250)]__Fflank[fw, <<1, >>2, @prev_left(5..100)]"], rows)
    assert [F.format_cuts(x) for x in v] == ["After(1):0,Before(2):0", "Before(1):1,After(2):1"]


def test_label_constraints_and_orientation():
    assert is_match('Ftag[fw, "BC14", @left(0..250)]', [mk(0, 0, 50, 0, 3)]) is True
    assert is_match('Ftag[fw, "BC14", @left(0..250)]', [mk(0, 0, 50, 0, 0)]) is False
    assert is_match("Ftag[fw, ~C1, @left(0..250)]", [mk(0, 0, 50, 0, 3)]) is True        # substring (pattern.rs:110-117)
    assert is_match("Ftag[fw, ~C1, @left(0..250)]", [mk(0, 0, 50, 0, 1)]) is False
    assert is_match("Ftag[rc, *, @left(0..250)]", [mk(0, 0, 50, 0, 0, strand=0)]) is False
    assert is_match("Fflank[fw, BC14]", [mk(0, 0, 50, 2, -1)]) is True                     # flanks ignore labels (pattern.rs:123-125)
    assert is_match("Rtag[fw, *]", [mk(0, 0, 50, 0, 0)]) is False                          # match type must be equal


# ---- check_filter_pass (filter.rs:183-214) -------------------------------------------------------
def test_check_filter_pass_longest_pattern_wins_and_all_rows_must_be_consumed():
    pats = ["Ftag[fw, *, @left(0..250), >>]", "Ftag[fw, ?1, @left(0..250)]__Ftag[fw, ?1, @prev_left(0..250), >>]"]
    one = [mk(7, 10, 40, 0, 0)]
    v = verdicts(pats, one)
    assert v[0]["pass"] and F.format_cuts(v[0]) == "After(0):0"
    two_same = [mk(7, 10, 40, 0, 0), mk(7, 60, 90, 0, 0)]
    v = verdicts(pats, two_same)                       # the 2-element pattern is longer and matches
    assert v["pass"].all() and [F.format_cuts(x) for x in v] == ["", "After(0):1"]
    two_diff = [mk(7, 10, 40, 0, 0), mk(7, 60, 90, 0, 1)]
    v = verdicts(pats, two_diff)                       # only the 1-element pattern matches: read has 2 rows -> dropped, cut kept
    assert not v["pass"].any() and [F.format_cuts(x) for x in v] == ["After(0):0", ""]
    # reads are independent groups of consecutive rows
    v = verdicts(pats, one + [mk(8, 300, 340, 0, 0)] + [mk(9, 10, 40, 0, 2), mk(9, 60, 90, 0, 2)])
    assert list(v["pass"]) == [1, 0, 1, 1] and list(v["match_idx"]) == [0, 0, 0, 1]
    assert not verdicts([], one)["pass"].any()          # no patterns: nothing passes


# ---- GPU parity ----------------------------------------------------------------------------------
def _random_patterns(rng, n):
    types = ["Ftag", "Rtag", "Fflank", "Rflank"]
    out = []
    for _ in range(n):
        els = []
        for e in range(int(rng.integers(1, 4))):
            params = [str(rng.choice(["fw", "rc", "*"]))]
            r = rng.random()
            if r < 0.3:
                params.append(f"?{int(rng.integers(1, 3))}")
            elif r < 0.4:
                params.append("~NB1")
            elif r < 0.5:
                params.append("NB07")
            pos = str(rng.choice(["left", "right", "prev_left", ""]))
            if pos:
                lo = int(rng.integers(0, 100))
                params.append(f"@{pos}({lo}..{lo + int(rng.integers(0, 4000))})")
            for _c in range(int(rng.integers(0, 3))):
                params.append(str(rng.choice([">>", "<<", ">>1", "<<2"])))
            els.append(f"{rng.choice(types)}[{', '.join(params)}]")
        out.append("__".join(els))
    return out


@pytest.mark.gpu
def test_filter_gpu_matches_oracle():
    from barbell_amd import annotate as A
    from tests.common import config_groups

    rng = np.random.default_rng(4)
    for cfg, kit in (("nbd96", "SQK-NBD114-96"), ("rbk24", "SQK-RBK114-24"), ("dual", None)):
        groups = config_groups(cfg)
        bases, offsets = A.synth_reads_host(groups, 9, 300, 3000, 0, 3000)
        dm = A.Demuxer()
        for g in groups:
            dm.add_query_group(g)
        rows = dm.demux_packed(bases, offsets)
        orc = po.Oracle([g.as_tuple() for g in groups])
        sets = [[F.pattern_from_str(p) for p in _random_patterns(rng, 12)]]
        if kit:
            sets += [F.kit_patterns(kit, False), F.kit_patterns(kit, True)]
        for pats in sets:
            got = F.Filter(dm, pats).verdicts(rows)
            want = orc.filter_rows(pats, groups, rows)
            assert got.tobytes() == want.tobytes()
        if kit:
            assert got["pass"].mean() > 0.3   # the preset patterns keep a good share of the synthetic reads


@pytest.mark.gpu
def test_filtered_tsv_and_device_pointer_variant(tmp_path):
    import torch

    from barbell_amd import annotate as A
    from tests.common import config_groups

    groups = config_groups("nbd96")
    bases, offsets = A.synth_reads_host(groups, 10, 500, 2500, 0, 800)
    ids = [f"r{i}" for i in range(800)]
    dm = A.Demuxer()
    for g in groups:
        dm.add_query_group(g)
    rows = dm.demux_packed(bases, offsets)
    flt = F.Filter(dm, F.kit_patterns("SQK-NBD114-96", True))
    v = flt.verdicts(rows)
    d_rows = torch.from_numpy(rows.view(np.uint8).copy()).cuda()
    d_out = torch.empty(len(rows) * 16, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    flt.verdicts_dev(d_rows.data_ptr(), len(rows), d_out.data_ptr())
    assert d_out.cpu().numpy().tobytes() == v.tobytes()
    lines = A.format_rows(rows[v["pass"] == 1], ids, groups, v[v["pass"] == 1])
    assert len(lines) == int(v["pass"].sum()) > 100
    assert any(l.endswith("\tAfter(0):0") for l in lines) and all(l.count("\t") == 14 for l in lines)


@pytest.mark.gpu
def test_annotate_with_filter_files(tmp_path):
    """annotate -> filter fused: filtered.tsv + dropped.tsv partition annotation.tsv's reads, rows of a
    read stay together, and the kept rows carry cuts."""
    import gzip

    from barbell_amd import annotate as A
    from barbell_amd import kits

    groups = kits.groups_from_kit("SQK-NBD114-96", flank_max_errors=3)
    bases, offsets = A.synth_reads_host(groups, 21, 400, 2000, 0, 600)
    fq = tmp_path / "r.fastq.gz"
    with gzip.open(fq, "wb") as f:
        for i in range(600):
            sq = bytes(bases[int(offsets[i]):int(offsets[i + 1])])
            f.write(b"@q%d\n" % i + sq + b"\n+\n" + b"I" * len(sq) + b"\n")
    a, k, d = tmp_path / "a.tsv", tmp_path / "k.tsv", tmp_path / "d.tsv"
    A.annotate([str(fq)], str(a), groups, filter_patterns=F.kit_patterns("SQK-NBD114-96", True), filtered_file=str(k),
               dropped_file=str(d), batch_reads=256)
    la, lk, ld = [x.read_text().splitlines() for x in (a, k, d)]
    assert la[0] == lk[0] == ld[0] == A.TSV_HEADER
    strip = lambda l: l.rsplit("\t", 1)[0]
    assert sorted(strip(x) for x in lk[1:] + ld[1:]) == sorted(strip(x) for x in la[1:])
    ids_k, ids_d = {x.split("\t")[0] for x in lk[1:]}, {x.split("\t")[0] for x in ld[1:]}
    assert ids_k and ids_d and not (ids_k & ids_d)
    assert any("After(0):" in x for x in lk[1:])
