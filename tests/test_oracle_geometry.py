"""Geometry / collapse / label known-answer tests transcribed from the reference's own unit tests
(src/annotate/barcodes.rs:465-546, src/annotate/interval.rs:154-256, src/kits/kits.rs:1109-1182,
src/annotate/edit_model.rs) run against the CPU oracle and the host-side kit loader."""
import os
import random

import numpy as np
import pytest

from barbell_amd import _abi, kits
from oracle import pyoracle as po

EX = os.path.join(os.path.dirname(__file__), "golden", "examples")


def test_barcode_group():  # barcodes.rs:487-504
    o = po.Oracle([([b"AAATTTGGG", b"AAACCCGGG"], _abi.BB_FTAG, None)])
    i = o.info(0)
    assert o.flank(0) == b"AAANNNGGG"
    assert (i.bar_lo, i.bar_hi) == (3, 5)
    assert o.pattern(0, 0) == b"AAATTTGGG" and o.pattern(0, 1) == b"AAACCCGGG"
    assert o.pattern(0, 0, rc=True) == b"CCCAAATTT"


def test_group_errors():  # barcodes.rs:506-530, :113-133
    with pytest.raises(ValueError, match=str(_abi.BB_E_NOT_IUPAC)):
        po.Oracle([([b"@@@@@@@@@", b"AAACCCGGG"], 0, None)])
    with pytest.raises(ValueError, match=str(_abi.BB_E_UNEQUAL_LEN)):
        po.Oracle([([b"AAATTTGGG", b"AAAAAAACCCGGG"], 0, None)])
    with pytest.raises(ValueError, match=str(_abi.BB_E_ONE_QUERY)):
        po.Oracle([([b"AAATTTGGG"], 0, None)])
    with pytest.raises(ValueError, match=str(_abi.BB_E_NO_FLANK)):
        po.Oracle([([b"CAATTTGGT", b"AAACCCGGG"], 0, None)])
    with pytest.raises(ValueError, match=str(_abi.BB_E_NO_BARCODE)):
        po.Oracle([([b"AAACCCGGG", b"AAACCCGGG"], 0, None)])


def test_fasta_read():  # barcodes.rs:532-546
    g = kits.group_from_fasta(os.path.join(EX, "rapid_bars.fasta"))
    o = po.Oracle([g.as_tuple()])
    i = o.info(0)
    assert o.flank(0) == b"GCTTGGGTGTTTAACCNNNNNNNNNNNNNNNNNNNNNNNNGTTTTCGCATTTATCGTGAAACGCTTTCGCGTTTTTCGTGCGCCGCTTCA"
    assert (i.bar_lo, i.bar_hi) == (16, 39)
    assert len(g.seqs) == 96
    assert o.pattern(0, 0)[10:34] == b"AAGAAAGTTGTCGGTGTCTTTGTG"
    # SURVEY §8 geometry table
    assert (i.pad_lo, i.pad_hi, i.pattern_len, i.bar_k1, i.bar_k2, i.flank_k) == (6, 50, 44, 17, 44, 20)


def test_nbd114_96_geometry():  # SURVEY §8 row 2 (kits.rs:311-316, 945-1042)
    g = kits.groups_from_kit("SQK-NBD114-96", flank_max_errors=3)
    assert len(g) == 1 and len(g[0].seqs) == 96 and g[0].labels[0] == "NB01" and g[0].labels[95] == "NB96"
    o = po.Oracle([x.as_tuple() for x in g])
    i = o.info(0)
    assert (i.flank_len, i.prefix_len, i.mask_len, i.suffix_len) == (46, 14, 24, 8)
    assert (i.bar_lo, i.bar_hi, i.pad_lo, i.pad_hi, i.pattern_len) == (14, 37, 4, 48, 42)
    assert (i.bar_k1, i.bar_k2, i.flank_k) == (16, 42, 3)
    assert po.Oracle([(g[0].seqs, 0, None)]).info(0).flank_k == 4  # auto cutoff for 22 flank bases
    # --use-extended is a no-op for this kit, and adds a second group for SQK-RBK114-96
    assert len(kits.groups_from_kit("SQK-NBD114-96", use_extended=True)) == 1
    assert len(kits.groups_from_kit("SQK-RBK114-96", use_extended=True)) == 2
    assert len(kits.groups_from_kit("SQK-RBK114-96", use_extended=False)) == 1


def test_dual_end_geometry():
    gl = kits.group_from_fasta(os.path.join(EX, "native_left.fasta"), _abi.BB_FTAG, 5)
    gr = kits.group_from_fasta(os.path.join(EX, "native_right.fasta"), _abi.BB_RTAG, 5)
    o = po.Oracle([gl.as_tuple(), gr.as_tuple()])
    a, b = o.info(0), o.info(1)
    assert (a.flank_len, a.prefix_len, a.mask_len, a.suffix_len, a.pattern_len) == (76, 44, 24, 8, 42)
    assert (b.flank_len, b.prefix_len, b.mask_len, b.suffix_len, b.pattern_len) == (67, 7, 24, 36, 41)
    assert (a.bar_lo, a.bar_hi, a.pad_lo, a.pad_hi) == (44, 67, 34, 78)
    assert (b.bar_lo, b.bar_hi, b.pad_lo, b.pad_hi) == (7, 30, 0, 41)


def test_edit_cut_off():  # edit_model.rs:2-11 ; SURVEY §8 table
    f = po.lib().bbo_edit_cut_off
    assert [f(l) for l in (0, 1, 22, 43, 66, 52)] == [0, 0, 4, 11, 20, 15]


def test_rel_dist_to_end():  # searcher.rs:183-199
    f = po.lib().bbo_rel_dist_to_end
    assert [f(p, 100) for p in (-3, 0, 7, 50, 51, 100, 99)] == [1, 1, 7, 50, -49, -1, -1]


# ---- kits.rs:1109-1182 label tests -------------------------------------------------------------
def test_get_barcodes():
    assert kits.get_barcodes("BC01", "BC12") == [f"BC{i:02d}" for i in range(1, 13)]
    v = kits.get_barcodes("BC01", "BC12", True)
    assert v[11] == "BC12A" and v[0] == "BC01"
    v = kits.get_barcodes("NB01", "NB24")
    assert v[0] == "NB01" and v[23] == "NB24" and len(v) == 24
    v = kits.get_barcodes("RBK01", "RBK96")
    assert v[25] == "RBK26" and v[38] == "RBK39" and v[0] == "BC01" and v[59] == "RBK60" and len(v) == 96
    assert kits.get_barcodes("NB13", "NB24")[0] == "NB13"
    assert kits.lookup_barcode_seq("BC12A") == "GTTGAGTTACAAAGCACCGATCAG"
    assert kits.lookup_barcode_seq("RBK26") == "ACTATGCCTTTCCGTGAAACAGTT"
    assert kits.lookup_barcode_seq("NB01") == "CACAAAGACACCGACAACTTTCTT"
    assert kits.lookup_barcode_seq("BC01") == "AAGAAAGTTGTCGGTGTCTTTGTG"


# ---- interval.rs:154-256 -----------------------------------------------------------------------
def tmpl(start, end, mtype, bcost, label):
    r = np.zeros(1, dtype=_abi.ROW_DTYPE)[0]
    r["read_start_bar"], r["read_end_bar"] = start, end
    r["read_start_flank"], r["read_end_flank"] = start, end
    r["bar_end"] = 10
    r["match_type"], r["barcode_cost"], r["barcode_idx"], r["read_len"] = mtype, bcost, label, 100
    return r


def coll(rows, thr):
    return po.collapse(np.array(rows, dtype=_abi.ROW_DTYPE), thr) if rows else np.zeros(0, dtype=_abi.ROW_DTYPE)


def test_collapse_basic():
    assert len(coll([], 0.5)) == 0
    r = coll([tmpl(0, 10, 0, 3, 1)], 0.5)
    assert len(r) == 1 and r[0]["barcode_idx"] == 1
    r = coll([tmpl(0, 10, 0, 3, 1), tmpl(10, 20, 0, 3, 2)], 0.5)
    assert list(r["barcode_idx"]) == [1, 2]
    r = coll([tmpl(0, 20, 0, 0, 1), tmpl(15, 20, 0, 3, 2)], 0.5)
    assert list(r["barcode_idx"]) == [1]


def test_overlap_threshold():
    m = [tmpl(0, 20, 0, 0, 1), tmpl(10, 35, 0, 3, 2)]
    assert list(coll(m, 0.5)["barcode_idx"]) == [1]
    assert list(coll(m, 0.6)["barcode_idx"]) == [1, 2]


def test_correct_sorting():
    m = [tmpl(0, 10, 0, 0, 1), tmpl(10, 20, 0, 3, 2), tmpl(0, 15, 0, 3, 2), tmpl(100, 110, 0, 3, 3)]
    rng = random.Random(1)
    for _ in range(10):
        rng.shuffle(m)
        assert list(coll(m, 0.5)["barcode_idx"]) == [1, 3]


def test_small_overlap():
    m = [tmpl(0, 10, 0, 3, 1), tmpl(10, 20, 0, 1, 2)]
    for _ in range(4):
        m[1]["read_start_flank"] -= 1
        m[1]["read_end_flank"] -= 1
        assert list(coll(m, 0.5)["barcode_idx"]) == [1, 2]
    m[1]["read_start_flank"] -= 1
    m[1]["read_end_flank"] -= 1
    assert list(coll(m, 0.5)["barcode_idx"]) == [2]


def test_tags_beat_flanks_and_longest_flank():  # interval.rs:47-76
    r = coll([tmpl(0, 20, _abi.BB_FFLANK, 42, -1), tmpl(1, 20, _abi.BB_FTAG, 5, 7)], 0.8)
    assert list(r["barcode_idx"]) == [7]
    r = coll([tmpl(0, 20, _abi.BB_FFLANK, 42, 1), tmpl(0, 22, _abi.BB_RFLANK, 42, 2)], 0.8)
    assert list(r["barcode_idx"]) == [2]
