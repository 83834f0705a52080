"""Pins the CPU oracle against every known-answer test the reference holds for the sassy boundary
(src/annotate/cigar_parse.rs:104-176).  Inputs and expected values are the reference's test vectors."""
from oracle import pyoracle as po

P = b"AAAAACCCAAAA"
# the five vectors as data (pattern, text, k, (-, text span of pattern rows [5, 8) or None, sub-path cost)); the first match of
# Searcher::<Iupac>::new_rc().search is the one the reference's tests look at
KATS = [
    (P, b"GGGGAAAAACCCAAAAGGGGG", 0, (None, None, 0)),
    (P, b"GGGGAAAAACGCAAAA", 1, (None, None, 1)),
    (P, b"ACGCAAAAGGGGGGGGGGGG", 5, (None, (1, 4), 1)),
    (P, b"GAAAAACGC", 5, (None, (6, 9), 1)),
    (P, b"GCAAAAGGGGGGGGGGGG", 8, (None, (0, 2), 2)),
]


def revcomp(s):  # cigar_parse.rs:90-102
    t = {65: 84, 84: 65, 67: 71, 71: 67}
    return bytes(t.get(c, 78) for c in reversed(s))


def first_sub(p, t, k):
    ms, h = po.search(p, t, k, alpha=None, rc=True)  # Searcher::<Iupac>::new_rc()
    assert ms, "no match"
    r = po.map_pat_to_text_with_cost(h, 0, 5, 7 + 1)
    po.free_matches(h)
    return ms[0], r


def test_cost_extraction_no_edits():  # cigar_parse.rs:104-123
    t = b"GGGGAAAAACCCAAAAGGGGG"
    m, (_, _, cost) = first_sub(P, t, 0)
    assert cost == 0 and m.strand == 0 and m.cost == 0
    m, (_, _, cost) = first_sub(revcomp(P), revcomp(t), 0)
    assert cost == 0


def test_cost_extraction_1_edits():  # :125-135
    _, (_, _, cost) = first_sub(P, b"GGGGAAAAACGCAAAA", 1)
    assert cost == 1


def test_overhang_left_flank():  # :137-148
    m, (_, (ts, te), cost) = first_sub(P, b"ACGCAAAAGGGGGGGGGGGG", 5)
    assert (cost, ts, te) == (1, 1, 4)
    assert m.strand == 0


def test_overhang_right_flank():  # :150-161
    _, (_, (ts, te), cost) = first_sub(P, b"GAAAAACGC", 5)
    assert (cost, ts, te) == (1, 6, 9)


def test_overhang_including_bar():  # :163-176
    m, (_, (ts, te), cost) = first_sub(P, b"GCAAAAGGGGGGGGGGGG", 8)
    assert (cost, ts, te) == (2, 0, 2)
    assert m.strand == 0  # forward match is returned first
