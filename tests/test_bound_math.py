"""The fast path's score bound, as arithmetic (no GPU): for random op strings the bound k_barcode_lane computes over the pattern's ROWS —
Ins ops dropped, a row without a Match taking min(eS, eD) (lodhi_bound_table_entry) or its class's exponent (lodhi_bound_table_entry4:
Match / Sub / Del / unknown) — and the one k_barcode_pfx computes over the text's COLUMNS (Del ops dropped, min(eS, eI)) are upper bounds
of cigar-lodhi's score as the checker computes it (oracle/bb_oracle.c lodhi_pol), under every decay-exponent policy the tests use.
The recurrence below is the kernels' table recurrence (barbell_amd/csrc/bb_k_bar_common.h) written out per row in float64."""
import zlib

import numpy as np
import pytest

from oracle import pyoracle as po

M, S, I, D = 0, 1, 2, 3   # BBO_MATCH, BBO_SUB, BBO_INS, BBO_DEL (oracle/bb_oracle.h)


def bound(classes, exps, lam=0.5):
    """sum over triples of matched positions of lam^(time span), time advancing by the class's exponent: the affine recurrence of the
    tables, one position at a time (sc += w P2; P2 += P1; P1 += 2^(s - eM))"""
    sc = p1 = p2 = 0.0
    s = 0
    for c in classes:
        if c == "M":
            s += exps["M"]
            w = lam ** s
            sc += w * p2
            p2 += p1
            p1 += lam ** -(s - exps["M"])
        else:
            s += exps[c]
    return sc


def exact(ops, pol):
    with po.policy(pol):
        return po.lodhi(bytes(ops))


@pytest.mark.parametrize("pol", ["lodhi=3:0.5:1111", "lodhi=3:0.5:2211", "lodhi=3:0.5:1110", "lodhi=3:0.5:1011", "lodhi=3:0.5:2131", "lodhi=3:0.5:1121", "lodhi=3:0.5:2012"])
def test_row_and_column_bounds_are_upper_bounds(pol):
    eM, eS, eI, eD = (int(ch) for ch in pol.split(":")[-1])
    rng = np.random.default_rng(zlib.crc32(pol.encode()))
    for _ in range(400):
        n = int(rng.integers(20, 64))
        ops = rng.choice([M, S, I, D], size=n, p=[0.7, 0.12, 0.08, 0.10]).astype(np.uint8)
        ex = exact(ops, pol)
        rows = [o for o in ops if o != I]                     # one op per pattern row
        cols = [o for o in ops if o != D]                     # one op per text column
        name = {M: "M", S: "S", D: "D", I: "I"}
        b_rows2 = bound(["M" if o == M else "X" for o in rows], {"M": eM, "X": min(eS, eD)})
        b_rows4 = bound([name[o] for o in rows], {"M": eM, "S": eS, "D": eD})
        # unknown shared rows: any non-Match row may be told as "X" (the smaller exponent) instead of its class
        mixed = [("X" if (name[o] != "M" and rng.random() < 0.3) else name[o]) for o in rows]
        b_rows4x = bound(mixed, {"M": eM, "S": eS, "D": eD, "X": min(eS, eD)})
        b_cols = bound(["M" if o == M else "X" for o in cols], {"M": eM, "X": min(eS, eI)})
        for b in (b_rows2, b_rows4, b_rows4x, b_cols):
            assert b >= ex * (1.0 - 1e-12), (pol, ops.tolist(), b, ex)
        assert b_rows4 <= b_rows4x * (1.0 + 1e-12) <= b_rows2 * (1.0 + 1e-9) + 1e-300   # finer classes, tighter bound


@pytest.mark.parametrize("pol", ["lodhi=3:0.5:1111", "lodhi=3:0.5:2211", "lodhi=3:0.5:1110", "lodhi=3:0.5:1011", "lodhi=3:0.5:2131", "lodhi=3:0.5:1121", "lodhi=3:0.5:2012"])
def test_granting_shared_rows_as_matches_keeps_the_bound_an_upper_bound(pol):
    """the non-NM k_barcode_lane bound does not know what the path did in the P leading rows every barcode shares — it GRANTS them
    as Matches (the first table steps are the same for every barcode: lodhi_bound_first_rows4 / first_byte, computed once per block).  Where
    eM > min(eS, eD) a granted Match advances the time by MORE than the row's real op did, so it is not obvious that granting can only raise
    the sum: checked here as arithmetic — any subset of the leading non-Match rows told as 'M', all of them (the kernel's case), under all
    seven exponent policies, with both forms of the rest of the rows (two classes / four classes)."""
    eM, eS, eI, eD = (int(ch) for ch in pol.split(":")[-1])
    rng = np.random.default_rng(zlib.crc32(pol.encode()) ^ 0x5A5A)
    name = {M: "M", S: "S", D: "D", I: "I"}
    worst = np.inf
    for _ in range(600):
        n = int(rng.integers(20, 64))
        # leading rows with more damage than a real pad sees: the grant is furthest from the truth there
        p_lead = [0.45, 0.25, 0.05, 0.25] if rng.random() < 0.5 else [0.7, 0.12, 0.08, 0.10]
        P = int(rng.integers(1, 17))
        lead = rng.choice([M, S, I, D], size=P + 4, p=p_lead).astype(np.uint8)
        rest = rng.choice([M, S, I, D], size=n, p=[0.7, 0.12, 0.08, 0.10]).astype(np.uint8)
        ops = np.concatenate([lead, rest])
        ex = exact(ops, pol)
        rows = [o for o in ops if o != I]
        for grant_all in (True, False):
            g = [("M" if (i < P and (grant_all or rng.random() < 0.5)) else name[o]) for i, o in enumerate(rows)]
            b4 = bound(g, {"M": eM, "S": eS, "D": eD})
            b2 = bound(["M" if c == "M" else "X" for c in g], {"M": eM, "X": min(eS, eD)})
            for b in (b4, b2):
                assert b >= ex * (1.0 - 1e-12), (pol, P, grant_all, ops.tolist(), b, ex)
                if ex > 0:
                    worst = min(worst, b / ex)
    assert worst >= 1.0 - 1e-12
