"""Property tests of the CPU oracle itself (independent of the GPU):
  * reported costs equal a brute-force semi-global edit distance written independently in Python;
  * the (m+k)-column traceback window gives exactly the matches of the full DP matrix;
  * CIGAR/path invariants; Lodhi score against a brute-force triple sum."""
import itertools
import random

import numpy as np

from oracle import pyoracle as po


def brute_costs(p, t):
    """C[i] = min edit distance of p against any substring of t ending at i (unit costs, ACGT)."""
    m, n = len(p), len(t)
    prev = list(range(m + 1))
    out = [m]
    for i in range(1, n + 1):
        cur = [0] * (m + 1)
        for j in range(1, m + 1):
            cur[j] = min(prev[j - 1] + (p[j - 1] != t[i - 1]), prev[j] + 1, cur[j - 1] + 1)
        out.append(cur[m])
        prev = cur
    return out


def local_minima(C, k):
    res, dec, prev = [], True, C[0]
    for i in range(1, len(C)):
        if C[i] > prev:
            if dec and prev <= k:
                res.append((i - 1, prev))
            dec = False
        elif C[i] < prev:
            dec = True
        prev = C[i]
    if dec and prev <= k:
        res.append((len(C) - 1, prev))
    return res


def rand_seq(rng, n):
    return bytes(rng.choice(b"ACGT") for _ in range(n))


def test_search_against_brute_force():
    rng = random.Random(11)
    for _ in range(150):
        m, n, k = rng.randint(3, 20), rng.randint(0, 60), rng.randint(0, 6)
        p = rand_seq(rng, m)
        t = bytearray(rand_seq(rng, n))
        if n > m and rng.random() < 0.7:  # plant a noisy copy
            pos = rng.randint(0, n - m)
            t[pos:pos + m] = p
            for _ in range(rng.randint(0, 3)):
                t[rng.randrange(n)] = rng.choice(b"ACGT")
        t = bytes(t)
        ms, h = po.search(p, t, k, alpha=None, rc=False)
        want = local_minima(brute_costs(p, t), k)
        assert [(x.text_end, x.cost) for x in ms] == want, (p, t, k)
        for x in ms:  # CIGAR invariants
            nonmatch = sum(1 for o in x.ops if o != 0)
            assert nonmatch == x.cost
            assert sum(1 for o in x.ops if o != 2) == m                      # every pattern char consumed once
            assert sum(1 for o in x.ops if o != 3) == x.text_end - x.text_start
            assert [i for i, _ in x.path] == sorted(i for i, _ in x.path)
        po.free_matches(h)


def test_window_trace_equals_full_matrix_trace():
    rng = random.Random(5)
    L = po.lib()
    for trial in range(60):
        m, n, k = rng.randint(8, 40), rng.randint(100, 400), rng.randint(0, 8)
        p = rand_seq(rng, m)
        t = bytearray(rand_seq(rng, n))
        for _ in range(3):
            pos = rng.randint(0, n - m)
            t[pos:pos + m] = p
            for _ in range(rng.randint(0, 4)):
                t[rng.randrange(n)] = rng.choice(b"ACGTN")
        t = bytes(t)
        for alpha in (None, 0.4):
            L.bbo_set_full_trace(0)
            a, ha = po.search(p, t, k, alpha=alpha, rc=True)
            L.bbo_set_full_trace(1)
            b, hb = po.search(p, t, k, alpha=alpha, rc=True)
            L.bbo_set_full_trace(0)
            assert len(a) == len(b)
            for x, y in zip(a, b):
                assert (x.text_start, x.text_end, x.cost, x.strand, x.ops, x.path) == (y.text_start, y.text_end, y.cost, y.strand, y.ops, y.path)
            po.free_matches(ha)
            po.free_matches(hb)


def test_rc_match_mirrors_forward_match():
    rng = random.Random(9)
    comp = bytes.maketrans(b"ACGT", b"TGCA")
    for _ in range(40):
        m, n = rng.randint(10, 30), rng.randint(60, 120)
        p, t = rand_seq(rng, m), bytearray(rand_seq(rng, n))
        pos = rng.randint(0, n - m)
        t[pos:pos + m] = p.translate(comp)[::-1]  # plant rc(p)
        ms, h = po.search(p, bytes(t), 1, alpha=0.4, rc=True)
        hit = [x for x in ms if x.strand == 1 and x.cost == 0 and len(x.ops) == m]
        assert hit and hit[0].text_start == pos and hit[0].text_end == pos + m
        # path: pattern index ascending, text index descending from text_end-1
        assert hit[0].path[0] == (0, pos + m - 1) and hit[0].path[-1] == (m - 1, pos)
        po.free_matches(h)


def test_overhang_costs():
    p = b"ACGTACGTAGCTAGCTAGGA"  # m = 20
    # left overhang: first 5 pattern chars missing -> cost floor(0.4*5) = 2
    ms, h = po.search(p, p[5:] + b"TTTTTTTTTTTTTTTTTTTTTTTTT", 3, alpha=0.4, rc=False)
    assert any(x.cost == 2 and x.text_start == 0 and x.pattern_start == 5 and x.text_end == 15 for x in ms)
    po.free_matches(h)
    # right overhang: last 7 missing -> floor(2.8) = 2, match clamped to the text end
    t = b"TTTTTTTTTTTTTTTTTTTTTTTTT" + p[:13]
    ms, h = po.search(p, t, 3, alpha=0.4, rc=False)
    assert any(x.cost == 2 and x.text_end == len(t) and x.pattern_end == 13 for x in ms)
    po.free_matches(h)
    # without overhang the same text needs 5 pattern-only ops
    ms, h = po.search(p, p[5:] + b"TTTTTTTTTTTTTTTTTTTTTTTTT", 5, alpha=None, rc=False)
    assert any(x.cost == 5 and x.cigar.startswith("DDDDD") for x in ms)
    po.free_matches(h)


def test_lodhi_against_brute_force_triples():
    rng = random.Random(3)
    for _ in range(50):
        n = rng.randint(0, 40)
        ops = bytes(rng.choice([0, 0, 0, 1, 2, 3]) for _ in range(n))
        idx = [i for i, o in enumerate(ops) if o == 0]
        want = sum(0.5 ** (c - a + 1) for a, b, c in itertools.combinations(idx, 3))
        got = po.lodhi(ops)
        assert abs(got - want) <= 1e-12 * max(1.0, want)
    assert po.lodhi(bytes(3)) == 0.125
    assert po.lodhi(b"\x00\x01\x00\x00") == 0.0625


def test_demux_rows_sorted_and_types():
    from barbell_amd import annotate as A
    from tests.common import config_groups

    groups = config_groups("dual")
    b, o = A.synth_reads_host(groups, 77, 800, 1600, 0, 120)
    rows = po.Oracle([g.as_tuple() for g in groups]).annotate(b, o, n_threads=4)
    key = rows["read_idx"].astype(np.int64) * (1 << 32) + rows["read_start_flank"]
    assert (np.diff(key) >= 0).all()
    assert set(np.unique(rows["group_idx"]).tolist()) == {0, 1}
    tags = rows[rows["barcode_idx"] >= 0]
    assert ((tags["group_idx"] == 0) == (tags["match_type"] == 0)).all()  # Ftag for group 0, Rtag for group 1
    assert len(tags) > 60
