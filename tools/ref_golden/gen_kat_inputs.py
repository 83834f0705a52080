#!/usr/bin/env python3
"""gen_kat_inputs.py — writes tools/ref_golden/kat_inputs.tsv, the inputs `kat.rs` feeds to the real crates.

The first block is hand-made: for every alternative of every switchable assumption (include/barbell_amd_policy.h, hazards
H1-H4, H7, H8) at least one input on which that alternative gives a DIFFERENT answer than the others
(tests/test_ref_golden.py::test_kat_inputs_discriminate_every_alternative proves it with the CPU checker), so that the
crates' answers single out one policy (tools/ref_fit.py).  The second block is the pseudo-random bulk of rounds 1-2
(flank searches on the SQK-NBD114-96 and a 90-nt rapid flank, barcode-set searches, CIGARs).

  lodhi       <ops>                                  ops over = X I D (pa-types Match / Sub / Ins / Del)
  search      <alpha | -1> <k> <pattern> <text>      Searcher::<Iupac>::new_rc() / new_rc_with_overhang(alpha) .search
  search_set  <k> <text> <pat,pat,...>               new_fwd().encode_patterns + new_rc().search_encoded_patterns
"""
import os

HERE = os.path.dirname(os.path.abspath(__file__))


def lcg(st):
    st[0] = (st[0] * 6364136223846793005 + 1442695040888963407) & 0xFFFFFFFFFFFFFFFF
    return st[0] >> 33


def rand_seq(st, n):
    return "".join("ACGT"[lcg(st) & 3] for _ in range(n))


def mutate(st, s, n):
    v = list(s)
    for _ in range(n):
        if not v:
            break
        p = lcg(st) % len(v)
        kind = lcg(st) % 3
        if kind == 0:
            v[p] = "ACGT"[lcg(st) & 3]
        elif kind == 1:
            v.insert(p, "ACGT"[lcg(st) & 3])
        else:
            del v[p]
    return "".join(v)


def revcomp(s):
    return s[::-1].translate(str.maketrans("ACGT", "TGCA"))


def discriminators():
    L = []
    # [H8] which ops stretch a span, by how much, and the subsequence length
    for ops in ("=", "==", "===", "====", "=X==", "=I==", "=D==", "=XX==", "=II==", "=DD==", "==X=I=D==", "=" * 41, "=" * 42, "=" * 44,
                "X===", "===X", "D===", "===D", "I===", "=IDI=D==X=="):
        L.append(("lodhi", ops))
    # [H1] plateaus of the end-position cost: AAAA in a run of five A; a strict minimum; a plateau at the text's end
    for k in (0, 1):
        L.append(("search", -1, k, "AAAA", "GGAAAAAGG"))
        L.append(("search", -1, k, "ACGT", "GGACGTGG"))
        L.append(("search", -1, k, "ACGTT", "GGGACGTTT"))
        L.append(("search", -1, k, "AAAA", "CCAAAAAA"))
    L.append(("search", -1, 1, "ACGTACGT", "TTTACGTACGTACGTTTT"))       # periodic: several minima, costs 0 and 0
    # [H2] two reverse-complement occurrences (and two forward ones) in one text
    L.append(("search", -1, 0, "ACGTTG", "TTTTCAACGTTTTTTTTTCAACGTTTTT"))
    L.append(("search", -1, 1, "ACGTTG", "TTACGTTGTTTTCAACGTTTTTTACGTTGTTTCAACGTTT"))
    # [H3] one extra / one missing / one wrong character inside a run: where the gap sits is the traceback's preference
    for pat, text in (("ACCGTT", "GGACCCGTTGG"), ("ACCCGTT", "GGACCGTTGG"), ("ACGTAC", "GGACTTACGG"), ("AACCGGTT", "TTAACGGTTAA"),
                      ("ACGGT", "TTACGTGTT"), ("TTACGGCATT", "GGTTACGCATTGG"), ("GATTACA", "CCGATACACC"), ("GATTACA", "CCGATTTACACC")):
        for k in (1, 2):
            L.append(("search", -1, k, pat, text))
    # [H4] characters hanging over either end at alpha per character: 1..6 of them, alpha where floor / ceil / nearest and
    # the f32 / f64 products differ (0.5: 0.5, 1.5, 2.5 tie to even; 0.7f * 10 = 7.0 in f32, 6.99999988 in f64; 0.4f * 5)
    P = "ACGTTGCATGCATCAGGA"
    for alpha in (0.5, 0.7, 0.4, 0.3, 1.0, 0.0):
        for o in (1, 2, 3, 5, 6, 10):
            L.append(("search", alpha, 8, P, P[o:] + "TTTTTTTTTTTT"))     # the first o characters hang over the text's start
            L.append(("search", alpha, 8, P, "TTTTTTTTTTTT" + P[:-o]))    # the last o over its end
        L.append(("search", alpha, 9, "A" * 10 + "CGCG", "CGCG" + "T" * 8))
        L.append(("search", alpha, 9, "CGCG" + "A" * 10, "T" * 8 + "CGCG"))
    # [H7] one barcode pattern with two equally cheap strict minima inside one window (period 4, 8 rows in 12 columns)
    L.append(("search_set", 3, "ACGTACGTACGT", ["ACGTACGT", "ACGTACGA", "TTTTTTTT"]))
    L.append(("search_set", 8, "ACGTACGTACGT", ["ACGTACGT", "ACGTACGA", "TTTTTTTT"]))
    L.append(("search_set", 4, "GGACGTACGTACGTGG", ["CGTACGTA", "ACGTTCGT", "GGGGGGGG"]))
    return L


def bulk():
    st = [0xBA7BE11]
    L = []
    cig = ["=" * 10, "==X==", "==I==", "==D==", "=X=X=X=", "====DDDD====", "====IIII====", "=" * 20 + "X" + "=" * 21, "=" * 10 + "ID" + "=" * 30]
    for _ in range(8):
        n = 30 + lcg(st) % 20
        cig.append("".join("===XID"[lcg(st) % 6] for _ in range(n)))
    L += [("lodhi", c) for c in cig]
    flanks = ["ATTGCTAAGGTTAANNNNNNNNNNNNNNNNNNNNNNNNCAGCACCT",
              "GCTTGGGTGTTTAACCNNNNNNNNNNNNNNNNNNNNNNNNGTTTTCGCATTTATCGTGAAACGCTTTCGCGTTTTTCGTGCGCCGCTTCA"]
    for fi, flank in enumerate(flanks):
        for case in range(12):
            k = (0, 3, 5)[case % 3] if fi == 0 else (5, 20)[case % 2]
            solid = "".join("ACGT"[lcg(st) & 3] if c == "N" else c for c in flank)
            inst = mutate(st, solid, case % 5)
            text = rand_seq(st, lcg(st) % 40)
            if case % 4 == 0:
                text += inst + rand_seq(st, 120)
            elif case % 4 == 1:
                text = inst[len(inst) // 3:] + rand_seq(st, 150)
            elif case % 4 == 2:
                text += rand_seq(st, 100) + inst[: len(inst) * 2 // 3]
            else:
                text += inst + rand_seq(st, 60) + revcomp(inst) + rand_seq(st, 30)
            for alpha in (-1, 0.4, 1.0):
                L.append(("search", alpha, k, flank, text))
    for case in range(10):
        m = (42, 44, 41)[case % 3]
        left, right = rand_seq(st, 10), rand_seq(st, m - 34)
        pats = [left + rand_seq(st, 24) + right for _ in range(12)]
        pick = lcg(st) % len(pats)
        win = rand_seq(st, lcg(st) % 4) + mutate(st, pats[pick], case % 6) + rand_seq(st, lcg(st) % 4)
        for k in (int(m * 0.4), m):
            L.append(("search_set", k, win, pats))
    return L


def all_inputs():
    return discriminators() + bulk()


def format_line(t):
    if t[0] == "lodhi":
        return "lodhi\t" + t[1]
    if t[0] == "search":
        return "search\t%s\t%d\t%s\t%s" % (("-1" if t[1] == -1 else repr(float(t[1]))), t[2], t[3], t[4])
    return "search_set\t%d\t%s\t%s" % (t[1], t[2], ",".join(t[3]))


def parse_line(line):
    f = line.rstrip("\n").split("\t")
    if f[0] == "lodhi":
        return ("lodhi", f[1])
    if f[0] == "search":
        return ("search", -1 if f[1] == "-1" else float(f[1]), int(f[2]), f[3], f[4])
    return ("search_set", int(f[1]), f[2], f[3].split(","))


def load(path=os.path.join(HERE, "kat_inputs.tsv")):
    return [parse_line(l) for l in open(path) if l.strip() and not l.startswith("#")]


if __name__ == "__main__":
    with open(os.path.join(HERE, "kat_inputs.tsv"), "w") as f:
        f.write("# written by gen_kat_inputs.py; read by kat.rs (real crates) and by tools/ref_fit.py (the CPU checker)\n")
        for t in all_inputs():
            f.write(format_line(t) + "\n")
    print(len(all_inputs()), "inputs")
