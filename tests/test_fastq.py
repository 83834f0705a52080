"""FASTQ ingest.  CPU: the scalar block parser of the oracle on hand-made blocks (line ends,
partial blocks, header split of io.rs:6-17 incl. the reference's own split_fastq_header tests, malformed
records).  GPU: bb_fastq_ingest == oracle on the same blocks and on FASTQ text of synthetic reads cut into
blocks at arbitrary byte positions; the ingested batch feeds annotate/trim without visiting the host."""
import ctypes as C

import numpy as np
import pytest

from barbell_amd import _abi
from oracle import pyoracle as po

REC = [(b"r1 runid=ab ch=3", b"ACGTACGT", b"IIIIHHHH"), (b"r2", b"GG", b"@+"), (b"r3\t  two  words ", b"ACGTN", b"!!!!!"),
       (b"r4 ", b"", b""), (b"", b"A", b"#")]


def fq(recs, nl=b"\n", last_nl=True):
    t = b"".join(b"@" + h + nl + s + nl + b"+" + nl + q + nl for h, s, q in recs)
    return t if last_nl else t[: -len(nl)]


def check(arr, recs):
    n = len(recs)
    off, hoff = arr["offsets"], arr["hdr_offsets"]
    assert len(off) == n + 1 and len(hoff) == n + 1
    for i, (h, s, q) in enumerate(recs):
        assert arr["bases"][int(off[i]):int(off[i + 1])].tobytes() == s
        assert arr["quals"][int(off[i]):int(off[i + 1])].tobytes() == q
        assert arr["hdr"][int(hoff[i]):int(hoff[i + 1])].tobytes() == h
        parts = h.split(None, 1)
        rid = parts[0] if parts and not h[:1].isspace() else b""
        assert int(arr["id_len"][i]) == len(rid)
        desc = h[len(rid):].lstrip()
        assert h[int(arr["desc_start"][i]):] == desc


def test_blocks_and_line_ends():
    for nl in (b"\n", b"\r\n"):
        for last_nl in (True, False):
            rc, info, arr = po.fastq_parse(fq(REC, nl, last_nl), True)
            assert rc == 0 and info["n_records"] == 5 and info["bad_record"] == -1 and info["consumed"] == len(fq(REC, nl, last_nl))
            check(arr, REC)
    # trailing blank lines are fine at the end of the stream
    rc, info, arr = po.fastq_parse(fq(REC) + b"\n\r\n\n", True)
    assert rc == 0 and info["n_records"] == 5
    # non-final block: the partial tail is left to the caller
    text = fq(REC)
    for cut in (1, 5, len(fq(REC[:2])), len(fq(REC[:2])) + 3, len(text) - 1):
        rc, info, arr = po.fastq_parse(text[:cut], False)
        k = sum(1 for i in range(1, 6) if len(fq(REC[:i])) <= cut)
        assert rc == 0 and info["n_records"] == k and info["consumed"] == len(fq(REC[:k]))
        if k:
            check(arr, REC[:k])
    rc, info, _ = po.fastq_parse(b"", True)
    assert rc == 0 and info["n_records"] == 0


def test_split_fastq_header_cases():  # io.rs:40-58 tests
    for h, rid, desc in ((b"read1 runid=abc sample=xyz", b"read1", b"runid=abc sample=xyz"), (b"read1", b"read1", b""),
                         (b"read1   lots of space", b"read1", b"lots of space"), (b"read1\tx", b"read1", b"x")):
        rc, info, a = po.fastq_parse(fq([(h, b"A", b"I")]), True)
        assert rc == 0 and h[: int(a["id_len"][0])] == rid and h[int(a["desc_start"][0]):] == desc


UNI = [("r1\u00a0desc one".encode(), b"r1", b"desc one"), ("r2\u2028\u3000 x\u2009y".encode(), b"r2", "x\u2009y".encode()),
       ("r3\u0085".encode(), b"r3", b""), ("caf\u00e9 ch=1".encode(), "caf\u00e9".encode(), b"ch=1"), ("r5\u1680\u205f\u202fz".encode(), b"r5", b"z"),
       (b"r6\xc2", b"r6\xc2", b""), ("r7\u200bx".encode(), "r7\u200bx".encode(), b"")]  # U+200B is not White_Space


def test_split_fastq_header_unicode_whitespace():
    """char::is_whitespace is Unicode White_Space (io.rs:6-17): the id also ends at U+0085, U+00A0, U+1680, U+2000-200A,
    U+2028/9, U+202F, U+205F, U+3000 (UTF-8 encoded); other multi-byte characters belong to the id"""
    from barbell_amd.annotate import split_fastq_header

    for h, rid, desc in UNI:
        rc, info, a = po.fastq_parse(fq([(h, b"A", b"I")]), True)
        assert rc == 0 and h[: int(a["id_len"][0])] == rid and h[int(a["desc_start"][0]):] == desc, h
        if h != b"r6\xc2":
            assert tuple(x.encode() for x in split_fastq_header(h.decode())) == (rid, desc)


@pytest.mark.gpu
def test_gpu_header_split_unicode_whitespace():
    from barbell_amd import annotate as A
    from barbell_amd import fastq as Q

    dm = A.Demuxer()
    for g in __import__("tests.common", fromlist=["x"]).config_groups("nbd96"):
        dm.add_query_group(g)
    text = fq([(h, b"ACGT", b"IIII") for h, _, _ in UNI])
    info, batch = Q.ingest(dm, text, True)
    a = Q.fetch(dm, info)
    rc, _, want = po.fastq_parse(text, True)
    assert rc == 0
    for k in ("id_len", "desc_start", "hdr_offsets"):
        assert a[k].tolist() == want[k].tolist(), k
    with pytest.raises(A.BarbellError):
        Q.read_ids(a)   # "r6\xc2" is not valid UTF-8: reported, not a UnicodeDecodeError


def test_malformed_records():
    good = fq(REC[:2])
    for bad, which in ((good + b"r3\nAC\n+\nII\n", 2), (good + b"@r3\nAC\n-\nII\n", 2), (good + b"@r3\nACG\n+\nII\n", 2),
                       (b"\n" + good + b"A\nC\nG\n", 0), (good + b"@r3\nAC\n", 2), (good + b"@r3\nAC\n+\n", 2),
                       (b"\n" + good, 2)):   # a truncated stream is reported first, as the record it breaks off in
        rc, info, _ = po.fastq_parse(bad, True)
        assert rc == _abi.BB_E_FASTQ and info["bad_record"] == which, bad
    # a quality line may start with '@' or '+'
    rc, info, arr = po.fastq_parse(b"@a\nAC\n+a\n@+\n@b\nG\n+\n+\n", True)
    assert rc == 0 and info["n_records"] == 2 and arr["quals"].tobytes() == b"@++"


# ---- GPU ---------------------------------------------------------------------------------------------
def gpu_parse(dm, text, final=True):
    from barbell_amd import fastq as Q
    from barbell_amd.annotate import BarbellError

    try:
        info, batch = Q.ingest(dm, text, final)
    except BarbellError as e:
        assert e.code == _abi.BB_E_FASTQ
        return e.code, None, None
    return 0, info, Q.fetch(dm, info, bases=True, quals=True)


def same(info, arr, oinfo, oarr):
    for k in ("n_records", "consumed", "n_bases", "n_hdr", "bad_record"):
        assert getattr(info, k) == oinfo[k], k
    for k in oarr:
        assert arr[k].tobytes() == oarr[k].tobytes(), k


@pytest.mark.gpu
def test_gpu_ingest_matches_oracle_on_handmade_blocks():
    from barbell_amd import annotate as A
    from tests.common import config_groups

    dm = A.Demuxer()
    for g in config_groups("rbk24"):
        dm.add_query_group(g)
    text = fq(REC)
    cases = [(fq(REC, nl, ln), True) for nl in (b"\n", b"\r\n") for ln in (True, False)]
    cases += [(fq(REC) + b"\n\r\n\n", True), (b"", True), (b"@a\nAC\n+a\n@+\n@b\nG\n+\n+\n", True)]
    cases += [(text[:cut], False) for cut in (1, 5, 40, len(text) - 1, len(text))]
    for t, final in cases:
        rc, info, arr = gpu_parse(dm, t, final)
        orc, oinfo, oarr = po.fastq_parse(t, final)
        assert rc == orc == 0
        same(info, arr, oinfo, oarr)
    good = fq(REC[:2])
    for bad in (good + b"r3\nAC\n+\nII\n", good + b"@r3\nAC\n-\nII\n", good + b"@r3\nACG\n+\nII\n", b"\n" + good, good + b"@r3\nAC\n"):
        assert gpu_parse(dm, bad, True)[0] == _abi.BB_E_FASTQ


@pytest.mark.gpu
def test_gpu_ingest_blocks_of_synthetic_reads_and_pipeline():
    """FASTQ text of 6000 synthetic reads cut at arbitrary byte positions: the concatenation of the ingested
    blocks equals the oracle's parse of the whole text; rows annotated from the HBM-resident batch equal the
    rows from the host-packed reads."""
    from barbell_amd import annotate as A, fastq as Q
    from tests.common import config_groups

    groups = config_groups("nbd96")
    n = 6000
    bases, offsets = A.synth_reads_host(groups, 77, 150, 5000, 0, n)
    rng = np.random.default_rng(3)
    quals = rng.integers(33, 90, size=len(bases), dtype=np.uint8)
    recs = []
    for i in range(n):
        a, b = int(offsets[i]), int(offsets[i + 1])
        recs.append(((b"read%d" % i) + (b" ch=%d  x" % (i % 9) if i % 3 else b""), bases[a:b].tobytes(), quals[a:b].tobytes()))
    text = fq(recs)
    orc, oinfo, oarr = po.fastq_parse(text, True)
    assert orc == 0 and oinfo["n_records"] == n and oarr["bases"].tobytes() == bases.tobytes()
    dm = A.Demuxer()
    for g in groups:
        dm.add_query_group(g)
    rc, info, arr = gpu_parse(dm, text, True)
    same(info, arr, oinfo, oarr)
    # arbitrary block boundaries with carry-over
    cuts = sorted(set(int(x) for x in rng.integers(1, len(text), size=7))) + [len(text)]
    got = {k: [] for k in ("bases", "quals", "hdr")}
    ids, carry, prev, total = [], b"", 0, 0
    for c in cuts:
        blk = carry + text[prev:c]
        rc, info, arr = gpu_parse(dm, blk, c == len(text))
        assert rc == 0
        carry, prev = blk[int(info.consumed):], c
        total += int(info.n_records)
        for k in got:
            got[k].append(arr[k].tobytes())
        ids += Q.read_ids(arr)
    assert total == n and carry == b""
    for k in got:
        assert b"".join(got[k]) == oarr[k].tobytes()
    assert ids == [r[0].split()[0].decode() for r in recs]
    # device-resident pipeline: ingest -> annotate from the batch's device pointers
    info, batch = Q.ingest(dm, text, True)
    import torch

    d_rows = torch.empty(4 * n * 48, dtype=torch.uint8, device="cuda")
    nr = dm.demux_dev(batch.d_bases, batch.d_offsets, n, d_rows.data_ptr(), 4 * n)
    rows_dev = np.frombuffer(d_rows[: nr * 48].cpu().numpy().tobytes(), dtype=_abi.ROW_DTYPE)
    rows_host = dm.demux_packed(bases, offsets)
    assert rows_dev.tobytes() == rows_host.tobytes()


# ---- the compact two-line form (BB_FASTQ_TWO_LINE): header and sequence lines only ---------------------------------
def two_line(recs, nl=b"\n", last_nl=True):
    t = b"".join(b"@" + h + nl + s + nl for h, s, q in recs)
    return t if last_nl else t[: -len(nl)]


def test_two_line_form_parses_like_the_four_line_form():
    for nl in (b"\n", b"\r\n"):
        for ln in (True, False):
            rc4, i4, a4 = po.fastq_parse(fq(REC, nl, ln), 1)
            rc2, i2, a2 = po.fastq_parse(two_line(REC, nl, ln), 1 | 2)
            assert rc4 == rc2 == 0 and i2["n_records"] == i4["n_records"] == len(REC)
            for k in ("offsets", "bases", "hdr", "hdr_offsets", "id_len", "desc_start"):
                assert a2[k].tobytes() == a4[k].tobytes(), k
    t = two_line(REC)
    rc, info, _ = po.fastq_parse(t[:-3], 2)          # not final: the partial record is left to the caller
    assert rc == 0 and info["n_records"] == len(REC) - 1 and t[info["consumed"]:].startswith(b"@")
    assert po.fastq_parse(t + b"r9\nAC\n", 3)[0] == _abi.BB_E_FASTQ   # a header without '@'
    assert po.fastq_parse(t + b"@r9\n", 3)[0] == _abi.BB_E_FASTQ       # the stream ends inside a record


@pytest.mark.gpu
def test_gpu_two_line_form():
    from barbell_amd import annotate as A, fastq as Q
    from tests.common import config_groups

    groups = config_groups("nbd96")
    dm = A.Demuxer()
    for g in groups:
        dm.add_query_group(g)

    def parse2(text, flags):
        try:
            info, batch = Q.ingest(dm, text, flags)
        except A.BarbellError as e:
            return e.code, None, None, None
        return 0, info, Q.fetch(dm, info, bases=True), batch

    for t, flags in [(two_line(REC, nl, ln), 3) for nl in (b"\n", b"\r\n") for ln in (True, False)] + [(two_line(REC)[:-3], 2), (b"", 3), (two_line(REC) + b"\n", 3)]:
        rc, info, arr, batch = parse2(t, flags)
        orc, oinfo, oarr = po.fastq_parse(t, flags)
        assert rc == orc == 0 and not batch.d_quals
        for k in ("n_records", "consumed", "n_bases", "n_hdr", "bad_record"):
            assert getattr(info, k) == oinfo[k], k
        for k in ("offsets", "bases", "hdr", "hdr_offsets", "id_len", "desc_start"):
            assert arr[k].tobytes() == oarr[k].tobytes(), k
    assert parse2(two_line(REC) + b"r9\nAC\n", 3)[0] == _abi.BB_E_FASTQ
    # synthetic reads: rows annotated from the two-line batch == rows from the four-line batch
    n = 3000
    bases, offsets = A.synth_reads_host(groups, 5, 150, 5000, 0, n)
    recs = [((b"read%d ch=%d" % (i, i % 7)), bases[int(offsets[i]):int(offsets[i + 1])].tobytes(), b"I" * int(offsets[i + 1] - offsets[i])) for i in range(n)]
    import torch

    outs = []
    for text, flags in ((fq(recs), 1), (two_line(recs), 3)):
        info, batch = Q.ingest(dm, text, flags)
        d_rows = torch.empty(4 * n * 48, dtype=torch.uint8, device="cuda")
        nr = dm.demux_dev(batch.d_bases, batch.d_offsets, n, d_rows.data_ptr(), 4 * n)
        outs.append(d_rows[: nr * 48].cpu().numpy().tobytes())
    assert outs[0] == outs[1] and len(outs[0]) > 48 * n // 2


# ---- the packed two-line form (BB_FASTQ_PACKED: two 4-bit base sets per byte, what `barbell-amd annotate` uploads) ---------------------
def test_packed_line_format_cpu():
    from barbell_amd import fastq as Q

    assert Q.pack_sequence_line(b"") == b"E"
    assert Q.pack_sequence_line(b"A") == bytes([(1 << 4) | (15 ^ 0xA)]) + b"O"
    assert Q.pack_sequence_line(b"ACGT") == bytes([(1 << 4) | (2 ^ 0xA), (4 << 4) | (8 ^ 0xA)]) + b"E"
    assert Q.pack_sequence_line(b"acgu") == Q.pack_sequence_line(b"ACGT")                 # case and U: the same base sets
    assert Q.pack_sequence_line(b"A**A") is not None and Q.pack_sequence_line(b"**AA") is None and Q.pack_sequence_line(b"A*") is not None
    rng = np.random.default_rng(1)
    for _ in range(200):   # no packable line holds a newline; every base set survives
        seq = bytes(rng.choice(np.frombuffer(b"ACGTNacgtnRYKMSWBDHVU*-.x", dtype=np.uint8), int(rng.integers(0, 70))))
        p = Q.pack_sequence_line(seq)
        if p is None:
            continue
        assert b"\n" not in p and p[-1:] in (b"E", b"O")
        body = np.frombuffer(p[:-1], dtype=np.uint8)
        codes = np.stack([body >> 4, (body & 15) ^ 0xA], axis=1).reshape(-1)[: len(seq)]
        assert codes.tolist() == Q.base_codes(seq).tolist()
        assert all(po.lib().bbo_iupac_code(Q.CANON[k]) % 255 == k for k in range(16))    # the canonical characters carry their code (0xFF = invalid = 0)


@pytest.mark.gpu
def test_gpu_ingest_packed_form_equals_the_text_form():
    """The same records as 4-line text, as two-line text and as packed two-line text: offsets, headers, ids identical; the unpacked bases are
    the canonical character of each original character's base set; annotate rows identical — on reads with lower case, N, IUPAC codes, U and
    single non-IUPAC characters, of every length parity, with record starts at every alignment."""
    from barbell_amd import annotate as A, fastq as Q
    from tests.common import config_groups

    groups = config_groups("nbd96")
    n = 3000
    bases, offsets = A.synth_reads_host(groups, 321, 1, 3000, 0, n)
    rng = np.random.default_rng(8)
    b = bases.copy()
    for alphabet, rate in ((b"acgtn", 0.05), (b"NRYKMSWBDHVU", 0.01), (b"*-.1x", 0.002)):
        pos = np.nonzero(rng.random(len(b)) < rate)[0]
        b[pos] = rng.choice(np.frombuffer(alphabet, dtype=np.uint8), len(pos))
    recs = [((b"read%d" % i) + (b" ch=%d  x" % (i % 9) if i % 3 else b""), b[int(offsets[i]):int(offsets[i + 1])].tobytes()) for i in range(n)]
    recs = [(h, s) for h, s in recs if Q.pack_sequence_line(s) is not None] + [(b"empty", b""), (b"one", b"G"), (b"two", b"gN")]
    n = len(recs)
    assert n > 2900
    text4 = b"".join(b"@" + h + b"\n" + s + b"\n+\n" + b"I" * len(s) + b"\n" for h, s in recs)
    text2 = b"".join(b"@" + h + b"\n" + s + b"\n" for h, s in recs)
    textp = Q.pack_two_line(recs)
    assert len(textp) < 0.55 * len(text2)
    dm = A.Demuxer()
    for g in groups:
        dm.add_query_group(g)
    res = {}
    for name, text, flags in (("4", text4, Q.BB_FASTQ_FINAL), ("2", text2, Q.BB_FASTQ_FINAL | Q.BB_FASTQ_TWO_LINE),
                              ("p", textp, Q.BB_FASTQ_FINAL | Q.BB_FASTQ_TWO_LINE | Q.BB_FASTQ_PACKED)):
        info, batch = Q.ingest(dm, text, flags)
        arr = Q.fetch(dm, info, bases=True)
        nr = dm.demux_dev(batch.d_bases, batch.d_offsets, int(info.n_records), dm.buf("rows").ensure(6 * n * 48), 6 * n)
        rows = dm.buf("rows").download(np.zeros(nr, dtype=_abi.ROW_DTYPE))
        res[name] = (info, arr, rows)
        assert int(info.n_records) == n and int(info.consumed) == len(text)
    for k in ("offsets", "hdr", "hdr_offsets", "id_len", "desc_start"):
        assert res["p"][1][k].tobytes() == res["4"][1][k].tobytes() == res["2"][1][k].tobytes(), k
    want = np.frombuffer(Q.CANON, dtype=np.uint8)[Q.base_codes(res["4"][1]["bases"].tobytes())]
    assert res["p"][1]["bases"].tobytes() == want.tobytes()
    assert res["2"][1]["bases"].tobytes() == res["4"][1]["bases"].tobytes()
    assert res["p"][2].tobytes() == res["4"][2].tobytes() == res["2"][2].tobytes() and len(res["p"][2]) > n // 2
    # a malformed packed line (no terminator) is a FASTQ error, not garbage
    with pytest.raises(A.BarbellError):
        Q.ingest(dm, b"@r\n\x1f\x2e\n", Q.BB_FASTQ_FINAL | Q.BB_FASTQ_TWO_LINE | Q.BB_FASTQ_PACKED)
