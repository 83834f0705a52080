"""The C++ host's staging of FASTQ text for upload (host/bb_feed.cpp: reader threads + sequencer), without a GPU: `barbell-amd stage` writes the
blocks it would upload.  The packed form (BB_FASTQ_PACKED: two 4-bit IUPAC base sets per byte in the sequence lines) must equal the Python
packer's text of the same records whatever the chunk size, reader count, line ends and file layout — the pairs are aligned to line starts, so
chunk boundaries inside a sequence line (odd and even positions, the byte before a line end, inside "\\r\\n") must not show."""
import gzip
import os
import subprocess

import numpy as np
import pytest

from barbell_amd import fastq as Q

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(ROOT, "barbell_amd", "bin", "barbell-amd")


@pytest.fixture(scope="module", autouse=True)
def _built():
    if not os.path.exists(CLI):
        import __graft_entry__ as g

        g.build()


def stage(files, out, *args, env=None):
    r = subprocess.run([CLI, "stage", "-i"] + [str(f) for f in files] + ["-o", str(out)] + list(args), capture_output=True, text=True,
                       env=dict(os.environ, **(env or {})))
    assert r.returncode == 0, r.stderr
    if os.environ.get("STAGE_DEBUG"):
        print(r.args, r.stdout, r.stderr)
    form, blocks = int(r.stdout.split()[1]), int(r.stdout.split()[3])
    return form, blocks, open(out, "rb").read()


def records(rng, n, lmin, lmax, alphabet=b"ACGT", junk=0.0):
    recs = []
    for i in range(n):
        L = int(rng.integers(lmin, lmax + 1))
        s = rng.choice(np.frombuffer(alphabet, dtype=np.uint8), L)
        if junk:
            pos = np.nonzero(rng.random(L) < junk)[0]
            pos = pos[np.concatenate([[True], np.diff(pos) > 1])] if len(pos) else pos      # never two adjacent: those have no packed form
            s[pos] = rng.choice(np.frombuffer(b"*-.1", dtype=np.uint8), len(pos))
        recs.append(((b"r%d" % i) + (b" ch=%d" % (i % 7) if i % 2 else b""), s.tobytes()))
    return recs


def fastq(recs, nl=b"\n", final_nl=True):
    t = b"".join(b"@" + h + nl + s + nl + b"+" + nl + b"I" * len(s) + nl for h, s in recs)
    return t if final_nl else t[: -len(nl)]


@pytest.mark.parametrize("nl", [b"\n", b"\r\n"])
def test_packed_staging_equals_the_python_packer(tmp_path, nl):
    rng = np.random.default_rng(5)
    recs = records(rng, 400, 0, 300, b"ACGTNacgtRYKMSWBDHVU", junk=0.01) + records(rng, 3, 5000, 9000)
    want = Q.pack_two_line(recs, nl)
    assert want is not None
    want2 = b"".join(b"@" + h + nl + s + nl for h, s in recs)
    for final_nl in (True, False):
        fq = tmp_path / "a.fastq"
        fq.write_bytes(fastq(recs, nl, final_nl))
        for block, threads in ((1 << 20, 1), (4096, 3), (1000, 4), (257, 2), (64, 5), (17, 3)):
            form, blocks, got = stage([fq], tmp_path / "o.bin", "--block-bytes", str(block), "-t", str(threads))
            assert form == 1, (block, threads)
            # (without a final newline it is the QUALITY line that lacks it: the staged header and sequence lines are whole either way)
            assert got == want, (block, threads, final_nl, len(got), len(want))
            form, blocks, got = stage([fq], tmp_path / "o.bin", "--block-bytes", str(block), "-t", str(threads), "--no-pack")
            assert form == 2 and got == want2
    assert len(want) < 0.56 * len(want2)


def test_several_files_gzip_and_blank_tails(tmp_path):
    rng = np.random.default_rng(9)
    parts = [records(rng, 150, 1, 400), records(rng, 1, 0, 0) + records(rng, 80, 30, 90), records(rng, 60, 200, 900)]
    files = []
    for i, recs in enumerate(parts):
        text = fastq(recs) + (b"\n" if i == 0 else b"\n\n" if i == 2 else b"")      # blank lines after the last record of a file
        p = tmp_path / f"f{i}.fastq{'.gz' if i == 1 else ''}"
        with (gzip.open(p, "wb") if i == 1 else open(p, "wb")) as f:
            f.write(text)
        files.append(p)
    for block in (1 << 20, 3000, 100):
        form, blocks, got = stage(files, tmp_path / "o.bin", "--block-bytes", str(block), "-t", "3")
        assert form == 1
        want = b"".join(Q.pack_two_line(recs) + (b"\n" if i == 0 else b"\n\n" if i == 2 else b"") for i, recs in enumerate(parts))
        assert got == want, block


def test_input_without_a_packed_form_falls_back(tmp_path):
    """two adjacent non-IUPAC characters at an even position of a line would pack to the byte '\\n': the run is staged as two-line text"""
    rng = np.random.default_rng(2)
    recs = records(rng, 50, 10, 200)
    recs[17] = (recs[17][0], b"ACGT**ACGT")
    fq = tmp_path / "a.fastq"
    fq.write_bytes(fastq(recs))
    form, blocks, got = stage([fq], tmp_path / "o.bin", "--block-bytes", "512", "-t", "2")
    assert form == 2 and got == b"".join(b"@" + h + b"\n" + s + b"\n" for h, s in recs)
    recs[17] = (recs[17][0], b"ACG**TACGT")      # at an odd position the pair is split over two bytes: packable
    fq.write_bytes(fastq(recs))
    form, blocks, got = stage([fq], tmp_path / "o.bin", "--block-bytes", "512", "-t", "2")
    assert form == 1 and got == Q.pack_two_line(recs)
    # files read with pread (no mapping) cannot be packed: no look-back
    form, blocks, got = stage([fq], tmp_path / "o.bin", "--block-bytes", "512", env={"BARBELL_AMD_NO_MMAP": "1"})
    assert form == 2


@pytest.mark.parametrize("nl", [b"\n", b"\r\n"])
def test_byte_range_shards_partition_the_records(tmp_path, nl):
    """--shard R/W --shard-by bytes: shard R stages the records that START in the R-th of W equal byte ranges of the file (the boundary is
    found in the text: BlockFeeder::record_start) — every record in exactly one shard, in file order, in every upload form; more shards
    than records leaves some empty; a gzip stream has no entry points and is refused"""
    rng = np.random.default_rng(31)
    recs = records(rng, 90, 1, 700)
    # quality lines that start with '@' and '+' must not be taken for headers / separators
    text = b"".join(b"@" + h + nl + s + nl + b"+" + nl + (b"@+" * len(s))[: len(s)] + nl for h, s in recs) + nl * 3
    fq = tmp_path / "a.fastq"
    fq.write_bytes(text)
    for flags in ([], ["--no-pack"], ["--no-compact"]):
        form, _, whole = stage([fq], tmp_path / "o.bin", "--block-bytes", "4096", *flags)
        for W in (2, 3, 16, 200):
            parts = [stage([fq], tmp_path / "s.bin", "--block-bytes", "4096", "--shard", f"{R}/{W}", "--shard-by", "bytes", *flags)[2] for R in range(W)]
            assert b"".join(parts) == whole, (flags, W)
            if W == 200:
                assert sum(1 for x in parts if not x) > 100      # more shards than records: most stage nothing
            if W == 2:
                assert all(len(x) > len(whole) // 4 for x in parts)
        env = {"BARBELL_AMD_NO_MMAP": "1"}
        parts = [stage([fq], tmp_path / "s.bin", "--block-bytes", "4096", "--shard", f"{R}/3", "--shard-by", "bytes", *flags, env=env)[2] for R in range(3)]
        assert b"".join(parts) == stage([fq], tmp_path / "o.bin", "--block-bytes", "4096", *flags, env=env)[2]
    gz = tmp_path / "a.fastq.gz"
    with gzip.open(gz, "wb") as f:
        f.write(text)
    r = subprocess.run([CLI, "stage", "-i", str(gz), "-o", str(tmp_path / "s.bin"), "--shard", "0/2", "--shard-by", "bytes"], capture_output=True, text=True)
    assert r.returncode != 0 and "--shard-by bytes" in r.stderr


def test_gzip_input_streams_in_bounded_pieces(tmp_path):
    """A gzip file (or a pipe) is inflated in record-aligned PIECES that the feeder chunks like small files (ParallelInflater; round 5: it used
    to be inflated whole — a 50 GB fastq.gz meant 200 GB of text in memory).  The staged text equals the plain file's whatever the piece size
    (64 bytes: every record its own piece; records larger than a piece), in every upload form, and what is held at once is bounded by the
    pieces in flight, not by the file."""
    import re

    rng = np.random.default_rng(41)
    recs = records(rng, 3000, 1, 900) + [(b"big", bytes(rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), 300_000)))] + records(rng, 200, 50, 400)
    text = fastq(recs) + b"\n\n"
    plain, gz = tmp_path / "a.fastq", tmp_path / "a.fastq.gz"
    plain.write_bytes(text)
    with gzip.open(gz, "wb", compresslevel=1) as f:
        f.write(text)
    for flags in ([], ["--no-pack"], ["--no-compact"]):
        form, _, want = stage([plain], tmp_path / "p.bin", "--block-bytes", "65536", *flags)
        for piece in ("64", "5000", "100000", None):
            env = {"BARBELL_AMD_GZ_PIECE": piece} if piece else {}
            f2, _, got = stage([gz], tmp_path / "g.bin", "--block-bytes", "65536", *flags, env=env)
            assert got == want and f2 == form, (flags, piece)
    # the bound: 3 MB of text in pieces of 100 kB — never more than a few pieces held (three published per file + the one a reader is copying)
    r = subprocess.run([CLI, "stage", "-i", str(gz), "-o", str(tmp_path / "g.bin"), "--block-bytes", "65536", "-t", "3"], capture_output=True, text=True,
                       env=dict(os.environ, BARBELL_AMD_GZ_PIECE="100000", BARBELL_AMD_PROFILE="1"))
    m = re.search(r"inflated in (\d+) piece\(s\) of at most (\d+) bytes; at most (\d+) bytes", r.stderr)
    assert r.returncode == 0 and m, r.stderr
    assert int(m.group(1)) >= len(text) // 400_000 and int(m.group(3)) <= 4 * 100_000 + 700_000 < len(text)   # (one piece grew to hold the 300 kb record: 600 KB of text)


def test_gzip_members_inflated_side_by_side(tmp_path):
    """`cat run/*.fastq.gz > all.fastq.gz` makes one gzip file of many members; zlib inflates 0.3 GB/s on one core.  The members are inflated
    on several (ParallelInflater::inflate_regular): the compressed bytes cut into ranges, a range's first member found by the gzip magic and
    its chain accepted only if it starts exactly where the text accepted so far ended — so a false start (stored members whose TEXT holds the
    magic bytes) costs time, never bytes.  Same staged text as from the plain file: members cut at arbitrary points of the text, ranges of
    200 bytes to 30 KB; a single member falls back to the serial path; trailing garbage is ignored and a truncated file is an error, as with gzread."""
    import re
    import zlib

    def member(t, level):
        c = zlib.compressobj(level, zlib.DEFLATED, 31)
        return c.compress(t) + c.flush()

    rng = np.random.default_rng(43)
    recs = records(rng, 700, 1, 700)
    recs = [(h + (b" \x1f\x8b\x08\x00\x00\x00\x00\x00\x00\x03" if i % 5 == 0 else b""), s) for i, (h, s) in enumerate(recs)]   # the magic in header lines
    text = fastq(recs)
    cuts = [0] + sorted(int(x) for x in rng.integers(0, len(text), 40)) + [len(text)]
    blob = b"".join(member(text[a:b], int(rng.choice([0, 1, 6]))) for a, b in zip(cuts[:-1], cuts[1:]))
    plain, gz = tmp_path / "a.fastq", tmp_path / "a.fastq.gz"
    plain.write_bytes(text)
    want = stage([plain], tmp_path / "p.bin", "--block-bytes", "65536", "--no-compact")[2]
    assert want == text

    def run(blob_, env):
        gz.write_bytes(blob_)
        r = subprocess.run([CLI, "stage", "-i", str(gz), "-o", str(tmp_path / "g.bin"), "--block-bytes", "65536", "--no-compact", "-t", "6"], capture_output=True,
                           text=True, env=dict(os.environ, BARBELL_AMD_PROFILE="1", **env), timeout=120)
        m = re.search(r"(\d+) range\(s\) of members", r.stderr)
        return r.returncode, (tmp_path / "g.bin").read_bytes() if r.returncode == 0 else b"", int(m.group(1)) if m else -1, r.stderr

    for rb, least in (("200", 1), ("8000", 8), ("30000", 5)):   # (with the range size fixed, members beyond 4 ranges of compressed bytes go to the serial path: 200-byte ranges give up early)
        rc, got, n_ranges, err = run(blob, {"BARBELL_AMD_GZ_RANGE": rb, "BARBELL_AMD_GZ_PIECE": "5000"})
        assert rc == 0 and got == text and n_ranges >= least, (rb, n_ranges, err[-300:])
    rc, got, n_ranges, _ = run(blob, {"BARBELL_AMD_GZ_SERIAL": "1"})
    assert rc == 0 and got == text and n_ranges == 0
    rc, got, n_ranges, _ = run(member(text, 6), {"BARBELL_AMD_GZ_RANGE": "200"})            # one member: nothing to do side by side
    assert rc == 0 and got == text and n_ranges <= 1
    # libdeflate (where the system has it) inflates members that fit its buffers; one that does not is zlib's, with the rest of its file
    for env in ({"BARBELL_AMD_LIBDEFLATE_MAX": "20000", "BARBELL_AMD_GZ_SERIAL": "1"}, {"BARBELL_AMD_NO_LIBDEFLATE": "1"}):
        rc, got, _, err = run(blob, env)
        assert rc == 0 and got == text, err[-300:]
    rc, got, n_ranges, _ = run(blob + b"\x00" * 700, {"BARBELL_AMD_GZ_RANGE": "200"})       # trailing bytes that are no member: ignored
    assert rc == 0 and got == text
    for env in ({"BARBELL_AMD_GZ_RANGE": "200"}, {"BARBELL_AMD_GZ_SERIAL": "1"}):           # cut inside the last member: an error, not a short file
        rc, got, _, err = run(blob[:-40], env)
        assert rc != 0 and "Error reading FASTQ file" in err, err[-300:]


def test_randomised_staging_slice():
    """a slice of tools/stage_fuzz.py (the full tool runs thousands of seeds): random layouts, line ends, blank tails, tiny chunks"""
    import sys

    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "stage_fuzz.py"), "100", "120"], capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0 and r.stdout.strip().endswith("120 seeds 0 bad"), r.stdout[-1500:] + r.stderr[-500:]
