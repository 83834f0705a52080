"""C++ host side (barbell_amd/csrc/host): the `barbell-amd annotate` driver mirrors the reference CLI's
annotate subcommand (bin/main.rs:64-112).  CPU: kit listing / argument errors / loud failure without
a GPU.  GPU: FASTQ (plain and gzip) -> annotation.tsv byte-identical to the Python host mirror and to
the oracle's rows formatted with the same schema."""
import gzip
import os
import subprocess

import numpy as np
import pytest

from barbell_amd import kits
from barbell_amd.parallel import effective_cpus  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(ROOT, "barbell_amd", "bin", "barbell-amd")


@pytest.fixture(scope="module", autouse=True)
def _built():
    if not os.path.exists(CLI):
        import __graft_entry__ as g

        g.build()


def write_fastq(path, ids, bases, offsets, gz=False):
    op = gzip.open if gz else open
    with op(path, "wb") as f:
        for i, rid in enumerate(ids):
            s = bytes(bases[int(offsets[i]):int(offsets[i + 1])])
            f.write(b"@" + rid.encode() + b" runid=xyz ch=1\n" + s + b"\n+\n" + b"I" * len(s) + b"\n")


def test_kits_listing():
    out = subprocess.run([CLI, "kits"], capture_output=True, text=True, check=True).stdout.split()
    assert out == kits.supported_kits() and "SQK-NBD114-96" in out and len(out) == 39


def test_argument_errors(tmp_path):
    assert subprocess.run([CLI, "annotate", "--kit", "SQK-NBD114-96"], capture_output=True).returncode == 2   # no input
    fq = tmp_path / "r.fastq"
    fq.write_bytes(b"@r1\nACGT\n+\nIIII\n")
    r = subprocess.run([CLI, "annotate", "-i", str(fq), "-o", str(tmp_path / "o.tsv")], capture_output=True)  # neither kit nor queries
    assert r.returncode == 2
    r = subprocess.run([CLI, "annotate", "-i", str(fq), "-o", str(tmp_path / "o.tsv"), "--kit", "NOPE"], capture_output=True, text=True)
    assert r.returncode == 1 and "Unknown or unsupported kit" in r.stderr
    # byte counts take binary suffixes (the documented default is "256Mi")
    base = [CLI, "stage", "-i", str(fq), "-o", str(tmp_path / "s.bin")]
    assert subprocess.run(base + ["--block-bytes", "1Ki"], capture_output=True).returncode == 0
    assert subprocess.run(base + ["--block-bytes", "256MiB"], capture_output=True).returncode == 0
    r = subprocess.run(base + ["--block-bytes", "12x"], capture_output=True, text=True)
    assert r.returncode == 2 and "byte count" in r.stderr
    r = subprocess.run(base + ["--shard", "0/2", "--shard-by", "lines"], capture_output=True, text=True)
    assert r.returncode == 2 and "files or bytes" in r.stderr


def test_shard_argument(tmp_path):
    fq = tmp_path / "r.fastq"
    fq.write_bytes(b"@r1\nACGT\n+\nIIII\n")
    base = [CLI, "annotate", "-i", str(fq), "-o", str(tmp_path / "o.tsv"), "--kit", "SQK-NBD114-96"]
    assert subprocess.run(base + ["--shard", "2/2"], capture_output=True).returncode == 2
    assert subprocess.run(base + ["--shard", "x"], capture_output=True).returncode == 2
    r = subprocess.run(base + ["--shard", "1/2"], capture_output=True, text=True)   # one file, shard 1 of 2: nothing to do
    assert r.returncode == 0 and "Nothing to do" in r.stderr


def _dump(p):
    from barbell_amd import _abi

    names = {_abi.BB_FTAG: "Ftag", _abi.BB_RTAG: "Rtag", _abi.BB_FFLANK: "Fflank", _abi.BB_RFLANK: "Rflank"}
    return [f"{names[e.match_type]}|{e.orientation}|{'*' if e.label is None else e.label}|{e.placeholder}|{e.relative_to}|"
            f"{e.range[0]}|{e.range[1]}|{','.join(str(c) for c in e.cuts)}" for e in p.elements] + ["--"]


def test_cpp_pattern_parser_matches_python():
    """C++ pattern_from_str == Python pattern_from_str on the kit pattern sets, the pattern.rs test strings
    and malformed input (pattern.rs:242-383)."""
    from barbell_amd import filter as F
    from barbell_amd.kits import _data

    strs = [p for ps in _data()["pattern_sets"].values() for p in ps] + [
        "Ftag[fw, *, @left(0..250)]", "Ftag[fw, ?1, @left(0..250)]__Ftag[<<, rc, ?1, @right(0..250)]",
        "Fflank[fw, *, @left(0..250)]__Ftag[fw, ~NB, @prev_left(5 .. 250), >>2, <<7]", "Rtag[rc, \"BC01\", @right(-10..+30)]",
        "Ftag[]", "Rflank[*]", "Ftag[fw, ?x, @middle(0..3), >>-1, <x, @left(1..), ?-1]", "Ftag[>>, <<, >>1]"]
    out = subprocess.run([CLI, "pattern"] + strs, capture_output=True, text=True, check=True).stdout.splitlines()
    assert out == [l for s in strs for l in _dump(F.pattern_from_str(s))]
    for bad in ["Ftag[fw]__Nope[rc]", "Ftag", "Ftag[fw]__", "Flank[fw]", "Ftag[fw, >]"]:
        with pytest.raises(ValueError):
            F.pattern_from_str(bad)
        r = subprocess.run([CLI, "pattern", bad], capture_output=True, text=True)
        assert r.returncode == 1 and "error:" in r.stderr, bad


def test_fails_loudly_without_gpu(tmp_path):
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    fq = tmp_path / "r.fastq"
    fq.write_bytes(b"@r1\nACGT\n+\nIIII\n")
    r = subprocess.run([CLI, "annotate", "-i", str(fq), "-o", str(tmp_path / "o.tsv"), "--kit", "SQK-NBD114-96"],
                       capture_output=True, text=True)
    assert r.returncode == 1 and "no usable HIP device" in r.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("gz", [False, True])
def test_cli_tsv_matches_python_and_oracle(tmp_path, gz):
    from barbell_amd import annotate as A
    from oracle import pyoracle as po

    groups = kits.groups_from_kit("SQK-NBD114-96", flank_max_errors=3)
    bases, offsets = A.synth_reads_host(groups, 2024, 300, 2500, 0, 900)
    ids = [f"read-{i:05d}" for i in range(900)]
    fq = tmp_path / ("reads.fastq.gz" if gz else "reads.fastq")
    write_fastq(fq, ids, bases, offsets, gz)
    out_cli, out_py = tmp_path / "cli.tsv", tmp_path / "py.tsv"
    env = dict(os.environ, BARBELL_AMD_NO_TORCH="1")
    r = subprocess.run([CLI, "annotate", "-i", str(fq), "-o", str(out_cli), "--kit", "SQK-NBD114-96", "--flank-max-errors", "3",
                        "--batch-reads", "250"], capture_output=True, text=True, env=env)
    assert r.returncode == 0, r.stderr
    A.annotate_with_kit([str(fq)], str(out_py), "SQK-NBD114-96", max_flank_errors=3, batch_reads=400)
    cli = out_cli.read_bytes()
    assert cli == out_py.read_bytes()
    # block reader edge: headroom smaller than the carried-over partial record (slow path), tiny blocks
    out2 = tmp_path / "cli2.tsv"
    r = subprocess.run([CLI, "annotate", "-i", str(fq), "-o", str(out2), "--kit", "SQK-NBD114-96", "--flank-max-errors", "3",
                        "--block-bytes", "5000"], capture_output=True, text=True, env=dict(env, BARBELL_AMD_HEAD_BYTES="64"))
    assert r.returncode == 0, r.stderr
    assert out2.read_bytes() == cli
    # blocks shorter than a record (reads of up to 2.5 kb, 5 kb of text each, in 1500-byte blocks): the record is carried over
    # several chunks instead of aborting the run
    out3 = tmp_path / "cli3.tsv"
    r = subprocess.run([CLI, "annotate", "-i", str(fq), "-o", str(out3), "--kit", "SQK-NBD114-96", "--flank-max-errors", "3",
                        "--block-bytes", "1500"], capture_output=True, text=True, env=env)
    assert r.returncode == 0, r.stderr
    assert out3.read_bytes() == cli
    # the quality lines are dropped on the host by default (annotate never reads them); --no-compact uploads them: same bytes out
    out6 = tmp_path / "cli6.tsv"
    r = subprocess.run([CLI, "annotate", "-i", str(fq), "-o", str(out6), "--kit", "SQK-NBD114-96", "--flank-max-errors", "3", "--no-compact",
                        "--block-bytes", "70000"], capture_output=True, text=True, env=env)
    assert r.returncode == 0, r.stderr
    assert out6.read_bytes() == cli
    # a pipe instead of a file (no size, no offsets; plain or gzip): read sequentially, same output
    fifo = tmp_path / "reads.pipe"
    os.mkfifo(fifo)
    import threading

    def feed():
        with open(fifo, "wb") as w:
            w.write(fq.read_bytes())

    th = threading.Thread(target=feed)
    th.start()
    out4 = tmp_path / "cli4.tsv"
    r = subprocess.run([CLI, "annotate", "-i", str(fifo), "-o", str(out4), "--kit", "SQK-NBD114-96", "--flank-max-errors", "3"],
                       capture_output=True, text=True, env=env, timeout=300)
    th.join()
    assert r.returncode == 0, r.stderr
    assert out4.read_bytes() == cli
    rows = po.Oracle([g.as_tuple() for g in groups]).annotate(bases, offsets, n_threads=effective_cpus())
    want = (A.TSV_HEADER + "\n" + "\n".join(A.format_rows(rows, ids, groups)) + "\n").encode()
    assert cli == want
    assert cli.split(b"\n")[0].decode() == A.TSV_HEADER and cli.count(b"\n") == len(rows) + 1
    # --policy reaches every context: the checker under the same policy gives the same bytes
    pol = "lm=left,trace=MSID,lodhi=3:0.5:2211"
    out5 = tmp_path / "cli5.tsv"
    r = subprocess.run([CLI, "annotate", "-i", str(fq), "-o", str(out5), "--kit", "SQK-NBD114-96", "--flank-max-errors", "3", "--policy", pol],
                       capture_output=True, text=True, env=env)
    assert r.returncode == 0, r.stderr
    rows = po.Oracle([g.as_tuple() for g in groups], policy=pol).annotate(bases, offsets, n_threads=effective_cpus())
    assert out5.read_bytes() == (A.TSV_HEADER + "\n" + "\n".join(A.format_rows(rows, ids, groups)) + "\n").encode() != cli


@pytest.mark.gpu
def test_cli_megabase_reads(tmp_path):
    """FASTQ records far larger than a block (reads of 0.3 / 1 / 2.5 M nt among ordinary ones, 64 KiB blocks: a record is carried over dozens
    of chunks), in every upload form and through a gzip stream: the TSV equals the oracle's rows rendered"""
    from barbell_amd import annotate as A
    from oracle import pyoracle as po
    from tests.common import long_batch

    groups = kits.groups_from_kit("SQK-NBD114-96", flank_max_errors=3)
    bases, offsets = long_batch(groups, 5, n_short=200)
    ids = [f"r{i}" for i in range(len(offsets) - 1)]
    rows = po.Oracle([g.as_tuple() for g in groups]).annotate(bases, offsets, n_threads=effective_cpus())
    want = (A.TSV_HEADER + "\n" + "\n".join(A.format_rows(rows, ids, groups)) + "\n").encode()
    env = dict(os.environ, BARBELL_AMD_NO_TORCH="1")
    for gz in (False, True):
        fq = tmp_path / ("reads.fastq.gz" if gz else "reads.fastq")
        write_fastq(fq, ids, bases, offsets, gz)
        for name, extra in (("packed", ["--block-bytes", "64Ki"]), ("text", ["--no-pack", "--block-bytes", "64Ki"]), ("whole", ["--no-compact", "--block-bytes", "1Mi"]),
                            ("default", [])):
            out = tmp_path / f"{name}.tsv"
            r = subprocess.run([CLI, "annotate", "-i", str(fq), "-o", str(out), "--kit", "SQK-NBD114-96", "--flank-max-errors", "3"] + extra,
                               capture_output=True, text=True, env=env, timeout=600)
            assert r.returncode == 0, (name, gz, r.stderr)
            assert out.read_bytes() == want, (name, gz)


@pytest.mark.gpu
@pytest.mark.parametrize("crlf", [False, True])
def test_cli_packed_upload_on_odd_characters(tmp_path, crlf):
    """By default `annotate` uploads the sequence lines two bases per byte (BB_FASTQ_PACKED: the kernels only look at a character's IUPAC base
    set).  Reads with lower case, N, IUPAC codes, U and single non-IUPAC characters, plain and gzip, LF and CRLF, tiny and ordinary blocks:
    annotation.tsv is byte-identical to --no-pack (sequence lines as text), --no-compact (whole records) and the oracle's rows.  A read with
    two adjacent non-IUPAC characters has no packed form: the run falls back by itself, same bytes."""
    from barbell_amd import annotate as A
    from oracle import pyoracle as po

    groups = kits.groups_from_kit("SQK-NBD114-96", flank_max_errors=3)
    n = 1200
    bases, offsets = A.synth_reads_host(groups, 4711, 1, 2500, 0, n)
    rng = np.random.default_rng(3)
    b = bases.copy()
    for alphabet, rate in ((b"acgtn", 0.06), (b"NRYKMSWBDHVUu", 0.01), (b"*-.1x", 0.003)):
        pos = np.nonzero(rng.random(len(b)) < rate)[0]
        b[pos] = rng.choice(np.frombuffer(alphabet, dtype=np.uint8), len(pos))
    # no two adjacent non-IUPAC characters in the first input (the packed form holds it) ...
    junk = np.isin(b, np.frombuffer(b"*-.1x", dtype=np.uint8))
    b[1:][junk[1:] & junk[:-1]] = ord("A")
    ids = [f"r{i}" for i in range(n)]
    nl = b"\r\n" if crlf else b"\n"

    def write(path, bb, gz=False):
        op = gzip.open if gz else open
        with op(path, "wb") as f:
            for i, rid in enumerate(ids):
                s_ = bytes(bb[int(offsets[i]):int(offsets[i + 1])])
                f.write(b"@" + rid.encode() + b" ch=2" + nl + s_ + nl + b"+" + nl + b"I" * len(s_) + nl)

    env = dict(os.environ, BARBELL_AMD_NO_TORCH="1", BARBELL_AMD_PROFILE="1")
    rows = po.Oracle([g.as_tuple() for g in groups]).annotate(b, offsets, n_threads=effective_cpus())
    want = (A.TSV_HEADER + "\n" + "\n".join(A.format_rows(rows, ids, groups)) + "\n").encode()
    assert len(rows) > n // 2
    outs = {}
    for gz in (False, True):
        fq = tmp_path / ("r.fastq.gz" if gz else "r.fastq")
        write(fq, b, gz)
        for name, extra in (("packed", []), ("packed_small", ["--block-bytes", "3000"]), ("text", ["--no-pack"]), ("whole", ["--no-compact"])):
            out = tmp_path / f"{name}{int(gz)}.tsv"
            r = subprocess.run([CLI, "annotate", "-i", str(fq), "-o", str(out), "--kit", "SQK-NBD114-96", "--flank-max-errors", "3"] + extra,
                               capture_output=True, text=True, env=env)
            assert r.returncode == 0, r.stderr
            assert "not representable" not in r.stderr
            assert out.read_bytes() == want, (name, gz)
    # ... and two adjacent ones in the second: packed staging gives up, the run starts over with text lines, same rows as the oracle's
    b2 = b.copy()
    k = int(offsets[n // 2]) + 10
    b2[k:k + 2] = np.frombuffer(b"**", dtype=np.uint8)
    if (k - int(offsets[n // 2])) % 2:
        b2[k + 2] = ord("*")            # whatever the alignment, one pair of the line is (junk, junk)
    fq2 = tmp_path / "r2.fastq"
    write(fq2, b2)
    out = tmp_path / "fallback.tsv"
    r = subprocess.run([CLI, "annotate", "-i", str(fq2), "-o", str(out), "--kit", "SQK-NBD114-96", "--flank-max-errors", "3", "--block-bytes", "200000"],
                       capture_output=True, text=True, env=env)
    assert r.returncode == 0 and "not representable in the packed upload form" in r.stderr, r.stderr
    rows2 = po.Oracle([g.as_tuple() for g in groups]).annotate(b2, offsets, n_threads=effective_cpus())
    assert out.read_bytes() == (A.TSV_HEADER + "\n" + "\n".join(A.format_rows(rows2, ids, groups)) + "\n").encode()


@pytest.mark.gpu
def test_cli_custom_dual_end_queries(tmp_path):
    from barbell_amd import annotate as A
    from tests.common import EX, config_groups

    groups = config_groups("dual")
    bases, offsets = A.synth_reads_host(groups, 77, 500, 3000, 0, 400)
    ids = [f"r{i}" for i in range(400)]
    fq = tmp_path / "reads.fastq"
    write_fastq(fq, ids, bases, offsets)
    out_cli = tmp_path / "cli.tsv"
    r = subprocess.run([CLI, "annotate", "-i", str(fq), "-o", str(out_cli), "-q", os.path.join(EX, "native_left.fasta"),
                        os.path.join(EX, "native_right.fasta"), "-b", "Ftag", "Rtag", "--flank-max-errors", "5"],
                       capture_output=True, text=True, env=dict(os.environ, BARBELL_AMD_NO_TORCH="1"))
    assert r.returncode == 0, r.stderr
    dm = A.Demuxer()
    for g in groups:
        dm.add_query_group(g)
    rows = dm.demux_packed(bases, offsets)
    want = (A.TSV_HEADER + "\n" + "\n".join(A.format_rows(rows, ids, groups)) + "\n").encode()
    assert out_cli.read_bytes() == want and b"Rtag" in want


@pytest.mark.gpu
def test_cli_fused_filter_matches_python(tmp_path):
    """annotate --kit-filter / -f pattern files: filtered.tsv and dropped.tsv byte-identical to the Python
    host's fused annotate->filter (whose verdicts the oracle checks in test_filter.py)."""
    from barbell_amd import annotate as A
    from barbell_amd import filter as F

    kit = "SQK-RBK114-24"
    groups = kits.groups_from_kit(kit, flank_max_errors=3)
    bases, offsets = A.synth_reads_host(groups, 77, 300, 2500, 0, 800)
    ids = [f"r{i}" for i in range(800)]
    fq = tmp_path / "reads.fastq"
    write_fastq(fq, ids, bases, offsets)
    env = dict(os.environ, BARBELL_AMD_NO_TORCH="1")
    pf = tmp_path / "patterns.txt"
    pf.write_text("Ftag[fw, *, @left(0..250), >>]\n\n  Ftag[fw, *, @left(0..250), >>]__Ftag[<<, rc, *, @right(0..250)]  \n")
    for name, flags, pats in (("kit", ["--kit-filter", "--maximize"], F.kit_patterns(kit, True)),
                              ("file", ["-f", str(pf)], F.patterns_from_files([str(pf)]))):
        o = {k: tmp_path / f"{name}_{k}.tsv" for k in ("a", "k", "d", "pa", "pk", "pd")}
        r = subprocess.run([CLI, "annotate", "-i", str(fq), "-o", str(o["a"]), "--kit", kit, "--flank-max-errors", "3", "--batch-reads", "300",
                            "--filtered", str(o["k"]), "--dropped", str(o["d"])] + flags, capture_output=True, text=True, env=env)
        assert r.returncode == 0, r.stderr
        assert "Filter:" in r.stderr
        A.annotate([str(fq)], str(o["pa"]), kits.groups_from_kit(kit), max_flank_errors=3, batch_reads=500, filter_patterns=pats,
                   filtered_file=str(o["pk"]), dropped_file=str(o["pd"]))
        for c, p in (("a", "pa"), ("k", "pk"), ("d", "pd")):
            assert o[c].read_bytes() == o[p].read_bytes(), (name, c)
        assert o["k"].stat().st_size > 100 and o["d"].stat().st_size > 100


@pytest.mark.gpu
@pytest.mark.parametrize("gz,render", [(False, "host"), (True, "host"), (False, "gpu")])
def test_cli_kit_matches_python_kit_driver(tmp_path, gz, render):
    """`barbell-amd kit` (use_kit.rs:11-109): every file of the output folder byte-identical to the Python kit
    driver's (whose trim output test_trim.py checks against the oracle) — with the records cut out of the staged text by the
    writer threads after the GPU's plan (the default) and rendered in HBM and downloaded (--gpu-render)."""
    import gzip as gzmod

    from barbell_amd import annotate as A
    from barbell_amd.use_kit import demux_using_kit

    kit = "SQK-RBK114-24"
    groups = kits.groups_from_kit(kit)
    n = 1100
    bases, offsets = A.synth_reads_host(groups, 3, 250, 2600, 0, n)
    rng = np.random.default_rng(2)
    fq = tmp_path / "reads.fastq"
    with open(fq, "wb") as f:
        for i in range(n):
            s = bytes(bases[int(offsets[i]):int(offsets[i + 1])])
            q = bytes(rng.integers(33, 90, size=len(s), dtype=np.uint8))
            f.write(b"@read%d%s\n" % (i, b" ch=%d  st=xyz" % (i % 50) if i % 5 else b"") + s + b"\n+\n" + q + b"\n")
    env = dict(os.environ, BARBELL_AMD_NO_TORCH="1")
    oc, op = tmp_path / "cli", tmp_path / "py"
    flags = ["--maximize", "--failed-out", str(tmp_path / "failed_cli.txt"), "--batch-reads", "333"] + (["--gzip"] if gz else []) + \
        (["--gpu-render"] if render == "gpu" else [])
    r = subprocess.run([CLI, "kit", "-k", kit, "-i", str(fq), "-o", str(oc)] + flags, capture_output=True, text=True, env=env)
    assert r.returncode == 0, r.stderr
    assert "Top 10 most common patterns" in r.stdout and "Found " in r.stdout and "Done!" in r.stdout
    logs = []
    _, _, insp = demux_using_kit([str(fq)], kit, str(op), maximize=True, failed_out=str(tmp_path / "failed_py.txt"), gzip=gz,
                                 batch_reads=500, log=logs.append)
    names = sorted(x.name for x in op.iterdir())
    assert names == sorted(x.name for x in oc.iterdir()) and len(names) > 10
    for x in names:
        a, b = (oc / x).read_bytes(), (op / x).read_bytes()
        if x.endswith(".gz"):
            a, b = gzmod.decompress(a), gzmod.decompress(b)
        assert a == b, x
    assert (tmp_path / "failed_cli.txt").read_bytes() == (tmp_path / "failed_py.txt").read_bytes()
    if gz:   # the .gz files are written by libdeflate (one member per span) where the system has it: zlib's gzwrite holds the same records
        oz = tmp_path / "cli_zlib"
        r2 = subprocess.run([CLI, "kit", "-k", kit, "-i", str(fq), "-o", str(oz)] + flags, capture_output=True, text=True, env=dict(env, BARBELL_AMD_NO_LIBDEFLATE="1"))
        assert r2.returncode == 0, r2.stderr
        for x in names:
            if x.endswith(".gz"):
                assert gzmod.decompress((oz / x).read_bytes()) == gzmod.decompress((oc / x).read_bytes()), x
    # the summary lines printed by the CLI are the inspector's
    want = insp.summary(10)
    got = [l for l in r.stdout.splitlines() if l.startswith(("Found", "\tPattern", "\t\t", "Showed"))]
    counts = lambda ls: [l for l in ls if not l.startswith("\t\t")]
    assert counts(got) == counts(want) and sorted(got) == sorted(want)


@pytest.mark.gpu
def test_cli_annotate_trim_flags(tmp_path):
    """annotate --trim-output with the `barbell trim` label flags == the Python host with the same TrimConfig"""
    from barbell_amd import annotate as A
    from barbell_amd import filter as F
    from barbell_amd import trim as T

    kit = "SQK-NBD114-96"
    groups = kits.groups_from_kit(kit, flank_max_errors=3)
    bases, offsets = A.synth_reads_host(groups, 12, 300, 2000, 0, 700)
    ids = [f"r{i}" for i in range(700)]
    fq = tmp_path / "reads.fastq.gz"
    write_fastq(fq, ids, bases, offsets, gz=True)
    env = dict(os.environ, BARBELL_AMD_NO_TORCH="1")
    for name, flags, cfg in (("default", [], T.TrimConfig()),
                             ("sorted", ["--sort-labels", "--no-flanks", "--flip"], T.TrimConfig(True, True, False, True, None, flip=True)),
                             ("right", ["--only-side", "right", "--no-orientation", "--skip-trim"],
                              T.TrimConfig(True, False, True, False, "right", skip_trim=True)),
                             ("nolabel", ["--no-label"], T.TrimConfig(False))):
        oc, op, og = tmp_path / f"{name}_cli", tmp_path / f"{name}_py", tmp_path / f"{name}_cli_gpu"
        for o, extra in ((oc, []), (og, ["--gpu-render"])):
            r = subprocess.run([CLI, "annotate", "-i", str(fq), "-o", str(tmp_path / "a.tsv"), "--kit", kit, "--flank-max-errors", "3",
                                "--kit-filter", "--maximize", "--trim-output", str(o), "--batch-reads", "256"] + flags + extra,
                               capture_output=True, text=True, env=env)
            assert r.returncode == 0, r.stderr
        assert sorted(x.name for x in og.iterdir()) == sorted(x.name for x in oc.iterdir())
        for x in oc.iterdir():
            assert x.read_bytes() == (og / x.name).read_bytes(), (name, x.name)
        A.annotate([str(fq)], str(tmp_path / "pa.tsv"), kits.groups_from_kit(kit), max_flank_errors=3, filter_patterns=F.kit_patterns(kit, True),
                   trim_folder=str(op), trim_config=cfg, batch_reads=300)
        names = sorted(x.name for x in op.iterdir())
        assert names == sorted(x.name for x in oc.iterdir()) and names, name
        for x in names:
            assert (oc / x).read_bytes() == (op / x).read_bytes(), (name, x)
    r = subprocess.run([CLI, "annotate", "-i", str(fq), "--kit", kit, "--kit-filter", "--trim-output", str(tmp_path / "x"), "--sort-labels",
                        "--only-side", "left"], capture_output=True, text=True, env=env)
    assert r.returncode == 2


@pytest.mark.gpu
def test_cli_many_gzip_files_parallel_inflate(tmp_path):
    """several .fastq.gz inputs are inflated by a thread pool and consumed in input order: same annotation.tsv and
    trimmed files as one plain file holding the same records (incl. an empty file in the middle)"""
    from barbell_amd import annotate as A

    kit = "SQK-NBD114-96"
    groups = kits.groups_from_kit(kit, flank_max_errors=3)
    n = 1300
    bases, offsets = A.synth_reads_host(groups, 555, 300, 2200, 0, n)
    ids = [f"r{i}" for i in range(n)]
    whole = tmp_path / "all.fastq"
    write_fastq(whole, ids, bases, offsets)
    cuts = [0, 200, 200, 650, 651, 1000, n]
    parts = []
    for k in range(len(cuts) - 1):
        a, b = cuts[k], cuts[k + 1]
        path = tmp_path / f"part{k}.fastq.gz"
        sub_off = offsets[a:b + 1] - offsets[a]
        write_fastq(path, ids[a:b], bases[int(offsets[a]):int(offsets[b])], sub_off, gz=True)
        parts.append(str(path))
    env = dict(os.environ, BARBELL_AMD_NO_TORCH="1")
    outs = {}
    for name, inputs in (("one", [str(whole)]), ("many", parts)):
        o = tmp_path / name
        r = subprocess.run([CLI, "kit", "-k", kit, "-i"] + inputs + ["-o", str(o), "--flank-max-errors", "3", "--maximize", "-t", "4",
                            "--batch-reads", "100"], capture_output=True, text=True, env=env)
        assert r.returncode == 0, r.stderr
        outs[name] = {x.name: x.read_bytes() for x in o.iterdir()}
    assert outs["one"].keys() == outs["many"].keys() and len(outs["one"]) > 20
    for k in outs["one"]:
        assert outs["one"][k] == outs["many"][k], k


@pytest.mark.gpu
def test_one_stream_over_several_contexts(tmp_path):
    """SURVEY §8e / VERDICT r01 #6: one FASTQ stream, block i -> context i mod G, rows merged in block order, histogram
    summed over the contexts.  With one GPU in the box the contexts share device 0 (--devices 0,0,0): annotation.tsv,
    filtered.tsv, the per-barcode FASTQ files and the counts must be byte-identical to the single-context run, and equal
    to the rows' own label counts; BARBELL_AMD_FORCE_RCCL=1 sends a single context's histogram through ncclAllReduce."""
    from barbell_amd import annotate as A

    groups = kits.groups_from_kit("SQK-NBD114-96", flank_max_errors=3)
    n = 3000
    bases, offsets = A.synth_reads_host(groups, 99, 500, 3000, 0, n)
    ids = [f"r{i}" for i in range(n)]
    fqs = []
    for part in range(2):  # two input files: the stream continues across them
        fq = tmp_path / f"reads{part}.fastq"
        lo, hi = part * n // 2, (part + 1) * n // 2
        write_fastq(fq, ids[lo:hi], bases[int(offsets[lo]):int(offsets[hi])], offsets[lo:hi + 1] - offsets[lo])
        fqs.append(str(fq))
    env = dict(os.environ, BARBELL_AMD_NO_TORCH="1")
    outs = {}
    for name, extra, e in (("one", ["--devices", "0"], env), ("three", ["--devices", "0,0,0"], env), ("default", [], env),
                           ("rccl1", ["--devices", "0"], dict(env, BARBELL_AMD_FORCE_RCCL="1"))):
        d = tmp_path / name
        d.mkdir()
        r = subprocess.run([CLI, "annotate", "-i"] + fqs + ["-o", str(d / "a.tsv"), "--kit", "SQK-NBD114-96", "--flank-max-errors", "3",
                            "--block-bytes", "300000", "--kit-filter", "--maximize", "--filtered", str(d / "f.tsv"), "--trim-output", str(d / "trim"),
                            "--counts", str(d / "counts.tsv")] + extra, capture_output=True, text=True, env=e)
        assert r.returncode == 0, r.stderr
        files = {"a": (d / "a.tsv").read_bytes(), "f": (d / "f.tsv").read_bytes(), "c": (d / "counts.tsv").read_bytes()}
        for f in sorted(os.listdir(d / "trim")):
            files["trim/" + f] = (d / "trim" / f).read_bytes()
        outs[name] = (files, r.stderr)
    assert "histogram summed by single" in outs["one"][1] and "summed by host" in outs["three"][1] and "summed by rccl" in outs["rccl1"][1]
    assert "librccl.so bound" in outs["rccl1"][1] and "librccl.so bound" not in outs["one"][1]      # the library was really mapped, and only then
    for name in ("three", "default", "rccl1"):
        assert outs[name][0].keys() == outs["one"][0].keys()
        for k in outs["one"][0]:
            assert outs[name][0][k] == outs["one"][0][k], (name, k)
    # counts.tsv = rows per label of annotation.tsv
    rows = [l.split("\t") for l in outs["one"][0]["a"].decode().splitlines()[1:]]
    from collections import Counter
    want = Counter(r[12] for r in rows)
    got = {l.split("\t")[1]: int(l.split("\t")[2]) for l in outs["one"][0]["c"].decode().splitlines()}
    assert sum(got.values()) == len(rows) and all(got[k] == v for k, v in want.items())
    assert len(outs["one"][0]) > 20  # many per-barcode files


@pytest.mark.gpu
def test_shards_one_process_per_gpu_reduce_their_histograms(tmp_path):
    """`--shard R/W --rccl-id PATH` (one process per GPU; annotator.rs:278-280 is the fan-out it replaces; north_star: "RCCL ...
    for the final per-barcode count reduction"): the W processes all-reduce their histograms and shard 0 writes ONE counts file equal to
    the single-process run's.  On the one-GPU box the two processes share device 0, where one communicator cannot hold both ranks: they are
    summed through the rendezvous files; `--shard 0/1 --rccl-id` is the same code with W = 1 and goes through ncclGetUniqueId /
    ncclCommInitRank / ncclAllReduce of librccl.so itself (checked in the process's own map, BARBELL_AMD_PROFILE)."""
    import subprocess as sp

    from barbell_amd import annotate as A

    groups = kits.groups_from_kit("SQK-NBD114-96", flank_max_errors=3)
    n = 4000
    bases, offsets = A.synth_reads_host(groups, 123, 500, 3000, 0, n)
    ids = [f"r{i}" for i in range(n)]
    fqs = []
    for part in range(2):
        fq = tmp_path / f"reads{part}.fastq"
        lo, hi = part * n // 2, (part + 1) * n // 2
        write_fastq(fq, ids[lo:hi], bases[int(offsets[lo]):int(offsets[hi])], offsets[lo:hi + 1] - offsets[lo])
        fqs.append(str(fq))
    env = dict(os.environ, BARBELL_AMD_NO_TORCH="1", BARBELL_AMD_RCCL_TIMEOUT="120", HSA_ENABLE_IPC_MODE_LEGACY="0")
    common = ["--kit", "SQK-NBD114-96", "--flank-max-errors", "3", "--device", "0"]

    def run(name, extra):
        d = tmp_path / name
        d.mkdir(exist_ok=True)
        return sp.Popen([CLI, "annotate", "-i"] + fqs + ["-o", str(d / "a.tsv"), "--counts", str(d / "counts.tsv")] + common + extra,
                        stdout=sp.PIPE, stderr=sp.PIPE, text=True, env=env), d

    p, d_one = run("one", [])
    _, err = p.communicate(timeout=600)
    assert p.returncode == 0, err
    want = (d_one / "counts.tsv").read_bytes()
    assert sum(int(l.split(b"\t")[2]) for l in want.splitlines()) > n // 2
    # two processes, one per shard, both on GPU 0, started together
    rid = str(tmp_path / "rendezvous")
    procs = [run(f"shard{r}", ["--shard", f"{r}/2", "--rccl-id", rid]) for r in range(2)]
    errs = []
    for p, _ in procs:
        _, err = p.communicate(timeout=600)
        assert p.returncode == 0, err
        errs.append(err)
    assert (procs[0][1] / "counts.tsv").read_bytes() == want              # ONE counts file, the run's
    assert not (procs[1][1] / "counts.tsv").exists()
    assert all("2 processes share a device" in e for e in errs), errs
    assert (procs[0][1] / "a.tsv").read_bytes() + b"".join((procs[1][1] / "a.tsv").read_bytes().splitlines(keepends=True)[1:]) == (d_one / "a.tsv").read_bytes()
    assert not [f for f in os.listdir(tmp_path) if f.startswith("rendezvous")]    # shard 0 cleaned the rendezvous files up
    # without --rccl-id each process keeps its own counts and says so
    p, d = run("alone", ["--shard", "0/2"])
    _, err = p.communicate(timeout=600)
    assert p.returncode == 0 and "without --rccl-id" in err and (d / "counts.tsv").read_bytes() != want
    # W = 1: the RCCL bootstrap and collective themselves
    p, d = run("w1", ["--shard", "0/1", "--rccl-id", str(tmp_path / "rv1")])
    _, err = p.communicate(timeout=600)
    assert p.returncode == 0, err
    assert "rccl (1 processes, ncclCommInitRank)" in err and (d / "counts.tsv").read_bytes() == want
    # a shard whose peers never come gives up with a message instead of hanging
    env["BARBELL_AMD_RCCL_TIMEOUT"] = "3"
    p, d = run("lonely", ["--shard", "0/2", "--rccl-id", str(tmp_path / "rv2")])
    _, err = p.communicate(timeout=600)
    assert p.returncode == 1 and "timed out after 3 s waiting for" in err and "rv2.r1.hello" in err, err   # (names the peer that never said hello)


@pytest.mark.gpu
@pytest.mark.parametrize("pack", [True, False])
def test_byte_range_shards_of_one_file(tmp_path, pack):
    """`--shard R/W --shard-by bytes`: ONE plain FASTQ over W processes (one per GPU in production, all on GPU 0 here) — shard R takes the
    records that start in the R-th of W byte ranges; the shards' TSVs, one after the other, are the single-process TSV and the reduced
    counts file is the single-process one.  Five blank lines after the last record (CRLF) ride along: every upload form ignores them."""
    import subprocess as sp

    from barbell_amd import annotate as A

    groups = kits.groups_from_kit("SQK-NBD114-96", flank_max_errors=3)
    n = 3001
    bases, offsets = A.synth_reads_host(groups, 77, 200, 4000, 0, n)
    fq = tmp_path / "reads.fastq"
    write_fastq(fq, [f"r{i} ch={i % 7}" for i in range(n)], bases, offsets)
    fq.write_bytes(fq.read_bytes().replace(b"\n", b"\r\n") + b"\r\n" * 5)
    env = dict(os.environ, BARBELL_AMD_NO_TORCH="1", BARBELL_AMD_RCCL_TIMEOUT="120", HSA_ENABLE_IPC_MODE_LEGACY="0")
    common = ["--kit", "SQK-NBD114-96", "--flank-max-errors", "3", "--device", "0", "--block-bytes", "1Mi"] + ([] if pack else ["--no-pack"])

    def run(name, extra):
        d = tmp_path / name
        d.mkdir(exist_ok=True)
        return sp.Popen([CLI, "annotate", "-i", str(fq), "-o", str(d / "a.tsv"), "--counts", str(d / "counts.tsv")] + common + extra,
                        stdout=sp.PIPE, stderr=sp.PIPE, text=True, env=env), d

    p, d_one = run("one", [])
    _, err = p.communicate(timeout=600)
    assert p.returncode == 0, err
    whole = (d_one / "a.tsv").read_bytes().splitlines(keepends=True)
    assert len({l.split(b"\t")[0] for l in whole[1:]}) > n // 2
    W = 3
    rid = str(tmp_path / "rendezvous")
    procs = [run(f"shard{r}", ["--shard", f"{r}/{W}", "--shard-by", "bytes", "--rccl-id", rid]) for r in range(W)]
    for p, _ in procs:
        _, err = p.communicate(timeout=600)
        assert p.returncode == 0, err
    parts = [(d / "a.tsv").read_bytes().splitlines(keepends=True) for _, d in procs]
    assert all(len(x) > len(whole) // 6 for x in parts)                      # every shard had its share
    assert parts[0] + parts[1][1:] + parts[2][1:] == whole
    assert (procs[0][1] / "counts.tsv").read_bytes() == (d_one / "counts.tsv").read_bytes()


@pytest.mark.gpu
def test_cli_drops_quality_lines_that_look_like_headers(tmp_path):
    """Two-line staging (the readers drop '+' and quality lines; the chunk's line phase is read off the text): quality lines that
    start with '@' or '+', CRLF line ends, many chunk boundaries — the TSV equals the one made with --no-compact and the Python host's."""
    from barbell_amd import annotate as A

    groups = kits.groups_from_kit("SQK-NBD114-96", flank_max_errors=3)
    n = 700
    bases, offsets = A.synth_reads_host(groups, 31, 120, 900, 0, n)
    rng = np.random.default_rng(5)
    fq = tmp_path / "tricky.fastq"
    with open(fq, "wb") as f:
        for i in range(n):
            s = bytes(bases[int(offsets[i]):int(offsets[i + 1])])
            q = bytearray(rng.integers(33, 75, size=len(s), dtype=np.uint8).tobytes())
            q[0] = b"@+@I"[i % 4]
            nl = b"\r\n" if i % 5 == 0 else b"\n"
            f.write(b"@t%d extra" % i + nl + s + nl + (b"+t%d" % i if i % 2 else b"+") + nl + bytes(q) + nl)
    env = dict(os.environ, BARBELL_AMD_NO_TORCH="1")
    outs = []
    for extra in (["--block-bytes", "4096"], ["--block-bytes", "4096", "--no-compact"], ["--block-bytes", "30011", "-t", "3"], []):
        o = tmp_path / ("o%d.tsv" % len(outs))
        r = subprocess.run([CLI, "annotate", "-i", str(fq), "-o", str(o), "--kit", "SQK-NBD114-96", "--flank-max-errors", "3"] + extra,
                           capture_output=True, text=True, env=env)
        assert r.returncode == 0, r.stderr
        outs.append(o.read_bytes())
    assert outs[0] == outs[1] == outs[2] == outs[3] and outs[0].count(b"\n") > n // 2
    # the block buffers page-locked (hipHostMalloc) instead of pageable, plain files read with pread instead of mapped: same bytes out
    for knob in ("BARBELL_AMD_PINNED_SLOTS", "BARBELL_AMD_NO_MMAP"):
        o = tmp_path / "knob.tsv"
        r = subprocess.run([CLI, "annotate", "-i", str(fq), "-o", str(o), "--kit", "SQK-NBD114-96", "--flank-max-errors", "3", "--block-bytes", "30011"],
                           capture_output=True, text=True, env=dict(env, **{knob: "1"}))
        assert r.returncode == 0, r.stderr
        assert o.read_bytes() == outs[0], knob
    # the host side alone: every staged byte counted, nothing annotated
    r = subprocess.run([CLI, "annotate", "-i", str(fq), "-o", str(tmp_path / "feed.tsv"), "--kit", "SQK-NBD114-96", "--block-bytes", "30011"],
                       capture_output=True, text=True, env=dict(env, BARBELL_AMD_FEED_ONLY="1"))
    assert r.returncode == 0 and "feed-only:" in r.stderr, r.stderr
    staged = int(r.stderr.split("feed-only: ")[1].split(" bytes")[0])
    assert 0 < staged < len(fq.read_bytes())   # headers and sequences only (the two-line form)
    py = tmp_path / "py.tsv"
    A.annotate_with_kit([str(fq)], str(py), "SQK-NBD114-96", max_flank_errors=3)
    assert py.read_bytes() == outs[0]
    # a file that is not 4-line FASTQ is refused in either mode
    bad = tmp_path / "bad.fastq"
    bad.write_bytes(fq.read_bytes()[:5000] + b"@x\nACGT\n-\nIIII\n" + fq.read_bytes()[5000:])
    for extra in ([], ["--no-compact"]):
        r = subprocess.run([CLI, "annotate", "-i", str(bad), "-o", str(tmp_path / "b.tsv"), "--kit", "SQK-NBD114-96"] + extra, capture_output=True, text=True, env=env)
        assert r.returncode == 1 and "FASTQ" in r.stderr
    # What the GPU parser checks in the 4-line form holds in the two-line form too, where the quality lines never leave the host: a quality
    # line as long as its sequence, a file that ends on a record boundary (interrupted copies are common).  Whatever the chunk size — the
    # pair of lines may lie in different chunks — and in either mode.
    text = fq.read_bytes()
    recs = text.split(b"\n@t")          # record i starts with "@t<i>" (the first keeps its '@')
    cases = {}
    for i in (0, 3, 350, n - 1):         # a quality line one character short: first record, a CRLF-free one, mid-file, the last
        r = recs[i]
        cr = r.endswith(b"\r") or r.endswith(b"\r\n")
        body = r.rstrip(b"\r\n")
        short = body[:-1] + (b"\r\n" if cr else b"\n") if i == n - 1 else body[:-1] + (b"\r" if cr else b"")
        cases["short_q_%d" % i] = b"\n@t".join(recs[:i] + [short] + recs[i + 1:])
    last = text.rfind(b"\n@t") + 1
    rec = text[last:]
    lines = rec.split(b"\n")
    cases["cut_mid_sequence"] = text[:last] + lines[0] + b"\n" + lines[1][: len(lines[1]) // 2]
    cases["cut_after_plus"] = text[:last] + lines[0] + b"\n" + lines[1] + b"\n" + lines[2] + b"\n"
    cases["cut_mid_quality"] = text[: len(text) - 40]
    cases["cut_after_header"] = text[:last] + lines[0] + b"\n"
    for name, data in cases.items():
        p = tmp_path / (name + ".fastq")
        p.write_bytes(data)
        for extra in ([], ["--no-compact"], ["--block-bytes", "4096"], ["--block-bytes", "1500", "-t", "3"]):
            r = subprocess.run([CLI, "annotate", "-i", str(p), "-o", str(tmp_path / "b.tsv"), "--kit", "SQK-NBD114-96"] + extra, capture_output=True, text=True, env=env)
            assert r.returncode == 1 and "FASTQ" in r.stderr, (name, extra, r.returncode, r.stderr[-300:])
    # still fine: no final newline, a blank line after the last record
    for name, data in (("no_final_newline", text.rstrip(b"\r\n")), ("one_blank", text + b"\n"), ("one_blank_crlf", text + b"\r\n")):
        p = tmp_path / (name + ".fastq")
        p.write_bytes(data)
        for extra in ([], ["--block-bytes", "1500"], ["--no-compact"]):
            o = tmp_path / "ok.tsv"
            r = subprocess.run([CLI, "annotate", "-i", str(p), "-o", str(o), "--kit", "SQK-NBD114-96", "--flank-max-errors", "3"] + extra, capture_output=True, text=True, env=env)
            assert r.returncode == 0, (name, extra, r.stderr[-300:])
            assert o.read_bytes() == outs[0], (name, extra)


@pytest.mark.gpu
def test_cli_kit_tiny_blocks_cut_from_assembled_blocks(tmp_path):
    """`barbell-amd kit` with 4 KB blocks and records of ~5 KB: every record spans blocks, so the text a block's records are cut out of
    is the sequencer's assembled copy, not a page-locked slot; the output folder is byte-identical to the default run's and to
    --gpu-render's."""
    kit = "SQK-RBK114-24"
    groups = kits.groups_from_kit(kit)
    from barbell_amd import annotate as A

    n = 300
    bases, offsets = A.synth_reads_host(groups, 11, 2300, 2600, 0, n)
    rng = np.random.default_rng(5)
    fq = tmp_path / "reads.fastq"
    with open(fq, "wb") as f:
        for i in range(n):
            s = bytes(bases[int(offsets[i]):int(offsets[i + 1])])
            q = bytes(rng.integers(33, 90, size=len(s), dtype=np.uint8))
            f.write(b"@r%d d=%d\n" % (i, i) + s + b"\n+\n" + q + b"\n")
    env = dict(os.environ, BARBELL_AMD_NO_TORCH="1")
    outs = {}
    for name, flags in (("default", []), ("tiny", ["--batch-reads", "1"]), ("tiny_gpu", ["--batch-reads", "1", "--gpu-render"])):
        o = tmp_path / name
        r = subprocess.run([CLI, "kit", "-k", kit, "-i", str(fq), "-o", str(o), "--maximize"] + flags, capture_output=True, text=True, env=env)
        assert r.returncode == 0, r.stderr
        outs[name] = {x.name: x.read_bytes() for x in o.iterdir()}
    assert len(outs["default"]) > 5
    assert outs["tiny"] == outs["default"] and outs["tiny_gpu"] == outs["default"]
