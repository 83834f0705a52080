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
    rows = po.Oracle([g.as_tuple() for g in groups]).annotate(bases, offsets, n_threads=os.cpu_count() or 1)
    want = (A.TSV_HEADER + "\n" + "\n".join(A.format_rows(rows, ids, groups)) + "\n").encode()
    assert cli == want
    assert cli.split(b"\n")[0].decode() == A.TSV_HEADER and cli.count(b"\n") == len(rows) + 1


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
