"""CPU-side checks of the product library: it loads without a GPU, exports every symbol the
headers declare, refuses to compute without a device (no CPU fallback), and the host synthetic
generator is deterministic and shard-consistent."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from barbell_amd import _abi, _lib, kits

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def L():
    if not os.path.exists(_lib.SO_PATH):
        import __graft_entry__ as g

        g.build()
    return _lib.lib()


def test_exports_match_headers(L):
    declared = set()
    for h in ("barbell_amd.h", "barbell_amd_synth.h", "barbell_amd_filter.h", "barbell_amd_trim.h", "barbell_amd_inspect.h", "barbell_amd_fastq.h", "barbell_amd_format.h"):
        src = open(os.path.join(ROOT, "include", h)).read()
        src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
        declared |= set(re.findall(r"\b(bb_[a-z_0-9]+)\s*\(", src))
    assert declared == set(_lib.EXPORTS)
    for s in declared:
        assert hasattr(L, s), s


def test_row_layout():
    assert _abi.ROW_DTYPE.itemsize == 48
    assert C.sizeof(_abi.GroupInfo) == 56


def test_build_holds_the_feasible_trace_classes(L):
    """bb_build_trace_classes (no GPU needed): the default build holds fast barcode kernels for the traceback classes the reference's own
    known-answer tests leave open (tests/golden/policy_feasible.json), the default order (class 0) among them; `make CLASSES=all` holds all 18."""
    from tests.common import policy_feasible

    m = L.bb_build_trace_classes()
    held = [i for i in range(18) if (m >> i) & 1]
    assert 0 in held and m < (1 << 18)
    assert set(policy_feasible()["feasible_trace_class_indices"]) <= set(held)


def test_strerror(L):
    assert L.bb_strerror(0) == b"ok"
    assert b"IUPAC" in L.bb_strerror(_abi.BB_E_NOT_IUPAC)


def test_create_validates_before_touching_the_gpu(L):
    # reference panics (barcodes.rs:113-133,325-328,45-47) surface as error codes even without a GPU
    def create(groups):
        arr, keep = _abi.make_group_descs(groups)
        p = _abi.Params(0.4, 0.2, 0.1, 0)
        h = C.c_void_p()
        return L.bb_create(arr, len(groups), C.byref(p), C.byref(h)), h

    assert create([([b"AAATTTGGG"], 0, None)])[0] == _abi.BB_E_ONE_QUERY
    assert create([([b"AAATTTGGG", b"AAAAAAACCCGGG"], 0, None)])[0] == _abi.BB_E_UNEQUAL_LEN
    assert create([([b"@@@@@@@@@", b"AAACCCGGG"], 0, None)])[0] == _abi.BB_E_NOT_IUPAC
    assert create([([b"CAATTTGGT", b"AAACCCGGG"], 0, None)])[0] == _abi.BB_E_NO_FLANK
    assert create([([b"AAACCCGGG", b"AAACCCGGG"], 0, None)])[0] == _abi.BB_E_NO_BARCODE
    # geometry limits (include/barbell_amd.h): what the reference accepts and only the any-geometry kernels compute is accepted
    # (rc NO_DEVICE here, no GPU) — a 231-nt flank with its automatic cutoff 92, 70-nt padded barcodes; beyond the tables' sizes
    # the refusal leaves its reason for bb_last_error(NULL)
    ok = (_abi.BB_OK, _abi.BB_E_NO_DEVICE)
    assert create([([b"A" * 200 + b"C" + b"G" * 30, b"A" * 200 + b"T" + b"G" * 30], 0, 20)])[0] in ok
    assert create([([b"A" * 200 + b"C" + b"G" * 30, b"A" * 200 + b"T" + b"G" * 30], 0, None)])[0] in ok          # automatic cutoff 92
    assert create([([b"ACGTACGTAC" + b"C" * 50 + b"GTGTGTGTGT", b"ACGTACGTAC" + b"T" * 50 + b"GTGTGTGTGT"], 0, None)])[0] in ok  # 70-nt patterns
    assert create([([b"A" * 200 + b"C" + b"G" * 30, b"A" * 200 + b"T" + b"G" * 30], 0, 128)])[0] == _abi.BB_E_UNSUPPORTED
    assert b"error budget 128" in L.bb_last_error(None)
    assert create([([b"A" * 250 + b"C" + b"G" * 30, b"A" * 250 + b"T" + b"G" * 30], 0, None)])[0] == _abi.BB_E_UNSUPPORTED
    assert b"flank of 281 nt" in L.bb_last_error(None)
    assert create([([b"ACGTACGTAC" + b"C" * 110 + b"GTGTGTGTGT", b"ACGTACGTAC" + b"T" * 110 + b"GTGTGTGTGT"], 0, None)])[0] == _abi.BB_E_UNSUPPORTED
    assert b"padded barcode pattern of 130 nt" in L.bb_last_error(None)
    assert create([([b"AAATTTGGG", b"AAACTTGGG"], 0, None)] * 33)[0] == _abi.BB_E_UNSUPPORTED and b"32 query groups" in L.bb_last_error(None)


def test_no_cpu_fallback(L):
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    arr, keep = _abi.make_group_descs([([b"AAATTTGGG", b"AAACCCGGG"], 0, None)])
    p = _abi.Params(0.4, 0.2, 0.1, 0)
    h = C.c_void_p()
    assert L.bb_create(arr, 1, C.byref(p), C.byref(h)) == _abi.BB_E_NO_DEVICE


def test_product_does_not_reference_oracle():
    for dp, _, files in os.walk(os.path.join(ROOT, "barbell_amd")):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".sh", ".cpp")):
                txt = open(os.path.join(dp, f)).read()
                assert "pyoracle" not in txt and "bb_oracle" not in txt and "libbb_oracle" not in txt, f


def test_synth_deterministic_and_shardable():
    from barbell_amd import annotate as A

    g = kits.groups_from_kit("SQK-NBD114-96", flank_max_errors=3)
    b1, o1 = A.synth_reads_host(g, 7, 300, 900, 0, 64)
    b2, o2 = A.synth_reads_host(g, 7, 300, 900, 0, 64)
    assert b1.tobytes() == b2.tobytes() and (o1 == o2).all()
    b3, o3 = A.synth_reads_host(g, 7, 300, 900, 32, 32)  # a later shard equals the tail of the stream
    assert b3.tobytes() == b1[int(o1[32]):].tobytes()
    assert set(np.unique(b1).tolist()) <= set(b"ACGT")
    b4, _ = A.synth_reads_host(g, 8, 300, 900, 0, 64)
    assert b4.tobytes() != b1.tobytes()
