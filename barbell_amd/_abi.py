"""ctypes mirror of include/barbell_amd.h (struct layouts shared by the product binding and the
test-only oracle binding)."""
import ctypes as C

import numpy as np

BB_OK = 0
BB_E_INVALID, BB_E_ONE_QUERY, BB_E_UNEQUAL_LEN, BB_E_NO_BARCODE, BB_E_NO_FLANK = -1, -2, -3, -4, -5
BB_E_NOT_IUPAC, BB_E_CAPACITY, BB_E_NO_DEVICE, BB_E_HIP, BB_E_UNSUPPORTED, BB_E_NOMEM = -6, -7, -8, -9, -10, -11
BB_E_FASTQ = -12

BB_FTAG, BB_RTAG, BB_FFLANK, BB_RFLANK = 0, 1, 2, 3
BB_FWD, BB_RC = 0, 1
MATCH_TYPE_STR = ("Ftag", "Rtag", "Fflank", "Rflank")  # barcodes.rs:25-32
STRAND_STR = ("Fwd", "Rc")  # searcher.rs:67-75


class GroupDesc(C.Structure):
    _fields_ = [
        ("seqs", C.POINTER(C.c_char_p)),
        ("seq_lens", C.POINTER(C.c_uint32)),
        ("n_seqs", C.c_uint32),
        ("type", C.c_uint8),
        ("flank_k", C.c_int32),
    ]


class Params(C.Structure):
    _fields_ = [
        ("alpha", C.c_float),
        ("min_score", C.c_double),
        ("min_score_diff", C.c_double),
        ("device", C.c_int32),
    ]


class GroupInfo(C.Structure):
    _fields_ = [
        ("flank_len", C.c_uint32), ("prefix_len", C.c_uint32), ("suffix_len", C.c_uint32), ("mask_len", C.c_uint32),
        ("bar_lo", C.c_uint32), ("bar_hi", C.c_uint32), ("pad_lo", C.c_uint32), ("pad_hi", C.c_uint32),
        ("pattern_len", C.c_uint32), ("flank_k", C.c_int32), ("bar_k1", C.c_int32), ("bar_k2", C.c_int32),
        ("perfect_score", C.c_double),
    ]


# bb_row as a numpy structured dtype (48 bytes)
ROW_DTYPE = np.dtype(
    [
        ("read_idx", "<u4"), ("read_len", "<u4"), ("rel_dist_to_end", "<i4"),
        ("read_start_bar", "<u4"), ("read_end_bar", "<u4"),
        ("read_start_flank", "<u4"), ("read_end_flank", "<u4"),
        ("bar_start", "<u4"), ("bar_end", "<u4"),
        ("flank_cost", "<i2"), ("barcode_cost", "<i2"), ("barcode_idx", "<i2"),
        ("group_idx", "u1"), ("match_type", "u1"), ("strand", "u1"), ("_pad", "u1", (3,)),
    ]
)
assert ROW_DTYPE.itemsize == 48


def make_group_descs(groups):
    """groups: list of (seqs: list[bytes], type: int, flank_k: int|None).  Returns (array, keepalive)."""
    arr = (GroupDesc * len(groups))()
    keep = []
    for i, (seqs, typ, fk) in enumerate(groups):
        seqs = [bytes(s) for s in seqs]
        lens = (C.c_uint32 * max(1, len(seqs)))(*[len(s) for s in seqs])
        ptrs = (C.c_char_p * max(1, len(seqs)))(*seqs)
        keep.append((seqs, ptrs, lens))
        arr[i].seqs = C.cast(ptrs, C.POINTER(C.c_char_p))
        arr[i].seq_lens = C.cast(lens, C.POINTER(C.c_uint32))
        arr[i].n_seqs = len(seqs)
        arr[i].type = typ
        arr[i].flank_k = -1 if fk is None else int(fk)
    return arr, keep


def pack_reads(reads):
    """list[bytes] -> (bases uint8[n_total], offsets uint64[n+1])"""
    offsets = np.zeros(len(reads) + 1, dtype=np.uint64)
    if reads:
        offsets[1:] = np.cumsum([len(r) for r in reads], dtype=np.uint64)
    bases = np.frombuffer(b"".join(reads), dtype=np.uint8).copy() if reads else np.zeros(0, dtype=np.uint8)
    return bases, offsets
