"""FASTQ ingest on the GPU — host side of include/barbell_amd_fastq.h.

`ingest(dm, text)` hands one block of raw FASTQ text to the HIP library, which finds the records and packs
sequences, qualities and headers into the batch layout of the other entry points, in HBM.  `BlockReader`
cuts plain or gzip FASTQ files into such blocks (carrying the partial record at a block's end over to the
next one), replacing the per-record reader loop of annotator.rs:245-262 / trim.rs:364-384."""
import ctypes as C
import gzip

import numpy as np

from . import _abi
from .trim import HeadersC


class FastqInfo(C.Structure):
    _fields_ = [("n_records", C.c_uint64), ("consumed", C.c_uint64), ("n_bases", C.c_uint64), ("n_hdr", C.c_uint64),
                ("bad_record", C.c_int64)]


class FastqBatchDev(C.Structure):
    _fields_ = [("d_bases", C.c_void_p), ("d_quals", C.c_void_p), ("d_offsets", C.c_void_p), ("d_headers", HeadersC)]


BB_FASTQ_FINAL, BB_FASTQ_TWO_LINE, BB_FASTQ_PACKED = 1, 2, 4   # include/barbell_amd_fastq.h
CANON = b"-ACMGRSVTWYHKDBN"                                    # what a packed base set unpacks to, by code
_CODE = np.zeros(256, dtype=np.uint8)
for _ch, _k in zip(b"ACGTURYSWKMBDHVN", (1, 2, 4, 8, 8, 5, 10, 6, 9, 12, 3, 14, 13, 11, 7, 15)):
    _CODE[_ch] = _CODE[_ch | 0x20] = _k


def base_codes(seq):
    """4-bit IUPAC base set of every read character (A=1 C=2 G=4 T=8 and unions, case-insensitive, U = T; 0 = not an IUPAC letter)"""
    return _CODE[np.frombuffer(bytes(seq), dtype=np.uint8)]


def pack_sequence_line(seq):
    """the packed form of one sequence line (BB_FASTQ_PACKED), without the '\n'; None where the form cannot hold it (two adjacent
    non-IUPAC characters at an even position would pack to the byte '\n')"""
    c = base_codes(seq).astype(np.uint16)
    n = len(c)
    if n & 1:
        c = np.concatenate([c, np.array([15], dtype=np.uint16)])
    hi, lo = c[0::2], c[1::2]
    if np.any((hi == 0) & (lo == 0)):
        return None
    return ((hi << 4) | (lo ^ 0xA)).astype(np.uint8).tobytes() + (b"O" if n & 1 else b"E")


def pack_two_line(records, nl=b"\n"):
    """records: (header without '@', sequence) pairs -> the text of a BB_FASTQ_TWO_LINE | BB_FASTQ_PACKED block (what the C++ host's readers
    stage); None if a sequence cannot be packed"""
    out = []
    for h, seq in records:
        p = pack_sequence_line(seq)
        if p is None:
            return None
        out.append(b"@" + bytes(h) + nl + p + b"\n")
    return b"".join(out)


def ingest(dm, text, final_block=True, device_ptr=None):
    """text: bytes-like block (or `device_ptr` + length given as text=int).  -> (FastqInfo, FastqBatchDev)"""
    from ._lib import lib
    from .annotate import BarbellError

    info, batch = FastqInfo(), FastqBatchDev()
    if device_ptr is not None:
        rc = lib().bb_fastq_ingest_dev(dm._ctx(), device_ptr, int(text), int(final_block), C.byref(info), C.byref(batch))
    else:
        buf = np.frombuffer(text, dtype=np.uint8)
        rc = lib().bb_fastq_ingest(dm._ctx(), buf.ctypes.data, len(buf), int(final_block), C.byref(info), C.byref(batch))
    if rc == _abi.BB_E_FASTQ:
        raise BarbellError(rc, lib().bb_last_error(dm._ctx()).decode())
    dm._check(rc)
    return info, batch


def fetch(dm, info, bases=False, quals=False):
    """host copies of the last ingested batch: dict with offsets, hdr, hdr_offsets, id_len, desc_start [, bases, quals]"""
    from ._lib import lib

    n = int(info.n_records)
    a = {"offsets": np.zeros(n + 1, np.uint64), "hdr": np.zeros(int(info.n_hdr), np.uint8), "hdr_offsets": np.zeros(n + 1, np.uint64),
         "id_len": np.zeros(n, np.uint32), "desc_start": np.zeros(n, np.uint32)}
    if bases:
        a["bases"] = np.zeros(int(info.n_bases), np.uint8)
    if quals:
        a["quals"] = np.zeros(int(info.n_bases), np.uint8)
    ptr = lambda k: a[k].ctypes.data if k in a else None
    dm._check(lib().bb_fastq_fetch(dm._ctx(), ptr("offsets"), ptr("hdr"), ptr("hdr_offsets"), ptr("id_len"), ptr("desc_start"), ptr("bases"),
                                   ptr("quals")))
    return a


def fetch_lines(dm, info, lines_per_record=4):
    """bb_fastq_fetch_lines: offsets of the line ends of the last ingested block's records (lines_per_record * n_records)"""
    from ._lib import lib

    out = np.zeros(int(info.n_records) * lines_per_record, dtype=np.uint64)
    dm._check(lib().bb_fastq_fetch_lines(dm._ctx(), out.ctypes.data))
    return out


def read_ids(a):
    """read ids (header up to the first whitespace) of a fetched batch"""
    from .annotate import BarbellError

    blob, off, idl = a["hdr"].tobytes(), a["hdr_offsets"], a["id_len"]
    try:
        return [blob[int(off[i]): int(off[i]) + int(idl[i])].decode() for i in range(len(idl))]
    except UnicodeDecodeError as e:  # the reference: "FASTQ header is not valid UTF-8" (annotator.rs:124-125)
        raise BarbellError(_abi.BB_E_FASTQ, f"FASTQ header is not valid UTF-8 ({e})") from None


def is_gzip(path):
    """gzip by its magic bytes, not by its name (the C++ host's gzopen and the reference's reader sniff too)"""
    with open(path, "rb") as f:
        return f.read(2) == b"\x1f\x8b"


class BlockReader:
    """yields (block bytes, is_final) over the concatenation of FASTQ files; every file ends on a record
    boundary, so a file's last block is final for the parser"""

    def __init__(self, paths, block_bytes=256 << 20):
        self.paths, self.block_bytes = list(paths), block_bytes

    def __iter__(self):
        for path in self.paths:
            op = gzip.open if is_gzip(path) else open
            with op(path, "rb") as f:
                nxt = f.read(self.block_bytes)
                while True:
                    cur, nxt = nxt, f.read(self.block_bytes)
                    yield cur, not nxt
                    if not nxt:
                        break


def batches(dm, paths, block_bytes=256 << 20):
    """ingests the files block by block; yields (FastqInfo, FastqBatchDev) per block that holds >= 1 record.
    The partial record at the end of a block is carried over to the next block of the same file."""
    carry = b""
    for block, final in BlockReader(paths, block_bytes):
        text = carry + block if carry else block
        info, batch = ingest(dm, text, final)
        carry = bytes(text[int(info.consumed):])
        if info.n_records:
            yield info, batch
