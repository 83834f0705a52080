"""bb_pack_bases (include/barbell_amd.h): the two-bases-per-byte form of the boundary, against a byte-wise packer — no GPU needed."""
import ctypes as C

import numpy as np

from barbell_amd._lib import lib

CODES = {}
for letter, code in zip("ACGTURYSWKMBDHVN", (1, 2, 4, 8, 8, 5, 10, 6, 9, 12, 3, 14, 13, 11, 7, 15)):
    CODES[ord(letter)] = code
    CODES[ord(letter.lower())] = code


def py_pack(b):
    c = [CODES.get(x, 0) for x in b]
    if len(c) & 1:
        c.append(15)
    return bytes(((c[i] << 4) | (c[i + 1] ^ 0xA)) & 0xFF for i in range(0, len(c), 2))


def test_pack_bases_equals_the_byte_wise_packer():
    L = lib()
    rng = np.random.default_rng(11)
    alphabet = np.frombuffer(b"ACGTacgtUuNnRYSWKMBDHVryswkmbdhvX-@ \n\r09*", dtype=np.uint8)
    for n in list(range(0, 70)) + [127, 128, 129, 1000, 4001, 65537]:
        src = rng.choice(alphabet, n).astype(np.uint8)
        if n > 40:
            src[: n // 2] = rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), n // 2)   # (mostly plain bases, as reads are)
        for shift in (0, 1, 3):   # unaligned sources
            buf = np.zeros(n + 8, dtype=np.uint8)
            buf[shift: shift + n] = src
            out = np.full((n + 1) // 2 + 8, 0xEE, dtype=np.uint8)
            wrote = L.bb_pack_bases(buf.ctypes.data + shift, n, out.ctypes.data)
            assert wrote == (n + 1) // 2
            assert out[:wrote].tobytes() == py_pack(src.tobytes()), (n, shift)
            assert np.all(out[wrote:] == 0xEE)   # nothing written beyond
