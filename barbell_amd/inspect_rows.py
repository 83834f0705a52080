"""Inspect step — host mirror of src/inspect/inspect.rs over include/barbell_amd_inspect.h.

The GPU computes one pattern element per row (`get_group_structure`, inspect.rs:15-117); this module
joins the elements of a read into the pattern string, writes pattern_per_read.tsv and keeps the pattern
counts of `inspect` (inspect.rs:128-208)."""
from collections import Counter

import numpy as np

from . import _abi

INSPECT_DTYPE = np.dtype([("match_type", "u1"), ("strand", "u1"), ("has_cut", "u1"), ("tag", "u1"), ("lo", "<u4"), ("hi", "<u4"),
                          ("first", "<u4")])
assert INSPECT_DTYPE.itemsize == 16
_TAG = {1: "@left", 2: "@right", 3: "@prev_left"}


def elements(dm, rows, verdicts=None, bucket_size=250):
    """rows (+ verdicts) -> INSPECT_DTYPE array, computed by the HIP library"""
    from ._lib import lib

    rows = np.ascontiguousarray(rows, dtype=_abi.ROW_DTYPE)
    out = np.zeros(len(rows), dtype=INSPECT_DTYPE)
    v = None if verdicts is None else np.ascontiguousarray(verdicts)
    dm._check(lib().bb_inspect_rows(dm._ctx(), rows.ctypes.data, None if v is None else v.ctypes.data, len(rows), bucket_size,
                                    out.ctypes.data))
    return out


def elements_dev(dm, d_rows, d_verdicts, n_rows, bucket_size=250):
    """same for rows (and verdicts) that are already in HBM"""
    from ._lib import lib

    out = np.zeros(n_rows, dtype=INSPECT_DTYPE)
    if n_rows:
        d = dm.buf("elems").ensure(n_rows * 16)
        dm._check(lib().bb_inspect_rows_dev(dm._ctx(), d_rows, d_verdicts, n_rows, bucket_size, d))
        dm.buf("elems").download(out)
    return out


def element_str(e):  # inspect.rs:90-101
    cut = (", >>" if e["strand"] else ", <<") if e["has_cut"] else ""
    return f"{_abi.MATCH_TYPE_STR[int(e['match_type'])]}[{'rc' if e['strand'] else 'fw'}, *{cut}, {_TAG[int(e['tag'])]}({int(e['lo'])}..{int(e['hi'])})]"


def patterns(elems, rows):
    """-> list of (read_idx, pattern string), one per read with rows, in row order"""
    out = []
    cache = {}
    cur, parts = None, []
    for e, r in zip(elems, rows["read_idx"]):
        if e["first"] and parts:
            out.append((cur, "__".join(parts)))
            parts = []
        cur = int(r)
        k = e.tobytes()[:12]
        s = cache.get(k)
        if s is None:
            s = cache[k] = element_str(e)
        parts.append(s)
    if parts:
        out.append((cur, "__".join(parts)))
    return out


class Inspector:
    """inspect(annotated_file, top_n, read_pattern_out, bucket_size) fed batch by batch"""

    def __init__(self, dm, read_pattern_out=None, bucket_size=250):
        self.dm, self.bucket_size = dm, bucket_size
        self.counts = Counter()
        self.out = open(read_pattern_out, "w") if read_pattern_out else None

    def add(self, rows, read_ids, verdicts=None, d_rows=None):
        el = elements_dev(self.dm, d_rows, None, len(rows), self.bucket_size) if d_rows is not None else \
            elements(self.dm, rows, verdicts, self.bucket_size)
        pats = patterns(el, rows)
        self.counts.update(p for _, p in pats)
        if self.out is not None:
            self.out.write("".join(f"{read_ids[i]}\t{p}\n" for i, p in pats))

    def close(self):
        if self.out is not None:
            self.out.close()

    def summary(self, top_n=10):
        """the lines `inspect` prints (inspect.rs:186-205), colours left out"""
        lines = [f"Found {len(self.counts)} unique patterns"]
        for i, (p, c) in enumerate(sorted(self.counts.items(), key=lambda kv: -kv[1])[:top_n]):
            lines.append(f"\tPattern {i + 1}: {c} occurrences")
            lines.append(f"\t\t{p}")
        lines.append(f"Showed {top_n} / {len(self.counts)} patterns")
        return lines
