"""annotation.tsv lines rendered on the GPU — host side of include/barbell_amd_format.h.

`RowFormatter(dm, groups)` installs the label strings (they never cross bb_create) and turns the rows of the batch
FASTQ ingest left in HBM into the bytes of the TSV lines: `BarbellMatch`'s serde layout written by the csv crate
(src/annotate/searcher.rs:31-142, annotator.rs:13-26), with the `cuts` column of filtered.tsv when verdicts are given
(filter.rs:87-119).  `annotate.format_rows` is the host-side statement of the same format (tests compare the two)."""
import ctypes as C

import numpy as np

from . import _abi
from ._lib import lib

FMT_ALL, FMT_KEPT, FMT_DROPPED = 0, 1, 2


class RowFormatter:
    def __init__(self, demuxer, groups):
        self.dm = demuxer
        labels = []
        for g in groups:
            labels += [l.encode() for l in g.labels] + [b"flank"]
        off = np.zeros(len(labels) + 1, dtype=np.uint32)
        off[1:] = np.cumsum([len(l) for l in labels])
        blob = np.frombuffer(b"".join(labels) + b"\0", dtype=np.uint8)
        n = lib().bb_counts_len(demuxer._ctx())
        if n != len(labels):
            raise ValueError(f"{len(labels)} labels for {n} histogram slots")
        demuxer._check(lib().bb_format_set_labels(demuxer._ctx(), blob.ctypes.data, off.ctypes.data))

    def render(self, d_rows, n_rows, batch, mode=FMT_ALL, d_verdicts=None):
        """-> (bytes of the TSV lines without header, number of lines); `batch` = the FastqBatchDev of the ingested block"""
        if n_rows == 0:
            return b"", 0
        buf = self.dm.buf("tsv%d" % mode)
        cap = max(1 << 16, 160 * int(n_rows))
        while True:
            d = buf.ensure(cap)
            tl, nl = C.c_uint64(), C.c_uint64()
            rc = lib().bb_format_rows_dev(self.dm._ctx(), d_rows, d_verdicts, n_rows, mode, C.byref(batch.d_headers), d, cap, C.byref(tl), C.byref(nl))
            if rc == _abi.BB_E_CAPACITY:
                cap = int(tl.value)
                continue
            self.dm._check(rc)
            out = np.empty(int(tl.value), dtype=np.uint8)
            buf.download(out)
            return out.tobytes(), int(nl.value)
