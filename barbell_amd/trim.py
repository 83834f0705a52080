"""Trim/split step on a batch — host mirror of src/trim/trim.rs over include/barbell_amd_trim.h.

`TrimConfig` is the reference's (config.rs:19-32, CLI defaults bin/main.rs:136-186).  The GPU library
plans the slices of every passing read (`preprocess_cuts`, trim.rs:127-254), cuts / flips them and
renders the FASTQ records grouped by output label; what is left here is `LabelConfig::create_label`'s
string formatting (trim.rs:58-105) on the label keys and the per-label file writers of `trim_matches`
(trim.rs:317-480).  No read bytes are touched on the host."""
import ctypes as C
import gzip
import os
import re
from dataclasses import dataclass

import numpy as np

from . import _abi
from .filter import VERDICT_DTYPE, slot_labels

SIDE = {None: 0, "left": 1, "right": 2}
TRIM_NONE, TRIM_TRIMMED, TRIM_FAILED = 0, 1, 2


@dataclass
class TrimConfig:  # config.rs:19-32
    add_labels: bool = True
    add_orientation: bool = True
    add_flank: bool = True
    sort_labels: bool = False
    only_side: str = None            # None | "left" | "right"
    failed_trimmed_writer: str = None
    write_full_header: bool = True
    skip_trim: bool = False
    flip: bool = False
    verbose: bool = False
    gzip: bool = False

    @staticmethod
    def for_kit(failed_out=None, gzip=False):  # use_kit.rs:87-99
        return TrimConfig(True, False, False, False, "left", failed_out, True, False, False, False, gzip)


class TrimConfigC(C.Structure):
    _fields_ = [(n, C.c_uint8) for n in ("add_labels", "add_orientation", "add_flank", "sort_labels", "only_side",
                                         "write_full_header", "skip_trim", "flip")]


class HeadersC(C.Structure):
    _fields_ = [("hdr", C.c_void_p), ("hdr_offsets", C.c_void_p), ("id_len", C.c_void_p), ("desc_start", C.c_void_p)]


SLICE_DTYPE = np.dtype([("read_idx", "<u4"), ("start", "<u4"), ("end", "<u4"), ("label_key", "<u4"), ("suffix", "<u2"),
                        ("flip", "u1"), ("_pad", "u1"), ("rec_len", "<u4"), ("out_off", "<u8")])
SPAN_DTYPE = np.dtype([("label_key", "<u4"), ("n_records", "<u4"), ("first", "<u8"), ("off", "<u8"), ("len", "<u8")])
assert SLICE_DTYPE.itemsize == 32 and SPAN_DTYPE.itemsize == 32


def config_c(cfg):
    if cfg.sort_labels and cfg.only_side is not None:  # trim.rs:330-334
        raise ValueError("Cannot enable only keeping left/right label and sorting; this is ambiguous")
    return TrimConfigC(cfg.add_labels, cfg.add_orientation, cfg.add_flank, cfg.sort_labels, SIDE[cfg.only_side],
                       cfg.write_full_header, cfg.skip_trim, cfg.flip)


class LabelTables:
    """label ids exactly as filter.compile_patterns assigns them, plus the two tables bb_trim_set wants"""

    def __init__(self, groups, cfg):
        ids = {}
        self.label_ids = np.array([ids.setdefault(l, len(ids)) for l in slot_labels(groups)], dtype=np.uint32)
        self.labels = list(ids)
        self.is_flank = np.array(["flank" in l for l in self.labels], dtype=np.uint8)
        parts = [self.part_str(i * 2 + s, cfg) for i in range(len(self.labels)) for s in (0, 1)]
        order = {p: r for r, p in enumerate(sorted(set(parts), key=lambda x: x.encode()))}
        self.part_rank = np.array([order[p] for p in parts], dtype=np.uint32)

    def part_str(self, part, cfg):  # trim.rs:70-80
        return self.labels[part >> 1] + (("_rc" if part & 1 else "_fw") if cfg.add_orientation else "")

    def label_of_key(self, key, cfg):
        """the group label of trim.rs:289 from a bb_slice.label_key"""
        if key == 0:
            return "none"
        p = [(key >> 16) - 1] + ([(key & 0xFFFF) - 1] if key & 0xFFFF else [])
        return "__".join(self.part_str(x, cfg) for x in p)


from .annotate import _WS, split_fastq_header  # noqa: E402  (char::is_whitespace split of io.rs:6-17)


def pack_headers(headers):
    """headers: list of header lines (bytes, without '@').  -> (blob, offsets, id_len, desc_start) with the
    split of split_fastq_header (io.rs:6-17): id up to the first whitespace, description = rest, left-trimmed."""
    n = len(headers)
    off = np.zeros(n + 1, dtype=np.uint64)
    np.cumsum([len(h) for h in headers], out=off[1:])
    id_len = np.zeros(n, dtype=np.uint32)
    desc = np.zeros(n, dtype=np.uint32)
    for i, h in enumerate(headers):
        sp = h.find(b" ")
        if sp >= 0 and h.isascii() and not any(c in h[:sp] for c in b"\t\n\x0b\x0c\r"):
            j = sp  # fast path: first whitespace is a space
            while j < len(h) and h[j] in b"\t\n\x0b\x0c\r ":
                j += 1
            id_len[i], desc[i] = sp, j
            continue
        rid, d = split_fastq_header(h.decode("utf-8"))
        id_len[i], desc[i] = len(rid.encode()), len(h) - len(d.encode())
    return np.frombuffer(b"".join(headers), dtype=np.uint8), off, id_len, desc


@dataclass
class TrimResult:
    text: np.ndarray      # uint8
    slices: np.ndarray    # SLICE_DTYPE, text order
    spans: np.ndarray     # SPAN_DTYPE, text order
    status: np.ndarray    # uint8 per read: TRIM_*


class Trimmer:
    """process_read_and_anno for every passing read of a batch, on the GPU of `demuxer`.  Needs the
    filter of the same demuxer installed (label ids come from it)."""

    def __init__(self, demuxer, cfg):
        from ._lib import lib

        self.dm, self.cfg = demuxer, cfg
        self.tables = LabelTables(demuxer.queries, cfg)
        c = config_c(cfg)
        demuxer._check(lib().bb_trim_set(demuxer._ctx(), C.byref(c), self.tables.is_flank.ctypes.data,
                                         self.tables.part_rank.ctypes.data, len(self.tables.labels)))

    def trim_batch(self, rows, verdicts, bases, quals, offsets, headers):
        from ._lib import lib

        rows = np.ascontiguousarray(rows, dtype=_abi.ROW_DTYPE)
        verdicts = np.ascontiguousarray(verdicts, dtype=VERDICT_DTYPE)
        bases = np.ascontiguousarray(bases, dtype=np.uint8)
        quals = np.ascontiguousarray(quals, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        n = len(offsets) - 1
        blob, hoff, id_len, desc = headers if isinstance(headers, tuple) else pack_headers(headers)
        blob = np.ascontiguousarray(blob)
        h = HeadersC(blob.ctypes.data, hoff.ctypes.data, id_len.ctypes.data, desc.ctypes.data)
        status = np.zeros(n, dtype=np.uint8)
        tcap, scap, pcap = 2 * len(bases) + 64 * n + 1024, 2 * n + 64, 4096
        while True:
            text = np.empty(tcap, dtype=np.uint8)
            slices = np.zeros(scap, dtype=SLICE_DTYPE)
            spans = np.zeros(pcap, dtype=SPAN_DTYPE)
            tl, ns, nsp = C.c_uint64(), C.c_uint64(), C.c_uint32()
            rc = lib().bb_trim_batch(self.dm._ctx(), rows.ctypes.data, verdicts.ctypes.data, len(rows), bases.ctypes.data,
                                     quals.ctypes.data, offsets.ctypes.data, C.byref(h), n, text.ctypes.data, tcap, C.byref(tl),
                                     slices.ctypes.data, scap, C.byref(ns), spans.ctypes.data, pcap, C.byref(nsp), status.ctypes.data)
            if rc == _abi.BB_E_CAPACITY:
                tcap, scap, pcap = max(tcap, tl.value), max(scap, ns.value), max(pcap, nsp.value)
                continue
            self.dm._check(rc)
            return TrimResult(text[: tl.value], slices[: ns.value], spans[: nsp.value], status)


    def trim_batch_dev(self, d_rows, d_verdicts, n_rows, d_bases, d_quals, d_offsets, d_hdr, d_hdr_offsets, d_id_len, d_desc_start,
                       n_reads, d_text, text_cap, d_slices, slices_cap, d_spans, spans_cap, d_status):
        """device-pointer variant (ints = HIP device pointers) -> (text_len, n_slices, n_spans)"""
        from ._lib import lib

        h = HeadersC(d_hdr, d_hdr_offsets, d_id_len, d_desc_start)
        tl, ns, nsp = C.c_uint64(), C.c_uint64(), C.c_uint32()
        rc = lib().bb_trim_batch_dev(self.dm._ctx(), d_rows, d_verdicts, n_rows, d_bases, d_quals, d_offsets, C.byref(h), n_reads, d_text,
                                     text_cap, C.byref(tl), d_slices, slices_cap, C.byref(ns), d_spans, spans_cap, C.byref(nsp), d_status)
        if rc == _abi.BB_E_CAPACITY:
            from .annotate import BarbellError

            e = BarbellError(rc, f"need text {tl.value} slices {ns.value} spans {nsp.value}")
            e.need = (int(tl.value), int(ns.value), int(nsp.value))
            raise e
        self.dm._check(rc)
        return int(tl.value), int(ns.value), int(nsp.value)

    def trim_ingested(self, d_rows, d_verdicts, n_rows, batch, info):
        """trim the batch bb_fastq_ingest left in HBM with rows / verdicts that are there too -> TrimResult"""
        n = int(info.n_records)
        dm = self.dm
        tcap, scap, pcap = 2 * int(info.n_bases) + 2 * int(info.n_hdr) + 32 * n + 1024, 2 * n + 64, 4096
        d_st = dm.buf("status").ensure(n + 16)
        h = batch.d_headers
        while True:
            try:
                tl, ns, nsp = self.trim_batch_dev(d_rows, d_verdicts, n_rows, batch.d_bases, batch.d_quals, batch.d_offsets, h.hdr, h.hdr_offsets,
                                                  h.id_len, h.desc_start, n, dm.buf("text").ensure(tcap), tcap, dm.buf("slices").ensure(scap * 32),
                                                  scap, dm.buf("spans").ensure(pcap * 32), pcap, d_st)
                break
            except Exception as e:  # BB_E_CAPACITY carries the needed sizes
                if getattr(e, "code", None) != _abi.BB_E_CAPACITY:
                    raise
                tcap, scap, pcap = max(tcap, e.need[0]), max(scap, e.need[1]), max(pcap, e.need[2])
        return TrimResult(dm.buf("text").download(np.empty(tl, np.uint8)), dm.buf("slices").download(np.zeros(ns, SLICE_DTYPE)),
                          dm.buf("spans").download(np.zeros(nsp, SPAN_DTYPE)), dm.buf("status").download(np.zeros(n, np.uint8)))

    def plan_ingested(self, d_rows, d_verdicts, n_rows, batch, info):
        """bb_trim_plan_dev on the ingested batch: everything trim decides, none of the record text -> TrimResult whose
        `text` is None and `text_len` the bytes the records take.  `cut_records` turns it into the same bytes on the host."""
        from ._lib import lib

        n = int(info.n_records)
        dm = self.dm
        scap, pcap = 2 * n + 64, 4096
        d_st = dm.buf("status").ensure(n + 16)
        tl, ns, nsp = C.c_uint64(), C.c_uint64(), C.c_uint32()
        while True:
            rc = lib().bb_trim_plan_dev(dm._ctx(), d_rows, d_verdicts, n_rows, batch.d_offsets, C.byref(batch.d_headers), n, C.byref(tl),
                                        dm.buf("slices").ensure(scap * 32), scap, C.byref(ns), dm.buf("spans").ensure(pcap * 32), pcap,
                                        C.byref(nsp), d_st)
            if rc != _abi.BB_E_CAPACITY:
                break
            scap, pcap = max(scap, ns.value), max(pcap, nsp.value)
        dm._check(rc)
        res = TrimResult(None, dm.buf("slices").download(np.zeros(ns.value, SLICE_DTYPE)), dm.buf("spans").download(np.zeros(nsp.value, SPAN_DTYPE)),
                         dm.buf("status").download(np.zeros(n, np.uint8)))
        res.text_len = int(tl.value)
        return res

    def last_ms(self):
        from ._lib import lib

        return {k: lib().bb_trim_last_ms(self.dm._ctx(), i) for i, k in enumerate(("plan_sort", "render", "total"))}


class LabelWriters:
    """the per-label writers of trim_matches (trim.rs:356-446): '{folder}/{label}.trimmed.fastq[.gz]',
    created on first use, appended to per batch"""

    def __init__(self, output_folder, cfg, tables):
        os.makedirs(output_folder, exist_ok=True)
        self.folder, self.cfg, self.tables = output_folder, cfg, tables
        self.writers = {}
        self.failed = open(cfg.failed_trimmed_writer, "w") if cfg.failed_trimmed_writer else None
        self.n_trimmed = self.n_failed = self.n_split = 0

    def write(self, result, read_ids):
        mv = memoryview(result.text)
        for sp in result.spans:
            label = self.tables.label_of_key(int(sp["label_key"]), self.cfg)
            w = self.writers.get(label)
            if w is None:
                path = os.path.join(self.folder, label + (".trimmed.fastq.gz" if self.cfg.gzip else ".trimmed.fastq"))
                w = self.writers[label] = gzip.open(path, "wb") if self.cfg.gzip else open(path, "wb")
            w.write(mv[int(sp["off"]): int(sp["off"] + sp["len"])])
        self.n_trimmed += int((result.status == TRIM_TRIMMED).sum())
        failed = np.nonzero(result.status == TRIM_FAILED)[0]
        self.n_failed += len(failed)
        if len(result.slices):
            self.n_split += int((np.bincount(result.slices["read_idx"]) > 1).sum())
        if self.failed is not None:
            for i in failed:
                self.failed.write(read_ids[int(i)] + "\n")

    def close(self):
        for w in self.writers.values():
            w.close()
        if self.failed is not None:
            self.failed.close()


_COMP = bytes.maketrans(b"ATCGRYKMBVDHatcgrykmbvdh", b"TAGCYRMKVBHDtagcyrmkvbhd")  # trim.rs:486-530


def cut_records(text, line_ends, id_len, desc_start, plan, cfg):
    """The records of a plan (Trimmer.plan_ingested) cut out of the block's own FASTQ text on the host, laid out as bb_trim_batch
    lays them out (trim.rs:447-460) — what `barbell-amd kit` does in its writer threads so that neither the qualities' copy nor the
    rendered records cross PCIe.  line_ends: fastq.fetch_lines (4 per record); id_len / desc_start: fastq.fetch."""
    out = bytearray(plan.text_len)
    text = memoryview(text)

    def line(k, j):
        a = int(line_ends[4 * k + j - 1]) + 1 if (k or j) else 0
        b = int(line_ends[4 * k + j])
        if b > a and text[b - 1] == 13:
            b -= 1
        return a, b

    for s in plan.slices:
        k = int(s["read_idx"])
        hs, he = line(k, 0)
        ss, se = line(k, 1)
        qs, qe = line(k, 3)
        hdr = bytes(text[hs + 1:he])
        rec = b"@" + hdr[: int(id_len[k])] + (b"_%d" % int(s["suffix"]) if s["suffix"] else b"")
        if cfg.write_full_header and len(hdr) > int(desc_start[k]):
            rec += b" " + hdr[int(desc_start[k]):]
        a, b = (0, se - ss) if cfg.skip_trim else (int(s["start"]), int(s["end"]))
        seq, qual = bytes(text[ss + a:ss + b]), bytes(text[qs + a:qs + b])
        if s["flip"]:
            seq, qual = seq.translate(_COMP)[::-1], qual[::-1]
        rec += b"\n" + seq + b"\n+\n" + qual + b"\n"
        assert len(rec) == int(s["rec_len"])
        o = int(s["out_off"])
        out[o:o + len(rec)] = rec
    return bytes(out)
