"""Host-side mirror of the reference's annotate interface for the hot path.

`Demuxer` has the reference's constructor and `add_query_group` (src/annotate/searcher.rs:202-226);
instead of `demux(read_id, read) -> Vec<BarbellMatch>` per read (searcher.rs:430) it offers
`demux_batch(reads)`: one C-ABI call per batch, rows in input order and bit-identical to the CPU restatement's
(the test checker under oracle/), which restates the per-read loop of `DemuxProcessor::process_record`
(src/annotate/annotator.rs:123-135).  Identity with real Barbell is NOT established beyond the reference's own
known-answer tests — its search/score crates are absent from the build image (oracle/README.md, hazards H1-H8);
tools/ref_diff.py checks it in one command wherever a `barbell` binary exists.  `annotate()` mirrors `annotate_with_groups`/`annotate` (annotator.rs:207-285): FASTQ in,
annotation.tsv out, with the byte-exact schema of `BarbellMatch`'s serde layout (searcher.rs:31-64).
All computation happens in the HIP library; there is no CPU path here.
"""
import ctypes as C
import gzip
import re

import numpy as np

from . import _abi, kits
from ._lib import lib

TSV_HEADER = ("read_id\tread_len\trel_dist_to_end\tread_start_bar\tread_end_bar\tread_start_flank\t"
              "read_end_flank\tbar_start\tbar_end\tmatch_type\tflank_cost\tbarcode_cost\tlabel\tstrand\tcuts")


class BarbellError(RuntimeError):
    def __init__(self, code, detail=""):
        self.code = code
        msg = lib().bb_strerror(code).decode()
        super().__init__(f"barbell_amd error {code}: {msg}" + (f" ({detail})" if detail else ""))


class DevBuf:
    """a device allocation through the C-ABI (bb_dev_malloc / bb_dev_free); grows, never shrinks"""

    def __init__(self, dm):
        self.dm, self.ptr, self.cap = dm, None, 0

    def ensure(self, nbytes):
        if self.ptr is not None and nbytes <= self.cap:
            return self.ptr
        self.release()
        p = C.c_void_p()
        want = int(nbytes) + int(nbytes) // 4 + 256
        self.dm._check(lib().bb_dev_malloc(self.dm._ctx(), want, C.byref(p)))
        self.ptr, self.cap = p.value, want
        return self.ptr

    def download(self, arr, nbytes=None):
        nbytes = arr.nbytes if nbytes is None else nbytes
        if nbytes:
            self.dm._check(lib().bb_dev_download(self.dm._ctx(), arr.ctypes.data, self.ptr, nbytes))
        return arr

    def release(self):
        if self.ptr is not None and self.dm._h is not None:
            lib().bb_dev_free(self.dm._h, self.ptr)
        self.ptr, self.cap = None, 0


def build_trace_classes():
    """indices (barbell_amd/csrc/bb_prio.h order) of the traceback-order classes this build holds fast barcode kernels for"""
    m = lib().bb_build_trace_classes()
    return [i for i in range(18) if (m >> i) & 1]


class Demuxer:
    """Demuxer::new(alpha, verbose, min_score_frac, min_score_diff_frac) (searcher.rs:202)."""

    def __init__(self, alpha=0.4, verbose=False, min_score_frac=0.2, min_score_diff_frac=0.1, device=0, policy=None):
        # policy: text form or _abi.Policy (include/barbell_amd_policy.h) — what the un-vendored crates are assumed to do
        # where Barbell's own code does not pin it; None = the default (or $BARBELL_AMD_POLICY)
        self.policy = None if policy is None else _abi.policy_from_str(policy)
        self.alpha, self.verbose = float(alpha), bool(verbose)
        self.min_score_frac, self.min_score_diff_frac = float(min_score_frac), float(min_score_diff_frac)
        self.device = int(device)
        self.queries = []
        self._h = None
        self._bufs = {}

    def add_query_group(self, group):  # searcher.rs:220-226
        if self._h is not None:
            raise RuntimeError("add_query_group after the first demux call")
        self.queries.append(group)
        return self

    # -- context -------------------------------------------------------------------------------
    def _ctx(self):
        if self._h is None:
            L = lib()
            arr, keep = _abi.make_group_descs([g.as_tuple() for g in self.queries])
            p = _abi.Params(self.alpha, self.min_score_frac, self.min_score_diff_frac, self.device)
            h = C.c_void_p()
            if self.policy is None:
                rc = L.bb_create(arr, len(self.queries), C.byref(p), C.byref(h))
            else:
                rc = L.bb_create_policy(arr, len(self.queries), C.byref(p), C.byref(self.policy), C.byref(h))
            if rc != 0:
                raise BarbellError(rc, L.bb_last_error(None).decode())
            self._h = h
        return self._h

    def note(self):
        """what bb_last_error(ctx) holds: after a successful create, the note that the policy's traceback order has no fast kernels in this build"""
        return lib().bb_last_error(self._ctx()).decode()

    def close(self):
        if self._h is not None:
            for b in self._bufs.values():
                b.release()
            self._bufs = {}
            lib().bb_destroy(self._h)
            self._h = None

    def buf(self, name):
        """named device buffer owned by this demuxer (rows, verdicts, text, ...)"""
        if name not in self._bufs:
            self._bufs[name] = DevBuf(self)
        return self._bufs[name]

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc != 0:
            raise BarbellError(rc, lib().bb_last_error(self._h).decode() if self._h else "")

    # -- geometry ------------------------------------------------------------------------------
    def group_info(self, g):
        i = _abi.GroupInfo()
        self._check(lib().bb_group_get_info(self._ctx(), g, C.byref(i)))
        return i

    def flank(self, g):
        buf = C.create_string_buffer(self.group_info(g).flank_len)
        self._check(lib().bb_group_get_flank(self._ctx(), g, buf))
        return buf.raw

    def pattern(self, g, idx, rc=False):
        buf = C.create_string_buffer(self.group_info(g).pattern_len)
        self._check(lib().bb_group_get_pattern(self._ctx(), g, idx, int(rc), buf))
        return buf.raw

    # -- the hot path --------------------------------------------------------------------------
    def demux_packed(self, bases, offsets):
        """bases uint8[total], offsets uint64[n+1] (host arrays) -> rows (ROW_DTYPE)."""
        bases = np.ascontiguousarray(bases, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        n = len(offsets) - 1
        cap = max(64, 4 * n)
        while True:
            rows = np.zeros(cap, dtype=_abi.ROW_DTYPE)
            nr = C.c_uint64()
            rc = lib().bb_annotate_batch(self._ctx(), bases.ctypes.data, offsets.ctypes.data, n, rows.ctypes.data, cap, C.byref(nr))
            if rc == _abi.BB_E_CAPACITY:
                cap = int(nr.value)
                continue
            self._check(rc)
            return rows[: nr.value]

    def demux_nibbles(self, bases, offsets):
        """bb_annotate_batch_packed: the same batch with every read packed two bases per byte by bb_pack_bases (half the PCIe bytes; the
        kernels only look at a character's IUPAC base set, so the rows are demux_packed's)"""
        bases = np.ascontiguousarray(bases, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        n = len(offsets) - 1
        lens = (offsets[1:] - offsets[:-1]).astype(np.uint64)
        poff = np.zeros(n + 1, dtype=np.uint64)
        poff[1:] = np.cumsum((lens + np.uint64(1)) // np.uint64(2), dtype=np.uint64)
        packed = np.zeros(int(poff[n]) + 16, dtype=np.uint8)
        L = lib()
        for i in range(n):
            wrote = L.bb_pack_bases(bases.ctypes.data + int(offsets[i]), int(lens[i]), packed.ctypes.data + int(poff[i]))
            assert wrote == (int(lens[i]) + 1) // 2
        rel = offsets - offsets[0]
        cap = 4 * n + 64
        while True:
            rows = np.zeros(cap, dtype=_abi.ROW_DTYPE)
            got = C.c_uint64(0)
            rc = L.bb_annotate_batch_packed(self._ctx(), packed.ctypes.data, poff.ctypes.data, rel.ctypes.data, n, rows.ctypes.data, cap, C.byref(got))
            if rc == _abi.BB_E_CAPACITY:
                cap = int(got.value)
                continue
            self._check(rc)
            return rows[: got.value].copy()

    def demux_batch(self, reads):
        return self.demux_packed(*_abi.pack_reads(reads))

    def demux_dev(self, d_bases, d_offsets, n_reads, d_rows, rows_cap):
        """Device-pointer variant (ints = HIP device pointers). Returns the number of rows."""
        nr = C.c_uint64()
        rc = lib().bb_annotate_batch_dev(self._ctx(), d_bases, d_offsets, n_reads, d_rows, rows_cap, C.byref(nr))
        if rc == _abi.BB_E_CAPACITY:
            raise BarbellError(rc, f"need {nr.value} rows")
        self._check(rc)
        return int(nr.value)

    def demux_ingested(self, batch, n_reads):
        """annotate the batch bb_fastq_ingest left in HBM; rows stay in the "rows" device buffer and a host
        copy is returned"""
        cap = max(64, 4 * n_reads)
        while True:
            d = self.buf("rows").ensure(cap * 48)
            nr = C.c_uint64()
            rc = lib().bb_annotate_batch_dev(self._ctx(), batch.d_bases, batch.d_offsets, n_reads, d, cap, C.byref(nr))
            if rc == _abi.BB_E_CAPACITY:
                cap = int(nr.value)
                continue
            self._check(rc)
            return self.buf("rows").download(np.zeros(int(nr.value), dtype=_abi.ROW_DTYPE))

    # -- histogram / timing --------------------------------------------------------------------
    def counts(self):
        n = lib().bb_counts_len(self._ctx())
        out = np.zeros(n, dtype=np.uint64)
        self._check(lib().bb_counts(self._ctx(), out.ctypes.data))
        return out

    def counts_dev_ptr(self):
        return lib().bb_counts_dev(self._ctx()), lib().bb_counts_len(self._ctx())

    def counts_reset(self):
        self._check(lib().bb_counts_reset(self._ctx()))

    def set_timing(self, on=True):
        lib().bb_set_timing(self._ctx(), int(on))

    def kernel_ms(self):
        L = lib()
        return {L.bb_kernel_name(k).decode(): L.bb_last_kernel_ms(self._ctx(), k) for k in range(L.bb_n_kernels())}

    def dominant_kernel(self):
        """with timing on: (name as rocprofv3 prints it, ms) of the longest single kernel launch of the last batch's barcode stage
        (bb_last_dominant_kernel); ("", 0.0) if there was none"""
        buf, ms = C.create_string_buffer(96), C.c_float()
        self._check(lib().bb_last_dominant_kernel(self._ctx(), buf, 96, C.byref(ms)))
        return buf.value.decode(), float(ms.value)

    def scan_stats(self, g=0):
        """how the flank scan of group g ran on the last batch: {flagged_pieces, total_pieces, kind}; kind 0 = full scan, 1 = filter +
        verification, 2 = the filter flagged too much of the batch and the full scan took over, 3 = full scan without a filter pass while the group is
        backed off after a batch of kind 2 (bb_last_scan_stats)"""
        f, t, k = C.c_uint64(), C.c_uint64(), C.c_int()
        self._check(lib().bb_last_scan_stats(self._ctx(), g, C.byref(f), C.byref(t), C.byref(k)))
        return {"flagged_pieces": f.value, "total_pieces": t.value, "kind": k.value}

    def filter_twin(self, g=0):
        """(the group whose filter pass says it all for group g — its window is g's, reverse-complemented or as it is — or -1; how g's scan of the
        last batch used it: 0 not, 1 that group's flags with the strands swapped, 2 as they are): bb_filter_twin"""
        a, sh = C.c_int(), C.c_int()
        self._check(lib().bb_filter_twin(self._ctx(), g, C.byref(a), C.byref(sh)))
        return a.value, sh.value

    def host_syncs(self):
        """how often the host waited for the device inside the last batch call (bb_last_host_syncs)"""
        return int(lib().bb_last_host_syncs(self._ctx()))

    def length_stats(self):
        """the last batch's read lengths as the scans saw them: {min_lines, max_lines, work_items} (128-byte lines; work items = reads, or
        segments where the lengths differ: bb_last_length_stats)"""
        a, b, w = C.c_uint32(), C.c_uint32(), C.c_uint32()
        self._check(lib().bb_last_length_stats(self._ctx(), C.byref(a), C.byref(b), C.byref(w)))
        return {"min_lines": a.value, "max_lines": b.value, "work_items": w.value}

    def barcode_stats(self, g=0, strand=0):
        """the barcode stage of the last batch for (group, strand): {hits, undecided, lane_kernel} — flank hits listed, how many the fast
        kernel's bounds left to the exact pass, and whether the pair's next batch takes the one-lane-per-hit kernel (bb_last_barcode_stats)"""
        h, u, k = C.c_uint64(), C.c_uint64(), C.c_int()
        self._check(lib().bb_last_barcode_stats(self._ctx(), g, strand, C.byref(h), C.byref(u), C.byref(k)))
        return {"hits": h.value, "undecided": u.value, "lane_kernel": bool(k.value)}

    # -- synthetic reads -----------------------------------------------------------------------
    def synth_dev(self, seed, len_min, len_max, first_read, n, d_offsets, d_bases):
        self._check(lib().bb_synth_reads_dev(self._ctx(), seed, len_min, len_max, first_read, n, d_offsets, d_bases))


def synth_offsets(seed, len_min, len_max, first_read, n):
    off = np.zeros(n + 1, dtype=np.uint64)
    rc = lib().bb_synth_offsets(seed, len_min, len_max, first_read, n, off.ctypes.data)
    if rc != 0:
        raise BarbellError(rc)
    return off


def synth_reads_host(groups, seed, len_min, len_max, first_read, n):
    """groups: list[kits.QueryGroup].  Returns (bases, offsets) generated on the host (no GPU)."""
    off = synth_offsets(seed, len_min, len_max, first_read, n)
    bases = np.zeros(int(off[-1]), dtype=np.uint8)
    arr, keep = _abi.make_group_descs([g.as_tuple() for g in groups])
    rc = lib().bb_synth_reads_host(arr, len(groups), seed, len_min, len_max, first_read, n, off.ctypes.data, bases.ctypes.data)
    if rc != 0:
        raise BarbellError(rc)
    return bases, off


# ---- TSV (searcher.rs:31-142, annotator.rs:13-26,246-251) ------------------------------------------
def _csv_field(x):
    """csv crate, QuoteStyle::Necessary (annotator.rs:246-251): a field holding the delimiter, a quote, CR or LF is quoted, quotes doubled"""
    return '"' + x.replace('"', '""') + '"' if any(ch in x for ch in '\t"\n\r') else x


def format_rows(rows, read_ids, groups, verdicts=None):
    """rows -> list of TSV lines (no header), csv-crate style: tab-delimited, no quoting needed for
    these field types unless a read id contains a tab/quote/newline.  With `verdicts` (filter step)
    the `cuts` column is filled like the reference's filtered.tsv (searcher.rs:91-106)."""
    from .filter import format_cuts

    out = []
    for i, r in enumerate(rows):
        g = groups[int(r["group_idx"])]
        label = "flank" if r["barcode_idx"] < 0 else g.labels[int(r["barcode_idx"])]
        rid = _csv_field(read_ids[int(r["read_idx"])])
        label = _csv_field(label)
        out.append("\t".join((
            rid, str(int(r["read_len"])), str(int(r["rel_dist_to_end"])),
            str(int(r["read_start_bar"])), str(int(r["read_end_bar"])),
            str(int(r["read_start_flank"])), str(int(r["read_end_flank"])),
            str(int(r["bar_start"])), str(int(r["bar_end"])),
            _abi.MATCH_TYPE_STR[int(r["match_type"])], str(int(r["flank_cost"])), str(int(r["barcode_cost"])),
            label, _abi.STRAND_STR[int(r["strand"])], "" if verdicts is None else format_cuts(verdicts[i]),
        )))
    return out


_WS = re.compile("[\\t\\n\\x0b\\x0c\\r \\x85\\xa0\\u1680\\u2000-\\u200a\\u2028\\u2029\\u202f\\u205f\\u3000]")  # char::is_whitespace


def split_fastq_header(header):  # src/io/io.rs:6-17: id up to the first whitespace, description left-trimmed
    m = _WS.search(header)
    if not m:
        return header, ""
    j = m.start()
    while j < len(header) and _WS.match(header[j]):
        j += 1
    return header[: m.start()], header[j:]


def read_fastq_records(path):
    """Yields (header line without '@', seq, qual) as bytes.  Plain or gzip FASTQ, 4-line records."""
    from .fastq import is_gzip

    op = gzip.open if is_gzip(path) else open
    with op(path, "rb") as f:
        while True:
            h = f.readline()
            if not h:
                return
            s = f.readline().rstrip(b"\r\n")
            f.readline()
            q = f.readline().rstrip(b"\r\n")
            yield h[1:].rstrip(b"\r\n"), s, q


def read_fastq(path):
    """Yields (read_id, seq bytes)."""
    for h, s, _ in read_fastq_records(path):
        yield split_fastq_header(h.decode())[0], s


def annotate(read_files, out_file, query_groups, alpha=0.4, min_score=0.2, min_score_diff=0.1, max_flank_errors=None,
             batch_reads=0, block_bytes=512 << 20, device=0, filter_patterns=None, filtered_file=None, dropped_file=None, trim_folder=None,
             trim_config=None, inspector=None, policy=None, stats=None):
    """annotate_with_groups + annotate (annotator.rs:207-285): sets the flank threshold of each group
    (explicit --flank-max-errors or the automatic cutoff), hands the FASTQ text to the GPU block by block
    (`block_bytes`, or `batch_reads` * 4096; records are parsed there, barbell_amd/fastq.py) and writes annotation.tsv.  With `filter_patterns` the filter step (filter.rs:10-119) runs on
    the rows of every batch while they are in HBM and `filtered_file` / `dropped_file` get the rows
    of passing / failing reads with their `cuts` column — what `barbell filter` would write from the
    annotation file.  With `trim_folder` (needs the filter) the trim step (trim.rs:317-480) runs on the same
    batch as well: the GPU cuts the passing reads and renders the FASTQ records grouped by output label, the
    host appends each group to '{trim_folder}/{label}.trimmed.fastq[.gz]'.  `inspector(dm)` may build an
    inspect_rows.Inspector that is fed the rows of every batch (inspect.rs:119-208 without re-reading the TSV).
    Returns (total_reads, reads_with_rows); a dict passed as `stats` also gets the counters of the fused steps (reads kept / dropped by the
    filter, trimmed / split / failed: what the reference's progress logs hold, progress.rs:15-88)."""
    for g in query_groups:
        if max_flank_errors is not None:
            g.set_flank_threshold(max_flank_errors)
    dm = Demuxer(alpha, False, min_score, min_score_diff, device, policy=policy)
    for g in query_groups:
        dm.add_query_group(g)
    flt = None
    if filter_patterns is not None:
        from .filter import Filter

        flt = Filter(dm, filter_patterns)
    trimmer = writers = None
    if trim_folder is not None:
        if flt is None:
            raise ValueError("the trim step needs filter patterns (cuts come from the filter)")
        from .trim import LabelWriters, TrimConfig, Trimmer

        trim_config = trim_config or TrimConfig()
        trimmer = Trimmer(dm, trim_config)
        writers = LabelWriters(trim_folder, trim_config, trimmer.tables)
    insp = inspector(dm) if inspector is not None else None
    total = found = 0
    outs = {"anno": open(out_file, "wb"),
            "kept": open(filtered_file, "wb") if (flt and filtered_file) else None,
            "dropped": open(dropped_file, "wb") if (flt and dropped_file) else None}
    wrote = {k: False for k in outs}

    from . import fastq as Q

    from .format import FMT_DROPPED, FMT_KEPT, RowFormatter

    fmt = RowFormatter(dm, query_groups)
    need_ids = insp is not None or trimmer is not None

    def emit_bytes(key, text):
        f = outs[key]
        if f is None or not text:
            return
        if not wrote[key]:  # csv writer emits the header with the first record only
            f.write((TSV_HEADER + "\n").encode())
            wrote[key] = True
        f.write(text)

    def process(info, batch):
        nonlocal total, found
        n = int(info.n_records)
        rows = dm.demux_ingested(batch, n)
        total += n
        found += len(np.unique(rows["read_idx"]))
        d_rows = dm.buf("rows").ptr
        emit_bytes("anno", fmt.render(d_rows, len(rows), batch)[0])  # the TSV lines are rendered on the GPU
        ids = Q.read_ids(Q.fetch(dm, info)) if need_ids else None
        if insp is not None:
            insp.add(rows, ids, d_rows=d_rows)
        if flt is not None:
            ver = flt.verdicts_ingested(d_rows, len(rows), download=trimmer is not None or stats is not None)
            if stats is not None and len(rows):
                first = np.r_[True, rows["read_idx"][1:] != rows["read_idx"][:-1]]
                stats["kept"] = stats.get("kept", 0) + int((first & (ver["pass"] != 0)).sum())
                stats["dropped"] = stats.get("dropped", 0) + int((first & (ver["pass"] == 0)).sum())
            d_v = dm.buf("verdicts").ptr
            emit_bytes("kept", fmt.render(d_rows, len(rows), batch, FMT_KEPT, d_v)[0])
            emit_bytes("dropped", fmt.render(d_rows, len(rows), batch, FMT_DROPPED, d_v)[0])
            if trimmer is not None:
                writers.write(trimmer.trim_ingested(d_rows, d_v, len(rows), batch, info), ids)

    try:
        for info, batch in Q.batches(dm, read_files, batch_reads * 4096 if batch_reads else block_bytes):
            process(info, batch)
    finally:
        for f in outs.values():
            if f is not None:
                f.close()
        if writers is not None:
            writers.close()
        if stats is not None:
            stats.update(total=total, found=found)
            if writers is not None:
                stats.update(trimmed=writers.n_trimmed, split=writers.n_split, failed=writers.n_failed)
        if insp is not None:
            insp.close()
    dm.close()
    return total, found


def annotate_with_kit(read_files, out_file, kit, use_extended=False, **kw):  # annotator.rs:196-204
    return annotate(read_files, out_file, kits.groups_from_kit(kit, use_extended), **kw)
