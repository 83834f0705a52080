"""The reference's stand-alone steps on FILES: `barbell filter`, `barbell inspect`, `barbell trim` take an annotation.tsv (and the
FASTQ) written earlier — by the reference or by this package — instead of rows that are still in HBM.

The fused path (annotate.annotate with filter / trim / inspector, `barbell-amd kit`) stays the fast one: every batch goes to
the GPU once.  These entry points exist so that the reference's documented four-command workflow (README: annotate -> inspect
-> filter -> trim, with files in between; bin/main.rs:274-470) keeps working on the same files: the TSV is parsed back into
the C-ABI's `bb_row`s (searcher.rs:31-142 is the schema), the same HIP kernels decide (k_filter, k_inspect, k_trim_plan /
k_trim_render through bb_filter_rows, bb_inspect_rows, bb_trim_batch) and the same writers write.

Labels: a row's `label` column is mapped to a (group, barcode) slot.  With the run's query groups at hand (`groups=`: the kit or
the FASTA files of the annotate run) the slots are the run's own; without them (the reference's filter / trim / inspect need
no queries either) `LabelSpace.from_labels` builds stand-in groups that only carry the file's label strings — the kernels of
these steps read a row's slot and its label id, never the query sequences."""
import csv
import io
import os
import re

import numpy as np

from . import _abi
from .annotate import TSV_HEADER, Demuxer, _csv_field, read_fastq_records, split_fastq_header
from .filter import MAX_CUTS, VERDICT_DTYPE, Filter, format_cuts, patterns_from_files
from .inspect_rows import Inspector
from .kits import QueryGroup
from .trim import LabelWriters, TrimConfig, Trimmer

COLUMNS = TSV_HEADER.split("\t")
_TAG_OF = {"Ftag": _abi.BB_FTAG, "Rtag": _abi.BB_RTAG, "Fflank": _abi.BB_FTAG, "Rflank": _abi.BB_RTAG}
_MT = {s: i for i, s in enumerate(_abi.MATCH_TYPE_STR)}
_STRAND = {s: i for i, s in enumerate(_abi.STRAND_STR)}
GROUP_MAX_LABELS = 1022   # 1024 sequences per group (include/barbell_amd.h) minus the two stand-ins that pin the barcode region
_DUMMY = ("\x00barbell-amd stand-in A", "\x00barbell-amd stand-in T")


class TsvError(ValueError):
    pass


# ---- the `cuts` column (searcher.rs:91-140, pattern.rs:40-95) -----------------------------------------------------------
def parse_cuts(s):
    """"After(0):1,Before(2):1" -> ([(direction 1 = After, group_id), ...], position).  The position is the row's index among
    its read's rows (filter.rs:183-214 writes the element index); the verdict record holds one per row."""
    if not s:
        return [], None
    cuts, pos = [], None
    for part in s.split(","):
        cut, sep, p = part.partition(":")
        if not sep:
            raise TsvError("Invalid cut format: missing position part")
        if cut.startswith("Before(") and cut.endswith(")"):
            d, gid = 0, cut[7:-1]
        elif cut.startswith("After(") and cut.endswith(")"):
            d, gid = 1, cut[6:-1]
        else:
            raise TsvError(f"Invalid cut string: {cut}")
        if not gid.isdigit() or not p.isdigit():
            raise TsvError(f"Invalid cut string: {part}")
        if pos is not None and int(p) != pos:
            raise TsvError(f"cuts of one row with different positions ({s}): not what the filter step writes")
        pos = int(p)
        cuts.append((d, int(gid)))
    if len(cuts) > MAX_CUTS:
        raise TsvError(f"more than {MAX_CUTS} cuts on one row (kernel limit, include/barbell_amd_filter.h)")
    return cuts, pos


# ---- labels <-> slots ----------------------------------------------------------------------------------------------------
def _standin_seq(i):
    """A 46-nt sequence of SQK-NBD114-96's shape (14 / 24 / 8) whose 24-nt barcode encodes i: never searched, only there so
    that bb_create has a well-formed group to hang the labels on."""
    digits = "".join("ACGT"[(i >> (2 * j)) & 3] for j in range(22))
    return b"AAGGTTAACACAAA" + ("G" + digits + "C").encode() + b"CAGCACCT"


class LabelSpace:
    """(tag kind, label) -> (group_idx, barcode_idx) and back, over real query groups or stand-in ones"""

    def __init__(self, groups, standin=False):
        self.groups, self.standin = list(groups), standin
        self.slot, self.flank_group = {}, {}
        for gi, g in enumerate(self.groups):
            kind = _abi.BB_RTAG if g.match_type == _abi.BB_RTAG else _abi.BB_FTAG
            self.flank_group.setdefault(kind, gi)
            for bi, lab in enumerate(g.labels):
                self.slot.setdefault((kind, lab), (gi, bi))

    @staticmethod
    def from_labels(pairs):
        """pairs: iterable of (match_type string, label) as they stand in the file"""
        by_kind = {_abi.BB_FTAG: [], _abi.BB_RTAG: []}
        seen = set()
        for mt, lab in pairs:
            kind = _TAG_OF[mt]
            if mt in ("Fflank", "Rflank"):
                seen.add((kind, None))
            elif (kind, lab) not in seen:
                seen.add((kind, lab))
                by_kind[kind].append(lab)
        groups = []
        for kind in (_abi.BB_FTAG, _abi.BB_RTAG):
            labs = by_kind[kind]
            if not labs and (kind, None) not in seen:
                continue
            for a in range(0, max(1, len(labs)), GROUP_MAX_LABELS):
                part = labs[a:a + GROUP_MAX_LABELS]
                seqs = [_standin_seq(i) for i in range(len(part))] + [b"AAGGTTAACACAAA" + b"A" * 24 + b"CAGCACCT", b"AAGGTTAACACAAA" + b"T" * 24 + b"CAGCACCT"]
                groups.append(QueryGroup(seqs, part + list(_DUMMY), kind, flank_k=0))
        if len(groups) > 32:
            raise TsvError(f"{sum(len(v) for v in by_kind.values())} distinct labels need {len(groups)} stand-in groups; a context holds 32 "
                           "(include/barbell_amd.h) — pass the run's query groups instead")
        return LabelSpace(groups, standin=True)

    def lookup(self, match_type, label):
        kind = _TAG_OF[match_type]
        if match_type in ("Fflank", "Rflank"):
            gi = self.flank_group.get(kind)
            if gi is None:
                raise TsvError(f"a {match_type} row, but no query group of that tag type")
            return gi, -1
        s = self.slot.get((kind, label))
        if s is None:
            raise TsvError(f"label {label!r} ({match_type}) is not among the query groups' labels")
        return s


# ---- annotation.tsv -> rows -----------------------------------------------------------------------------------------------
class TsvBatch:
    """rows of some consecutive reads of the file: `rows` (ROW_DTYPE, read_idx = index into read_ids), `verdicts` from the
    cuts column (pass = 1: a filtered file holds passing reads only), the raw fields for writing a row back out unchanged"""

    def __init__(self, rows, verdicts, read_ids, has_cuts, fields=None, columns=None):
        self.rows, self.verdicts, self.read_ids, self.has_cuts, self.fields, self.columns = rows, verdicts, read_ids, has_cuts, fields, columns

    def lines(self, verdicts):
        """the records as TSV lines in the schema's column order — the fields as they stand in the file (the csv writer's own output is
        canonical: re-serialising it gives it back), a field that needs quoting quoted again — with the cuts the rows CAME with followed by
        the cuts of `verdicts` (filter.rs:204-209: `existing_cuts.push`; a filtered.tsv filtered again keeps what `trim` needs)"""
        order = [self.columns[name] for name in COLUMNS[:-1]]
        cut_col = self.columns[COLUMNS[-1]]
        cut_rows = np.nonzero(verdicts["n_cuts"])[0]
        cuts = dict(zip(cut_rows.tolist(), (format_cuts(verdicts[i]) for i in cut_rows)))
        too_many = np.nonzero(self.verdicts["n_cuts"].astype(np.int64) + verdicts["n_cuts"] > 3)[0]
        if len(too_many):
            raise TsvError(f"read {self.read_ids[int(self.rows['read_idx'][too_many[0]])]!r}: more than 3 cuts on one row after this filter "
                           "(kernel limit, include/barbell_amd_filter.h)")
        out = []
        for i, rec in enumerate(self.fields):
            f = [rec[j] for j in order]
            if any(ch in f[0] for ch in '\t"\n\r') or any(ch in f[12] for ch in '\t"\n\r'):
                f[0], f[12] = _csv_field(f[0]), _csv_field(f[12])
            old, new = (rec[cut_col] if self.verdicts["n_cuts"][i] else ""), cuts.get(i, "")
            f.append(old + "," + new if old and new else old or new)
            out.append("\t".join(f))
        return out


def _records(path):
    """-> (column index of every schema field, iterator over the records' field lists).  Lines without a quote character are split at
    the tabs (what the csv crate wrote for every ordinary read id); the others — a field holding a tab, a quote, CR or LF is quoted,
    quotes doubled, annotator.rs:246-251 — go through the csv module, together with the lines a quoted line break continues on."""
    f = open(path, newline="", encoding="utf-8")
    head = f.readline()
    if not head:
        f.close()
        return None, iter(())   # an empty file: the csv writer emits the header with the first record only
    header = next(csv.reader([head], delimiter="\t"))
    missing = [c for c in COLUMNS if c not in header]
    if missing:
        f.close()
        raise TsvError(f"{path}: columns missing from the header: {missing}")
    ncol = len(header)

    def it():
        with f:
            for line in f:
                if '"' not in line:
                    rec = line.rstrip("\r\n").split("\t")
                else:
                    while line.count('"') % 2:   # a quoted field with a line break in it
                        more = f.readline()
                        if not more:
                            raise TsvError(f"{path}: unterminated quoted field")
                        line += more
                    rec = next(csv.reader(io.StringIO(line, newline=""), delimiter="\t", quotechar='"', doublequote=True, strict=True))
                if len(rec) != ncol:
                    if rec in ([], [""]):
                        continue
                    raise TsvError(f"{path}: a record of {len(rec)} fields under a header of {ncol}")
                yield rec

    return [header.index(c) for c in COLUMNS], it()


def scan_labels(path):
    """the distinct (match_type, label) pairs of a file, in order of first appearance"""
    col, recs = _records(path)
    if col is None:
        return []
    i_mt, i_lab = col[COLUMNS.index("match_type")], col[COLUMNS.index("label")]
    seen = dict.fromkeys((rec[i_mt], rec[i_lab]) for rec in recs)
    for mt, _ in seen:
        if mt not in _MT:
            raise TsvError(f"{path}: match_type {mt!r}")
    return list(seen)


_INT_FIELDS = ("read_len", "rel_dist_to_end", "read_start_bar", "read_end_bar", "read_start_flank", "read_end_flank", "bar_start", "bar_end",
               "flank_cost", "barcode_cost")


_INTS_RE = re.compile(r"(?:-?[0-9]+)?(?:\n-?[0-9]+)*")


def _ints(strings, what, path):
    """decimal fields of many records in one C call"""
    import warnings

    try:
        with warnings.catch_warnings():
            warnings.simplefilter("ignore", DeprecationWarning)   # text that does not parse to its end: caught by the length check below
            a = np.fromstring(" ".join(strings), dtype=np.int64, sep=" ") if strings else np.zeros(0, dtype=np.int64)
    except ValueError:
        a = np.zeros(0, dtype=np.int64)
    # every field is checked, not a sample: "1 2" in one record and "" in another would parse to the right COUNT and shift values onto other rows
    if len(a) != len(strings) or not _INTS_RE.fullmatch("\n".join(strings)):
        bad = next((s for s in strings if not re.fullmatch(r"-?[0-9]+", s)), "?")
        raise TsvError(f"{path}: {what}: {bad!r} is not an integer")
    return a


def read_annotation_tsv(path, space, batch_rows=1 << 18, group_consecutive=True):
    """Yields TsvBatch objects.  Rows of one read (consecutive lines with the same read_id: filter.rs:52-85) are never split
    over two batches.  With group_consecutive=False every distinct read_id of the whole file is one read however its lines lie
    (trim.rs:337-358 collects them in a map) — the caller then gets ONE batch."""
    col, recs = _records(path)
    if col is None:
        return
    c = {name: col[i] for i, name in enumerate(COLUMNS)}
    c_id, c_mt, c_lab, c_strand, c_cuts = c["read_id"], c["match_type"], c["label"], c["strand"], c["cuts"]
    slot_cache = {}

    def slot_of(key):
        s = slot_cache.get(key)
        if s is None:
            if key[0] not in _MT:
                raise TsvError(f"{path}: match_type {key[0]!r}")
            s = slot_cache[key] = space.lookup(*key) + (_MT[key[0]],)
        return s

    def build(pend, id_col, ridx, read_ids):
        n = len(pend)
        rows = np.zeros(n, dtype=_abi.ROW_DTYPE)
        ver = np.zeros(n, dtype=VERDICT_DTYPE)
        for name in _INT_FIELDS:
            ci = c[name]
            v = _ints([r[ci] for r in pend], name, path)
            info = np.iinfo(rows.dtype[name])
            if len(v) and (v.min() < info.min or v.max() > info.max):
                raise TsvError(f"{path}: {name} outside the range of the row record")
            rows[name] = v
        slots = np.array([slot_of((r[c_mt], r[c_lab])) for r in pend], dtype=np.int64).reshape(n, 3)
        rows["group_idx"], rows["barcode_idx"], rows["match_type"] = slots[:, 0], slots[:, 1], slots[:, 2]
        try:
            rows["strand"] = [_STRAND[r[c_strand]] for r in pend]
        except KeyError as e:
            raise TsvError(f"{path}: Invalid strand: {e.args[0]}") from None
        rows["read_idx"] = ridx
        ver["pass"] = 1
        with_cuts = [i for i, r in enumerate(pend) if r[c_cuts]]
        for i in with_cuts:
            cuts, pos = parse_cuts(pend[i][c_cuts])
            v = ver[i]
            v["n_cuts"], v["match_idx"] = len(cuts), pos
            for q, (d, gid) in enumerate(cuts):
                if gid > 0xFFFF:
                    raise TsvError("cut group id above 65535 (kernel limit, include/barbell_amd_filter.h)")
                v["cuts"][q]["direction"], v["cuts"][q]["group_id"] = d, gid
        fields = pend
        if not group_consecutive:   # rows of a read together, reads in order of first appearance, a read's rows in file order
            order = np.argsort(rows["read_idx"], kind="stable")
            rows, ver = rows[order], ver[order]
            fields = [pend[i] for i in order]
        # the position of a row without cuts: its index among its read's rows (what k_filter writes for every row)
        if n:
            first = np.r_[True, rows["read_idx"][1:] != rows["read_idx"][:-1]]
            start = np.maximum.accumulate(np.where(first, np.arange(n), 0))
            nocut = ver["n_cuts"] == 0
            ver["match_idx"][nocut] = (np.arange(n) - start).astype(np.uint16)[nocut]
        return TsvBatch(rows, ver, read_ids, bool(with_cuts), fields, c)

    if not group_consecutive:
        pend = list(recs)
        if pend:
            id_of = {}
            ridx = np.array([id_of.setdefault(r[c_id], len(id_of)) for r in pend], dtype=np.uint32)
            yield build(pend, None, ridx, list(id_of))
        return
    pend, last = [], None
    for rec in recs:
        if len(pend) >= batch_rows and rec[c_id] != last:
            yield _consecutive(build, pend, c_id)
            pend = []
        pend.append(rec)
        last = rec[c_id]
    if pend:
        yield _consecutive(build, pend, c_id)


def _consecutive(build, pend, c_id):
    ids = [r[c_id] for r in pend]
    first = np.array([True] + [a != b for a, b in zip(ids[1:], ids[:-1])])
    ridx = (np.cumsum(first) - 1).astype(np.uint32)
    return build(pend, ids, ridx, [ids[i] for i in np.nonzero(first)[0]])


def _context(space, device, patterns=()):
    """a library context that carries the label space (and a filter: the trim step takes its label ids from it)"""
    dm = Demuxer(device=device)
    for g in space.groups:
        dm.add_query_group(g)
    return dm, Filter(dm, list(patterns))


def _space_for(path, groups):
    return LabelSpace(groups) if groups is not None else LabelSpace.from_labels(scan_labels(path))


def _format(rows, read_ids, space, verdicts):
    """annotate.format_rows, with the file's own label strings"""
    out = []
    for i, r in enumerate(rows):
        g = space.groups[int(r["group_idx"])]
        label = "flank" if r["barcode_idx"] < 0 else g.labels[int(r["barcode_idx"])]
        out.append("\t".join((
            _csv_field(read_ids[int(r["read_idx"])]), str(int(r["read_len"])), str(int(r["rel_dist_to_end"])),
            str(int(r["read_start_bar"])), str(int(r["read_end_bar"])), str(int(r["read_start_flank"])), str(int(r["read_end_flank"])),
            str(int(r["bar_start"])), str(int(r["bar_end"])), _abi.MATCH_TYPE_STR[int(r["match_type"])], str(int(r["flank_cost"])),
            str(int(r["barcode_cost"])), _csv_field(label), _abi.STRAND_STR[int(r["strand"])], format_cuts(verdicts[i]))))
    return out


def write_progress_log(step, log_dir, counts):
    """ProgressTracker::new_with_logging + ProgressLog::write (progress.rs:96-144, 188-195): '{log_dir}/{step}.{unix ms}.log' holding
    "step\\tmetric\\tcount" and a line per counter — what --verbose leaves behind besides the progress bars (those stay out of scope)"""
    import time

    path = os.path.join(log_dir or ".", f"{step}.{int(time.time() * 1000)}.log")
    with open(path, "w") as f:
        f.write("step\tmetric\tcount\n")
        for metric, n in counts:
            f.write(f"{step}\t{metric}\t{n}\n")
    return path


# ---- barbell filter (filter.rs:10-119) -----------------------------------------------------------------------------------
def filter_file(annotated_file, output_file, patterns, dropped_out_file=None, groups=None, device=0, batch_rows=1 << 18, log=print, verbose=False):
    """Reads of annotation.tsv (consecutive lines of one read_id) against the patterns on the GPU (k_filter); passing reads' rows
    go to `output_file` with their cuts, the others to `dropped_out_file`.  Returns (reads, kept, dropped)."""
    space = _space_for(annotated_file, groups)
    outs = {True: open(output_file, "w", encoding="utf-8", newline=""),
            False: open(dropped_out_file, "w", encoding="utf-8", newline="") if dropped_out_file else None}
    wrote = {True: False, False: False}
    total = kept = 0
    try:
        if space.groups:
            dm, flt = _context(space, device, patterns)
            for b in read_annotation_tsv(annotated_file, space, batch_rows):
                ver = flt.verdicts(b.rows)
                lines = b.lines(ver)
                first = np.r_[True, b.rows["read_idx"][1:] != b.rows["read_idx"][:-1]]
                total += int(first.sum())
                kept += int((first & (ver["pass"] != 0)).sum())
                for ok in (True, False):
                    f = outs[ok]
                    sel = [l for l, p in zip(lines, ver["pass"]) if bool(p) == ok]
                    if f is None or not sel:
                        continue
                    if not wrote[ok]:   # the csv writer emits the header with the first record only
                        f.write(TSV_HEADER + "\n")
                        wrote[ok] = True
                    f.write("\n".join(sel) + "\n")
            dm.close()
    finally:
        for f in outs.values():
            if f is not None:
                f.close()
    log(f"filter: {total} reads, {kept} kept, {total - kept} dropped")
    if verbose:   # filter.rs:18-25: next to the output file
        write_progress_log("filter", os.path.dirname(output_file), [("Total:", total), ("Kept:", kept), ("Dropped:", total - kept)])
    return total, kept, total - kept


# ---- barbell inspect (inspect.rs:119-208) --------------------------------------------------------------------------------
def inspect_file(annotated_file, top_n=10, read_pattern_out=None, bucket_size=250, groups=None, device=0, batch_rows=1 << 18, log=print):
    """-> the Inspector (counts, summary lines already logged)"""
    space = _space_for(annotated_file, groups)
    if not space.groups:
        insp = Inspector(None, read_pattern_out, bucket_size)
    else:
        dm, _ = _context(space, device)
        insp = Inspector(dm, read_pattern_out, bucket_size)
        for b in read_annotation_tsv(annotated_file, space, batch_rows):
            insp.add(b.rows, b.read_ids, b.verdicts if b.has_cuts else None)
        dm.close()
    insp.close()
    for line in insp.summary(top_n):
        log(line)
    return insp


# ---- barbell trim (trim.rs:317-480) ---------------------------------------------------------------------------------------
def trim_file(filtered_match_file, read_fastq_files, output_folder, config=None, groups=None, device=0, batch_reads=20000, log=print):
    """Annotations by read id (the whole filtered file in memory, as the reference holds it), the FASTQ streamed; every batch of
    annotated reads is cut and rendered on the GPU (bb_trim_batch) and appended to '{output_folder}/{label}.trimmed.fastq[.gz]'.
    Returns (total reads, trimmed, failed, split)."""
    cfg = config or TrimConfig()
    if cfg.sort_labels and cfg.only_side is not None:
        raise ValueError("Cannot enable only keeping left/right label and sorting; this is ambiguous")   # trim.rs:331-335
    os.makedirs(output_folder, exist_ok=True)
    space = _space_for(filtered_match_file, groups)
    total = 0
    if not space.groups:   # no annotations: every read is counted, none is written
        if cfg.failed_trimmed_writer:
            open(cfg.failed_trimmed_writer, "w").close()   # the reference creates it before the first read (trim.rs:364-370)
        for path in read_fastq_files:
            total += sum(1 for _ in read_fastq_records(str(path)))
        log(f"trim: {total} reads, 0 trimmed, 0 failed")
        if cfg.verbose:
            write_progress_log("trim", output_folder, [("Total:", total), ("Kept:", 0), ("Kept split:", 0), ("Failed:", 0)])
        return total, 0, 0, 0
    batches = list(read_annotation_tsv(filtered_match_file, space, group_consecutive=False))
    anno = batches[0]
    bounds = np.searchsorted(anno.rows["read_idx"], np.arange(len(anno.read_ids) + 1))
    by_id = {rid: k for k, rid in enumerate(anno.read_ids)}
    dm, _ = _context(space, device)
    trimmer = Trimmer(dm, cfg)
    writers = LabelWriters(output_folder, cfg, trimmer.tables)

    pend = []   # (annotation index, header line, seq, qual)

    def flush():
        if not pend:
            return
        rows = np.concatenate([anno.rows[bounds[k]:bounds[k + 1]] for k, *_ in pend])
        ver = np.concatenate([anno.verdicts[bounds[k]:bounds[k + 1]] for k, *_ in pend])
        rows["read_idx"] = np.repeat(np.arange(len(pend), dtype=np.uint32), [bounds[k + 1] - bounds[k] for k, *_ in pend])
        offsets = np.zeros(len(pend) + 1, dtype=np.uint64)
        np.cumsum([len(s) for _, _, s, _ in pend], out=offsets[1:])
        bases = np.frombuffer(b"".join(s for _, _, s, _ in pend), dtype=np.uint8)
        quals = np.frombuffer(b"".join(q for _, _, _, q in pend), dtype=np.uint8)
        res = trimmer.trim_batch(rows, ver, bases, quals, offsets, [h for _, h, _, _ in pend])
        writers.write(res, [anno.read_ids[k] for k, *_ in pend])
        pend.clear()

    try:
        for path in read_fastq_files:
            for h, s, q in read_fastq_records(str(path)):
                total += 1
                rid = split_fastq_header(h.decode("utf-8"))[0]
                k = by_id.get(rid)
                if k is None:
                    continue
                if len(q) != len(s):
                    raise ValueError(f"FASTQ record '{rid}' has no quality scores" if not q else f"FASTQ record '{rid}': {len(s)} bases, {len(q)} qualities")
                # the annotation must be THIS record's (another FASTQ with the same ids, re-basecalled reads, edited rows): the reference panics
                # on seq[start..end]; unchecked, the kernels would copy a neighbour's bases
                a = anno.rows[bounds[k]:bounds[k + 1]]
                if np.any(a["read_len"] != len(s)) or np.any(a["read_end_flank"] > len(s)) or np.any(a["read_start_flank"] > a["read_end_flank"]):
                    raise TsvError(f"{filtered_match_file}: read {rid!r}: the annotation says read_len {int(a['read_len'][0])}, flank ends up to "
                                   f"{int(a['read_end_flank'].max())}; the FASTQ record has {len(s)} bases (an annotation file of other reads?)")
                pend.append((k, h, s, q))
                if len(pend) >= batch_reads:
                    flush()
        flush()
    finally:
        writers.close()
        dm.close()
    log(f"trim: {total} reads, {writers.n_trimmed} trimmed, {writers.n_failed} failed, {writers.n_split} split")
    if cfg.verbose:   # trim.rs:341-345: in the output folder
        write_progress_log("trim", output_folder, [("Total:", total), ("Kept:", writers.n_trimmed), ("Kept split:", writers.n_split), ("Failed:", writers.n_failed)])
    return total, writers.n_trimmed, writers.n_failed, writers.n_split


def rows_to_tsv_text(rows, read_ids, space, verdicts=None):
    """header + lines, for tests and tools"""
    v = verdicts if verdicts is not None else np.zeros(len(rows), dtype=VERDICT_DTYPE)
    buf = io.StringIO()
    buf.write(TSV_HEADER + "\n")
    for line in _format(rows, read_ids, space, v):
        buf.write(line + "\n")
    return buf.getvalue()


__all__ = ["LabelSpace", "TsvError", "filter_file", "inspect_file", "trim_file", "read_annotation_tsv", "scan_labels", "parse_cuts",
           "patterns_from_files", "rows_to_tsv_text"]
