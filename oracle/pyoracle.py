"""ctypes binding of oracle/libbb_oracle.so — TEST INFRASTRUCTURE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the
product package `barbell_amd` never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

from barbell_amd import _abi

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libbb_oracle.so")


def build(force=False):
    src = os.path.join(_HERE, "bb_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _SO


class Match(C.Structure):
    _fields_ = [
        ("text_start", C.c_int32), ("text_end", C.c_int32),
        ("pattern_start", C.c_int32), ("pattern_end", C.c_int32),
        ("cost", C.c_int32), ("strand", C.c_int32), ("pattern_idx", C.c_int32),
        ("n_ops", C.c_int32), ("ops", C.POINTER(C.c_uint8)), ("rc_text_len", C.c_int32), ("rc_mirror_len", C.c_int32),
    ]


class Pos(C.Structure):
    _fields_ = [("i", C.c_int32), ("j", C.c_int32)]


class Diag(C.Structure):
    """bbo_diag (oracle/bb_oracle.h): the reference's invariants on the path, counted (tools/policy_feasible.py)"""
    _fields_ = [(k, C.c_uint64) for k in ("flank_matches", "region_none", "slice_panic", "subpath_none", "on_target", "window_overlaps",
                                           "window_covers", "tag_rows_on_target", "tag_rows_correct", "rows")]
TRUTH_PER_READ, TRUTH_FIELDS = 2, 7


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        L = C.CDLL(_SO)
        L.bbo_search.restype = C.c_int
        L.bbo_search.argtypes = [C.c_char_p, C.c_int, C.c_char_p, C.c_int, C.c_int, C.c_float, C.c_int,
                                 C.POINTER(C.POINTER(Match))]
        L.bbo_free_matches.argtypes = [C.POINTER(Match), C.c_int]
        L.bbo_to_path.argtypes = [C.POINTER(Match), C.POINTER(Pos)]
        L.bbo_map_pat_to_text_with_cost.argtypes = [C.POINTER(Match), C.c_int, C.c_int] + [C.POINTER(C.c_int)] * 5
        L.bbo_get_matching_region.argtypes = [C.POINTER(Match), C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.bbo_lodhi.restype = C.c_double
        L.bbo_lodhi.argtypes = [C.c_char_p, C.c_int]
        L.bbo_edit_cut_off.argtypes = [C.c_int]
        L.bbo_rel_dist_to_end.argtypes = [C.c_long, C.c_long]
        L.bbo_collapse.argtypes = [C.c_void_p, C.c_int, C.c_float]
        L.bbo_iupac_code.restype = C.c_uint8
        L.bbo_iupac_code.argtypes = [C.c_uint8]
        L.bbo_create.argtypes = [C.POINTER(_abi.GroupDesc), C.c_uint32, C.POINTER(_abi.Params), C.POINTER(C.c_void_p)]
        L.bbo_create_policy.argtypes = [C.POINTER(_abi.GroupDesc), C.c_uint32, C.POINTER(_abi.Params), C.POINTER(_abi.Policy), C.POINTER(C.c_void_p)]
        L.bbo_set_policy.argtypes = [C.POINTER(_abi.Policy)]
        L.bbo_destroy.argtypes = [C.c_void_p]
        L.bbo_group_get_info.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(_abi.GroupInfo)]
        L.bbo_group_get_flank.argtypes = [C.c_void_p, C.c_uint32, C.c_char_p]
        L.bbo_group_get_pattern.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_int, C.c_char_p]
        L.bbo_annotate_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint64,
                                         C.POINTER(C.c_uint64), C.c_int]
        L.bbo_annotate_batch_fast.argtypes = L.bbo_annotate_batch.argtypes
        L.bbo_annotate_diag.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_int, C.c_int, C.POINTER(Diag)]
        L.bbo_set_full_trace.argtypes = [C.c_int]
        L.bbo_filter_rows.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p]
        L.bbo_trim_batch.argtypes = [C.c_void_p] * 7 + [C.c_uint64] + [C.c_void_p] * 4 + [C.c_uint32, C.c_void_p, C.c_uint64, C.c_void_p,
                                     C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p]
        L.bbo_inspect_rows.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32, C.c_void_p]
        L.bbo_fastq_parse.argtypes = [C.c_void_p, C.c_uint64, C.c_int] + [C.c_void_p] * 8
        _lib = L
    return _lib


OPS = "=XID"  # Match, Sub, Ins, Del


class PyMatch:
    """Python copy of one bbo_match plus its path."""

    def __init__(self, m):
        L = lib()
        for f in ("text_start", "text_end", "pattern_start", "pattern_end", "cost", "strand", "n_ops"):
            setattr(self, f, getattr(m, f))
        self.ops = bytes(m.ops[i] for i in range(m.n_ops))
        path = (Pos * max(1, m.n_ops))()
        L.bbo_to_path(C.byref(m), path)
        self.path = [(path[i].i, path[i].j) for i in range(m.n_ops)]
        self.cigar = "".join(OPS[o] for o in self.ops)


def search(pattern, text, k, alpha=None, rc=True):
    """sassy Searcher::search restated.  Returns (list[PyMatch], raw handle tuple to free)."""
    L = lib()
    out = C.POINTER(Match)()
    n = L.bbo_search(pattern, len(pattern), text, len(text), k, -1.0 if alpha is None else float(alpha), int(rc), C.byref(out))
    res = [PyMatch(out[i]) for i in range(n)]
    return res, (out, n)


def map_pat_to_text_with_cost(handle, idx, p_start, p_end):
    L = lib()
    out, n = handle
    v = [C.c_int() for _ in range(5)]
    ok = L.bbo_map_pat_to_text_with_cost(C.byref(out[idx]), p_start, p_end, *[C.byref(x) for x in v])
    if not ok:
        return None
    pl, ph, tl, th, cost = [x.value for x in v]
    return (pl, ph), (tl, th), cost


def get_matching_region(handle, idx, start, end):
    L = lib()
    out, n = handle
    lo, hi = C.c_int(), C.c_int()
    ok = L.bbo_get_matching_region(C.byref(out[idx]), start, end, C.byref(lo), C.byref(hi))
    return (lo.value, hi.value) if ok else None


def free_matches(handle):
    out, n = handle
    lib().bbo_free_matches(out, n)


def lodhi(ops):
    return lib().bbo_lodhi(bytes(ops), len(ops))


class policy:
    """with pyoracle.policy("lm=left"): ... — the policy of the stand-alone entry points (search, lodhi) and of Oracle()
    objects made without one (include/barbell_amd_policy.h)"""

    _stack = []

    def __init__(self, text=None):
        self.p = _abi.policy_from_str(text)

    def __enter__(self):
        assert lib().bbo_set_policy(C.byref(self.p)) == 0
        policy._stack.append(self.p)
        return self.p

    def __exit__(self, *a):
        policy._stack.pop()
        lib().bbo_set_policy(C.byref(policy._stack[-1]) if policy._stack else None)   # nested uses restore the outer one


def collapse(rows, overlap=0.8):
    rows = np.ascontiguousarray(rows, dtype=_abi.ROW_DTYPE).copy()
    n = lib().bbo_collapse(rows.ctypes.data, len(rows), overlap)
    return rows[:n]


def inspect_rows(rows, verdicts=None, bucket_size=250):
    """get_group_structure per row (inspect.rs:15-117) -> INSPECT_DTYPE array"""
    from barbell_amd import inspect_rows as I

    rows = np.ascontiguousarray(rows, dtype=_abi.ROW_DTYPE)
    out = np.zeros(len(rows), dtype=I.INSPECT_DTYPE)
    v = None if verdicts is None else np.ascontiguousarray(verdicts)
    rc = lib().bbo_inspect_rows(rows.ctypes.data, None if v is None else v.ctypes.data, len(rows), bucket_size, out.ctypes.data)
    assert rc == 0, rc
    return out


def fastq_parse(text, final_block=True):
    """scalar FASTQ block parser -> (rc, info dict, arrays dict)"""
    from barbell_amd import fastq as Q

    text = np.ascontiguousarray(np.frombuffer(bytes(text), dtype=np.uint8))
    info = Q.FastqInfo()
    rc = lib().bbo_fastq_parse(text.ctypes.data, len(text), int(final_block), C.addressof(info), *([None] * 7))
    d = {k: getattr(info, k) for k in ("n_records", "consumed", "n_bases", "n_hdr", "bad_record")}
    if rc != 0:
        return rc, d, None
    n = info.n_records
    a = {"offsets": np.zeros(n + 1, np.uint64), "bases": np.zeros(info.n_bases, np.uint8), "quals": np.zeros(info.n_bases, np.uint8),
         "hdr": np.zeros(info.n_hdr, np.uint8), "hdr_offsets": np.zeros(n + 1, np.uint64), "id_len": np.zeros(n, np.uint32),
         "desc_start": np.zeros(n, np.uint32)}
    rc = lib().bbo_fastq_parse(text.ctypes.data, len(text), int(final_block), C.addressof(info),
                               *[a[k].ctypes.data for k in ("offsets", "bases", "quals", "hdr", "hdr_offsets", "id_len", "desc_start")])
    return rc, d, a


class Oracle:
    """Same shape as barbell_amd.Demuxer, CPU restatement underneath."""

    def __init__(self, groups, alpha=0.4, min_score=0.2, min_score_diff=0.1, policy=None):
        L = lib()
        arr, keep = _abi.make_group_descs(groups)
        p = _abi.Params(alpha, min_score, min_score_diff, -1)
        h = C.c_void_p()
        if policy is None:
            rc = L.bbo_create(arr, len(groups), C.byref(p), C.byref(h))
        else:
            self.policy = _abi.policy_from_str(policy)
            rc = L.bbo_create_policy(arr, len(groups), C.byref(p), C.byref(self.policy), C.byref(h))
        self.rc = rc
        self.h = h if rc == 0 else None
        self.n_groups = len(groups)
        if rc != 0:
            raise ValueError(f"bbo_create failed: {rc}")

    def info(self, g):
        i = _abi.GroupInfo()
        assert lib().bbo_group_get_info(self.h, g, C.byref(i)) == 0
        return i

    def flank(self, g):
        i = self.info(g)
        buf = C.create_string_buffer(i.flank_len)
        lib().bbo_group_get_flank(self.h, g, buf)
        return buf.raw

    def pattern(self, g, idx, rc=False):
        i = self.info(g)
        buf = C.create_string_buffer(i.pattern_len)
        assert lib().bbo_group_get_pattern(self.h, g, idx, int(rc), buf) == 0
        return buf.raw

    @staticmethod
    def fast_is_simd():
        """the timing path (fast=True) runs its AVX-512 forms on this CPU (oracle/bb_oracle_simd.h)"""
        return bool(lib().bbo_fast_is_simd())

    def annotate(self, bases, offsets, n_threads=1, fast=False):
        """fast=True: the bit-parallel timing path (bench.py's cpu_baseline); same rows"""
        bases = np.ascontiguousarray(bases, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        n = len(offsets) - 1
        cap = max(64, 4 * n)
        while True:
            rows = np.zeros(cap, dtype=_abi.ROW_DTYPE)
            nr = C.c_uint64()
            fn = lib().bbo_annotate_batch_fast if fast else lib().bbo_annotate_batch
            rc = fn(self.h, bases.ctypes.data, offsets.ctypes.data, n, rows.ctypes.data, cap, C.byref(nr), n_threads)
            if rc == _abi.BB_E_CAPACITY:
                cap = int(nr.value)
                continue
            assert rc == 0, rc
            return rows[: nr.value]

    def annotate_diag(self, bases, offsets, truth=None, n_threads=1, fast=True):
        """the per-read procedure with the reference's invariants counted instead of aborting -> dict of bbo_diag's counters;
        truth: int32[n_reads, TRUTH_PER_READ, TRUTH_FIELDS] (group, strand, construct lo / hi, barcode lo / hi, barcode idx; group -1 = none)"""
        bases = np.ascontiguousarray(bases, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        n = len(offsets) - 1
        if truth is not None:
            truth = np.ascontiguousarray(truth, dtype=np.int32)
            assert truth.shape == (n, TRUTH_PER_READ, TRUTH_FIELDS), truth.shape
        d = Diag()
        rc = lib().bbo_annotate_diag(self.h, bases.ctypes.data, offsets.ctypes.data, n, None if truth is None else truth.ctypes.data,
                                     n_threads, int(fast), C.byref(d))
        assert rc == 0, rc
        return {k: int(getattr(d, k)) for k, _ in Diag._fields_}

    def annotate_reads(self, reads, n_threads=1):
        return self.annotate(*_abi.pack_reads(reads), n_threads=n_threads)

    def filter_rows(self, patterns, groups, rows):
        """check_filter_pass per read (filter.rs:183-214) -> verdict array"""
        from barbell_amd import filter as F

        arr, label_ids, keep = F.compile_patterns(patterns, groups)
        rows = np.ascontiguousarray(rows, dtype=_abi.ROW_DTYPE)
        out = np.zeros(len(rows), dtype=F.VERDICT_DTYPE)
        rc = lib().bbo_filter_rows(self.h, arr, len(patterns), label_ids.ctypes.data, rows.ctypes.data, len(rows), out.ctypes.data)
        assert rc == 0, rc
        return out

    def trim_batch(self, groups, cfg, rows, verdicts, bases, quals, offsets, headers):
        """process_read_and_anno + record text for every passing read (trim.rs:127-300, 447-460) -> TrimResult"""
        from barbell_amd import filter as F
        from barbell_amd import trim as T

        tb = T.LabelTables(groups, cfg)
        c = T.config_c(cfg)
        rows = np.ascontiguousarray(rows, dtype=_abi.ROW_DTYPE)
        verdicts = np.ascontiguousarray(verdicts, dtype=F.VERDICT_DTYPE)
        bases = np.ascontiguousarray(bases, dtype=np.uint8)
        quals = np.ascontiguousarray(quals, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        n = len(offsets) - 1
        blob, hoff, id_len, desc = headers if isinstance(headers, tuple) else T.pack_headers(headers)
        blob = np.ascontiguousarray(blob)
        h = T.HeadersC(blob.ctypes.data, hoff.ctypes.data, id_len.ctypes.data, desc.ctypes.data)
        status = np.zeros(n, dtype=np.uint8)
        tl, ns, nsp = C.c_uint64(), C.c_uint64(), C.c_uint32()
        args = (self.h, C.addressof(c), tb.is_flank.ctypes.data, tb.part_rank.ctypes.data, tb.label_ids.ctypes.data, rows.ctypes.data,
                verdicts.ctypes.data, len(rows), bases.ctypes.data, quals.ctypes.data, offsets.ctypes.data, C.addressof(h), n)
        rc = lib().bbo_trim_batch(*args, None, 0, C.addressof(tl), None, 0, C.addressof(ns), None, 0, C.addressof(nsp), status.ctypes.data)
        assert rc in (0, _abi.BB_E_CAPACITY), rc                      # sizing call
        text = np.empty(tl.value + 1, dtype=np.uint8)
        slices = np.zeros(ns.value + 1, dtype=T.SLICE_DTYPE)
        spans = np.zeros(nsp.value + 1, dtype=T.SPAN_DTYPE)
        rc = lib().bbo_trim_batch(self.h, C.addressof(c), tb.is_flank.ctypes.data, tb.part_rank.ctypes.data, tb.label_ids.ctypes.data,
                                  rows.ctypes.data, verdicts.ctypes.data, len(rows), bases.ctypes.data, quals.ctypes.data,
                                  offsets.ctypes.data, C.addressof(h), n, text.ctypes.data, len(text), C.addressof(tl), slices.ctypes.data,
                                  len(slices), C.addressof(ns), spans.ctypes.data, len(spans), C.addressof(nsp), status.ctypes.data)
        assert rc == 0, rc
        return T.TrimResult(text[: tl.value], slices[: ns.value], spans[: nsp.value], status)

    def close(self):
        if self.h:
            lib().bbo_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
