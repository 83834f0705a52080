"""Filter step on the row stream — host mirror of src/filter/pattern.rs and src/filter/filter.rs.

`pattern_from_str` is the reference's `pattern_from_str!` macro (pattern.rs:242-383): same grammar,
same defaults, same "could not convert all your pattern elements" failure.  `Filter` resolves the
label constraints of the parsed patterns against the query groups (labels -> histogram slots) and
hands them to the HIP library (`bb_filter_set` / `bb_filter_rows`), which runs `match_pattern` +
`check_filter_pass` for every read of a batch on the GPU.  `format_cuts` is the `cuts` column of
filtered.tsv (searcher.rs:91-106)."""
import ctypes as C
import re
from dataclasses import dataclass, field

import numpy as np

from . import _abi
from .kits import _data

MAX_CUTS = 3
REL = {"left": 1, "right": 2, "prev_left": 3}
MT = {"Ftag": _abi.BB_FTAG, "Rtag": _abi.BB_RTAG, "Fflank": _abi.BB_FFLANK, "Rflank": _abi.BB_RFLANK}


@dataclass(frozen=True)
class Cut:  # pattern.rs:15-19
    group_id: int
    direction: str  # "Before" | "After"

    @staticmethod
    def from_pattern_string(s):  # pattern.rs:69-85: ">>", "<<", ">>3"
        if len(s) < 2:  # the reference slices [..2] and panics here
            raise ValueError(f"cut marker too short: {s!r}")
        if s[:2] not in (">>", "<<"):
            return None
        try:
            gid = 0 if len(s) == 2 else _int(s[2:], signed=False)
        except ValueError:
            return None
        return Cut(gid, "After" if s[:2] == ">>" else "Before")

    def __str__(self):  # pattern.rs:88-95
        return f"{self.direction}({self.group_id})"


@dataclass
class PatternElement:  # pattern.rs:21-30
    match_type: int
    orientation: int = -1          # -1 any, 0 Fwd, 1 Rc
    label: str = None
    placeholder: int = -1
    range: tuple = (0, 0)
    relative_to: int = 0           # 0 none, 1 left, 2 right, 3 prev_left
    cuts: list = field(default_factory=list)


@dataclass
class Pattern:
    elements: list


def _int(s, signed=True):
    """str::parse::<isize>/<usize>: optional sign (no '-' for usize), ASCII digits only"""
    if not re.fullmatch(r"[+-]?[0-9]+" if signed else r"\+?[0-9]+", s):
        raise ValueError(s)
    return int(s)


def _parse_range(s):  # pattern.rs:249-261
    parts = s.strip("()").split("..")
    if len(parts) != 2:
        return None
    try:
        return _int(parts[0].strip()), _int(parts[1].strip())
    except ValueError:
        return None


def _parse_position(p):  # pattern.rs:263-279
    pos_parts = p.split("(")
    if len(pos_parts) != 2:
        return None
    name = pos_parts[0].lstrip("@")
    if name not in REL:
        return None
    rng = _parse_range(p[len(pos_parts[0]):].strip())
    if rng is None:
        return None
    return REL[name], rng


def _parse_element(s):  # pattern.rs:287-356
    parts = s.split("[", 1)
    if len(parts) != 2:
        return None
    name = parts[0].strip()
    if name in ("Flank", "flank"):
        raise ValueError("Flank is not valid, use Fflank or Rflank")
    if name not in MT:
        return None
    el = PatternElement(MT[name])
    for param in (x.strip() for x in parts[1].rstrip("]").split(",")):
        if param == "fw":
            el.orientation = _abi.BB_FWD
        elif param == "rc":
            el.orientation = _abi.BB_RC
        elif param.startswith("@"):
            r = _parse_position(param)
            if r is not None:
                el.relative_to, el.range = r
        elif param.startswith("?"):
            try:
                el.placeholder = _int(param[1:], signed=False)
            except ValueError:
                pass
        elif param.startswith(">") or param.startswith("<"):
            c = Cut.from_pattern_string(param)
            if c is not None:
                el.cuts.append(c)
        elif param == "*":
            pass
        else:
            el.label = param.strip('"')
    return el


def pattern_from_str(s):
    elements = [e for e in (_parse_element(x.strip()) for x in s.split("__")) if e is not None]
    if s.count("__") + 1 != len(elements):  # basic_verify pattern.rs:281-285
        raise ValueError(f"Pattern parse error for: {s!r}")
    return Pattern(elements)


def kit_patterns(kit, maximize=False):
    """Default filter patterns of a kit preset (kits.rs:175-236; safe unless --maximize)."""
    d = _data()
    if kit not in d["kit_filter"] and "." in kit:
        kit = kit.replace(".", "-")
    return [pattern_from_str(p) for p in d["pattern_sets"][d["kit_filter"][kit]["maximize" if maximize else "safe"]]]


def patterns_from_files(paths):  # filter.rs:137-181
    if not paths:
        raise ValueError("No filter pattern files provided")
    pats = [pattern_from_str(l.strip()) for path in paths for l in open(path) if l.strip()]
    if not pats:
        raise ValueError("No filter patterns found")
    return pats


# ---- C structs --------------------------------------------------------------------------------
class CutC(C.Structure):
    _fields_ = [("direction", C.c_uint8), ("_pad", C.c_uint8), ("group_id", C.c_uint16)]


class PatternElemC(C.Structure):
    _fields_ = [("match_type", C.c_uint8), ("orientation", C.c_int8), ("relative_to", C.c_uint8), ("n_cuts", C.c_uint8),
                ("placeholder", C.c_int32), ("range_lo", C.c_int64), ("range_hi", C.c_int64),
                ("label_ok", C.POINTER(C.c_uint8)), ("cuts", CutC * MAX_CUTS)]


class PatternC(C.Structure):
    _fields_ = [("elems", C.POINTER(PatternElemC)), ("n_elems", C.c_uint32)]


VERDICT_DTYPE = np.dtype([("pass", "u1"), ("n_cuts", "u1"), ("match_idx", "<u2"),
                          ("cuts", [("direction", "u1"), ("_pad", "u1"), ("group_id", "<u2")], (MAX_CUTS,))])
assert VERDICT_DTYPE.itemsize == 16


def slot_labels(groups):
    """label of every histogram slot: per group n_seqs labels then "flank" (searcher.rs:258)"""
    out = []
    for g in groups:
        out.extend(g.labels)
        out.append("flank")
    return out


def compile_patterns(patterns, groups):
    """-> (PatternC array, label_ids uint32[slots], keepalive).  Label constraints become one byte per
    slot: exact label, or substring for "~xyz" (pattern.rs:108-121)."""
    labels = slot_labels(groups)
    ids = {}
    label_ids = np.array([ids.setdefault(l, len(ids)) for l in labels], dtype=np.uint32)
    arr = (PatternC * max(1, len(patterns)))()
    keep = []
    for p in patterns:
        if len({e.placeholder for e in p.elements if e.placeholder >= 0}) > 16:
            raise ValueError("more than 16 distinct ?N placeholders in one pattern (kernel limit, include/barbell_amd_filter.h)")
    for i, p in enumerate(patterns):
        elems = (PatternElemC * len(p.elements))()
        for j, e in enumerate(p.elements):
            if len(e.cuts) > MAX_CUTS:
                raise ValueError("more than 3 cut markers on one pattern element (kernel limit, include/barbell_amd_filter.h)")
            if any(cut.group_id > 0xFFFF for cut in e.cuts):
                raise ValueError("cut group id above 65535 (kernel limit, include/barbell_amd_filter.h)")
            c = elems[j]
            c.match_type, c.orientation, c.relative_to, c.n_cuts = e.match_type, e.orientation, e.relative_to, len(e.cuts)
            c.placeholder, c.range_lo, c.range_hi = e.placeholder, e.range[0], e.range[1]
            if e.label is not None:
                if e.label.startswith("~"):
                    ok = np.array([e.label[1:] in l for l in labels], dtype=np.uint8)
                else:
                    ok = np.array([e.label == l for l in labels], dtype=np.uint8)
                keep.append(ok)
                c.label_ok = ok.ctypes.data_as(C.POINTER(C.c_uint8))
            for q, cut in enumerate(e.cuts):
                c.cuts[q].direction = 1 if cut.direction == "After" else 0
                c.cuts[q].group_id = cut.group_id
        keep.append(elems)
        arr[i].elems = C.cast(elems, C.POINTER(PatternElemC))
        arr[i].n_elems = len(p.elements)
    return arr, label_ids, keep


def format_cuts(v):
    """`cuts` column: "After(0):1,Before(0):2" (searcher.rs:91-106); empty when the row has none"""
    return ",".join(f"{'After' if int(c['direction']) else 'Before'}({int(c['group_id'])}):{int(v['match_idx'])}"
                    for c in v["cuts"][: int(v["n_cuts"])])


class Filter:
    """check_filter_pass for every read of a batch, on the GPU of `demuxer`."""

    def __init__(self, demuxer, patterns):
        from ._lib import lib

        self.dm = demuxer
        self.patterns = list(patterns)
        arr, label_ids, keep = compile_patterns(self.patterns, demuxer.queries)
        rc = lib().bb_filter_set(demuxer._ctx(), arr, len(self.patterns), label_ids.ctypes.data)
        demuxer._check(rc)

    def verdicts(self, rows):
        from ._lib import lib

        rows = np.ascontiguousarray(rows, dtype=_abi.ROW_DTYPE)
        out = np.zeros(len(rows), dtype=VERDICT_DTYPE)
        if len(rows):
            self.dm._check(lib().bb_filter_rows(self.dm._ctx(), rows.ctypes.data, len(rows), out.ctypes.data))
        return out

    def verdicts_ingested(self, d_rows, n_rows, download=True):
        """verdicts for rows already in HBM; they stay in the demuxer's "verdicts" buffer, a host copy is returned
        (unless download=False: the TSV renderer and the trim step read them where they are)"""
        from ._lib import lib

        out = np.zeros(n_rows if download else 0, dtype=VERDICT_DTYPE)
        d = self.dm.buf("verdicts").ensure((n_rows + 1) * 16)
        if n_rows:
            self.dm._check(lib().bb_filter_rows_dev(self.dm._ctx(), d_rows, n_rows, d))
            if download:
                self.dm.buf("verdicts").download(out)
        return out

    def verdicts_dev(self, d_rows, n_rows, d_out):
        from ._lib import lib

        self.dm._check(lib().bb_filter_rows_dev(self.dm._ctx(), d_rows, n_rows, d_out))
