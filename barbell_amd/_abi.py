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


class Policy(C.Structure):
    """bb_policy (include/barbell_amd_policy.h): the switchable assumptions about sassy / cigar-lodhi-rs"""
    _fields_ = [
        ("lm_rule", C.c_uint8), ("rc_order", C.c_uint8), ("trace_prio", C.c_uint8 * 4), ("ovh_round", C.c_uint8), ("bar_tie", C.c_uint8),
        ("lodhi_p", C.c_uint8), ("lodhi_exp", C.c_uint8 * 4), ("rc_path", C.c_uint8), ("_pad", C.c_uint8 * 2), ("lodhi_lambda", C.c_double),
    ]


LM_RULES = ("right", "left", "strict")
OVH_ROUNDS = ("floor", "ceil", "near")
POLICY_DEFAULT = "lm=right,rc=scan,trace=MISD,ovh=floor,tie=first,lodhi=3:0.5:1111,rcpath=fwd"


def policy_from_str(text=None):
    """the text form of include/barbell_amd_policy.h (same grammar as bb_policy_parse) -> Policy; None / "" = default"""
    p = Policy(0, 0, (C.c_uint8 * 4)(0, 2, 1, 3), 0, 0, 3, (C.c_uint8 * 4)(1, 1, 1, 1), 0, (C.c_uint8 * 2)(), 0.5)
    if isinstance(text, Policy):
        return text
    for tok in (text or "").replace(" ", ",").split(","):
        if not tok:
            continue
        k, _, v = tok.partition("=")
        if k == "lm":
            p.lm_rule = LM_RULES.index(v)
        elif k == "rc":
            p.rc_order = ("scan", "fwd").index(v)
        elif k == "trace":
            if sorted(v) != sorted("MSID"):
                raise ValueError(f"trace={v}: a permutation of MSID")
            for i, ch in enumerate(v):
                p.trace_prio[i] = "MSID".index(ch)
        elif k == "ovh":
            r, _, w = v.partition(":")
            if w not in ("", "f32", "f64"):
                raise ValueError(tok)
            p.ovh_round = OVH_ROUNDS.index(r) | (4 if w == "f64" else 0)
        elif k == "tie":
            p.bar_tie = ("first", "last").index(v)
        elif k == "lodhi":
            pp, lam, e = v.split(":")
            if len(e) != 4 or not 1 <= int(pp) <= 4 or not 0.0 < float(lam) <= 1.0 or any(c not in "0123" for c in e):
                raise ValueError(tok)
            p.lodhi_p, p.lodhi_lambda = int(pp), float(lam)
            for i, ch in enumerate(e):
                p.lodhi_exp[i] = int(ch)
        elif k == "rcpath":
            p.rc_path = ("fwd", "mirror").index(v)
        else:
            raise ValueError(f"unknown policy key in {tok!r}")
    return p


def policy_to_str(p):
    return (f"lm={LM_RULES[p.lm_rule]},rc={('scan', 'fwd')[p.rc_order]},trace={''.join('MSID'[x] for x in p.trace_prio)},"
            f"ovh={OVH_ROUNDS[p.ovh_round & 3]}{':f64' if p.ovh_round & 4 else ''},tie={('first', 'last')[p.bar_tie]},"
            f"lodhi={p.lodhi_p}:{p.lodhi_lambda!r}:{''.join(str(x) for x in p.lodhi_exp)},rcpath={('fwd', 'mirror')[p.rc_path]}")


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
