"""Kit presets -> query groups.  Host-side mirror of the reference's kit expansion
(`BarcodeGroup::new_from_kit`, src/annotate/barcodes.rs:251-299; label ranges `get_barcodes`,
src/kits/kits.rs:741-816; sequence lookup `lookup_barcode_seq`, kits.rs:1074-1103; kit-name map
`get_kit_info`, kits.rs:635-708).  The sequence/template tables are data extracted into
data/kits.json by tools/extract_kits.py."""
import json
import os
import re

from . import _abi

_DATA = None


def _data():
    global _DATA
    if _DATA is None:
        with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "kits.json")) as f:
            _DATA = json.load(f)
    return _DATA


def parse_label_simple(label):
    """kits.rs:710-739: 'BC01' / 'NB12A' / 'RBK26' -> (prefix, number, a_flag), case-insensitive."""
    m = re.match(r"^([A-Za-z]*)(\d+)(A?)", label.upper())
    if not m or not m.group(2):
        raise ValueError(f"Invalid numeric part in label: {label}")
    return m.group(1), int(m.group(2)), m.group(3) == "A"


def get_barcodes(from_label, to_label, use_12a_flag=False):
    """kits.rs:741-816"""
    pf_from, from_num, from_a = parse_label_simple(from_label)
    pf_to, to_num, to_a = parse_label_simple(to_label)
    if pf_from != pf_to:
        raise ValueError(f"Mismatched label prefixes: {pf_from} vs {pf_to}")
    start, end = min(from_num, to_num), max(from_num, to_num)
    if pf_from == "AB":
        return [f"AB{i:02d}" for i in range(start, end + 1)]
    labels = [f"BC{i:02d}" for i in range(start, end + 1)]
    use_12a = use_12a_flag or ((from_a or to_a) and (start <= 12 <= end))
    if use_12a:
        labels = ["BC12A" if l == "BC12" else l for l in labels]
    if pf_from == "NB":
        labels = [l.replace("BC", "NB", 1) if l.startswith("BC") else l for l in labels]
    if pf_from == "RBK":
        special = {26, 39, 40, 48, 54, 60}
        labels = [l.replace("BC", "RBK", 1) if l.startswith("BC") and int(l[2:4]) in special else l for l in labels]
    return labels


def lookup_barcode_seq(label):
    """kits.rs:1074-1103"""
    d = _data()
    prefix, number, is_a = parse_label_simple(label)
    idx = max(number - 1, 0)

    def get(tab):
        return tab[idx] if idx < len(tab) else None

    if prefix in ("BC", "NB"):
        if is_a and number == 12:
            return d["BC12A"]
        return get(d["seqs"][prefix])
    if prefix in ("AB", "BP"):
        return get(d["seqs"][prefix])
    if prefix == "RBK":
        sp = d["RBK_special"].get(str(number))
        return sp if sp else get(d["seqs"]["BC"])
    return None


def kit_templates(kit):
    """kits.rs:635-708 (incl. the '.' -> '-' retry)"""
    d = _data()
    if kit not in d["kits"] and "." in kit:
        kit = kit.replace(".", "-")
    if kit not in d["kits"]:
        raise ValueError(f"Unknown or unsupported kit: {kit}")
    return d["templates"][d["kits"][kit]]


def supported_kits():
    return sorted(_data()["kits"])


class QueryGroup:
    """What `BarcodeGroup::new` takes: sequences, labels, type (barcodes.rs:106-110)."""

    def __init__(self, seqs, labels, match_type, flank_k=None):
        self.seqs = [bytes(s) for s in seqs]
        self.labels = list(labels)
        self.match_type = match_type
        self.flank_k = flank_k

    def set_flank_threshold(self, k):  # barcodes.rs:318-320
        self.flank_k = k

    def as_tuple(self):
        return (self.seqs, self.match_type, self.flank_k)


def groups_from_kit(kit, use_extended=False, flank_max_errors=None):
    """barcodes.rs:251-299 + annotator.rs:207-231"""
    groups = []
    for tmpl in kit_templates(kit):
        if tmpl["type"] == "Extended" and not use_extended:
            continue
        labels = get_barcodes(tmpl["from"], tmpl["to"], tmpl["use_12a"])
        seqs = []
        for lab in labels:
            bar = lookup_barcode_seq(lab)
            if bar is None:
                raise ValueError("Barcode not found - odd - raise issue")
            seqs.append("".join(bar if p in ("{BAR}", "**") else p for p in tmpl["parts"]).encode())
        typ = _abi.BB_FTAG if tmpl["side"] == "Left" else _abi.BB_RTAG
        groups.append(QueryGroup(seqs, labels, typ, flank_max_errors))
    return groups


def read_fasta(path):
    labels, seqs, cur = [], [], []
    with open(path) as f:
        for line in f:
            line = line.strip()
            if not line:
                continue
            if line.startswith(">"):
                if cur:
                    seqs.append("".join(cur))
                    cur = []
                labels.append(line[1:].split()[0])
            else:
                cur.append(line)
    if cur:
        seqs.append("".join(cur))
    return labels, seqs


def group_from_fasta(path, match_type=_abi.BB_FTAG, flank_max_errors=None):
    """barcodes.rs:302-315 (needletail normalize(true): upper-case, IUPAC kept)"""
    labels, seqs = read_fasta(path)
    return QueryGroup([s.upper().encode() for s in seqs], labels, match_type, flank_max_errors)
