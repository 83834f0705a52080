"""tests/golden/policy_feasible.json (tools/policy_feasible.py): the settings of the unpinned assumptions that the reference's own vectors
(cigar_parse.rs:104-176) and invariants (searcher.rs:388 `expect`, the window of :445-456 on its kits and documented examples) leave open.
CPU: the file is not stale (stage A recomputed in full, stage B on a sample), the default policy is feasible, the build's default class list
and every consumer's policy list come from it."""
import itertools
import os
import subprocess
import sys

import pytest

from tests.common import ALTERNATIVES, GPU_POLICIES, TRACE_CLASSES, is_feasible, policy_feasible, split_feasible

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def test_file_is_current_and_default_is_feasible():
    """stage A (every one of the 2 592 joint settings against the reference's five sassy KATs, both halves of :104-123) recomputed here; stage B
    (no-panic + window on the planted barcode) on 300 reads of the NBD114-96 and the custom dual-end geometry: nothing the file calls
    feasible is refuted, the digest of the inputs (vectors, space, kit tables, example queries) matches."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "policy_feasible.py"), "--reads", "300", "--geometries", "nbd96,dual", "--check"],
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "agrees" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
    f = policy_feasible()
    assert f["default_feasible"] and is_feasible("") and is_feasible(f["default"])
    n = 1
    for k, v in f["space"].items():
        n *= len(v)
    assert f["n_joint_total"] == n == 2592 and 0 < f["n_joint_feasible"] < n
    if f["feasible_is_product_of_fields"]:
        m = 1
        for k in f["space"]:
            m *= len(f["feasible"][k])
        assert m == f["n_joint_feasible"]


def test_what_the_vectors_refute():
    """cigar_parse.rs:163-176 (text span (0, 2) of pattern rows 5..7 on GCAAAAGGGGGGGGGGGG, k = 8) decides between traceback orders: only
    classes that prefer Sub over Del where both apply, and do not put Del first, reproduce it.  rcpath=mirror survives every KAT (their
    region is symmetric enough) and is refuted by the reference's own dual-end example: an asymmetric flank puts the window off the barcode."""
    f = policy_feasible()
    assert f["space"]["trace"] == TRACE_CLASSES
    assert sorted(f["feasible"]["trace"]) == sorted(["MISD", "IMSD", "MSDI", "MSID", "SIMD"])
    assert set(f["stage_a"]["refuted_single_field"]) == {"trace=" + c for c in TRACE_CLASSES if c not in f["feasible"]["trace"]}
    assert all(v == ["overhang_including_bar"] for v in f["stage_a"]["refuted_single_field"].values())
    assert f["feasible"]["rcpath"] == ["fwd"]
    mirror = [s for s in f["stage_b"]["refuted"] if "rcpath=mirror" in s]
    assert mirror and all("dual" in f["stage_b"]["refuted"][s] for s in mirror)
    # nothing the reference holds reaches these: all of their values stay open
    for k in ("lm", "rc", "ovh", "tie"):
        assert f["feasible"][k] == f["space"][k], k
    # the invariants themselves: the default never trips them on any geometry
    for g, d in f["stage_b"]["default"].items():
        assert d["subpath_none"] == 0 and d["slice_panic"] == 0 and d["window_on_barcode"] >= 0.99 and d["on_target"] > 1000, (g, d)


def test_consumers_take_their_lists_from_the_file():
    """policy_sensitivity / bench.py policy_variants / ref_fit range over the feasible settings; refuted ones are listed, not searched"""
    ok, refuted = split_feasible(GPU_POLICIES)
    assert "rcpath=mirror" in refuted and "trace=DSIM" in refuted and "trace=MSID" in ok and "lm=strict" in ok and "lodhi=3:0.5:1110" in ok
    assert "trace=SMID" in ok                                     # the other spelling of MSID's class
    assert len([p for p in ALTERNATIVES["H3"] if p in ok]) == 5   # IMSD MSDI MSID SIMD + SMID
    import ref_fit

    f = policy_feasible()
    assert [t for t in ref_fit.SPACE["trace"]] and all(t.replace("SM", "MS") in f["feasible"]["trace"] for t in ref_fit.SPACE["trace"])
    assert ref_fit.SPACE["rcpath"] == ["fwd"] and "mirror" in ref_fit.REFUTED["rcpath"]
    import policy_feasible as pf

    # the build's default class list = the feasible classes, in bb_prio.h's order
    order = pf.build_class_order()
    assert order[0] == "MISD" and sorted(order) == sorted(TRACE_CLASSES)
    assert f["feasible_trace_class_indices"] == pf.feasible_class_indices() == sorted(order.index(c) for c in f["feasible"]["trace"])
    mk = open(os.path.join(ROOT, "barbell_amd", "csrc", "Makefile")).read()
    assert "policy_feasible.json" in mk


def test_class_order_mirrors_bb_prio_h(tmp_path):
    """tools/policy_feasible.py::build_class_order against the header's constexpr table (host-only compile)"""
    import policy_feasible as pf

    src = tmp_path / "cls.cpp"
    src.write_text('#include <cstdio>\n#include "barbell_amd/csrc/bb_prio.h"\nint main() { for (int i = 0; i < BB_PRIO_CLASSES; ++i) { const uint32_t p = BB_PRIO_TABLE.cls[i]; '
                   'printf("%c%c%c%c\\n", "MSID"[p & 3], "MSID"[(p >> 2) & 3], "MSID"[(p >> 4) & 3], "MSID"[(p >> 6) & 3]); } return 0; }\n')
    exe = tmp_path / "cls"
    subprocess.check_call(["g++", "-std=c++17", "-I", ROOT, "-o", str(exe), str(src)])
    assert subprocess.run([str(exe)], capture_output=True, text=True).stdout.split() == pf.build_class_order()
