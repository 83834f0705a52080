"""integration/gpu.rs + integration/annotator.patch: the reference-side binding as complete text.  No Rust toolchain exists
in the image, so the text is checked instead of compiled: `#[repr(C)]` layouts and the extern block against include/*.h (offsets from a C
program compiled here) and against the ctypes mirror; the patch must apply to annotator.rs's seam when the reference tree is present."""
import os
import re
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOOL = os.path.join(ROOT, "tools", "check_rust_layout.py")
REF = "/root/reference/src/annotate"


def test_rust_layouts_match_the_headers():
    r = subprocess.run([sys.executable, TOOL], capture_output=True, text=True)
    assert r.returncode == 0 and "agree" in r.stdout, r.stdout + r.stderr


def test_the_checker_notices_a_wrong_layout(tmp_path):
    text = open(os.path.join(ROOT, "integration", "gpu.rs")).read()
    for old, new in (("pub barcode_idx: i16,", "pub barcode_idx: i32,"), ("pub min_score: f64,", "pub min_score: f32,"),
                     ("fn bb_destroy(ctx: *mut BbCtx);", "fn bb_destroy(ctx: *mut BbCtx, x: i32);"), ("BB_E_CAPACITY: i32 = -7", "BB_E_CAPACITY: i32 = -6")):
        assert old in text
        p = tmp_path / "gpu.rs"
        p.write_text(text.replace(old, new))
        r = subprocess.run([sys.executable, TOOL, str(p)], capture_output=True, text=True)
        assert r.returncode == 1, (old, r.stdout)


def test_binding_text_is_whole():
    rs = open(os.path.join(ROOT, "integration", "gpu.rs")).read()
    patch = open(os.path.join(ROOT, "integration", "annotator.patch")).read()
    assert rs.count("{") == rs.count("}") and rs.count("(") == rs.count(")")
    for name in ("bb_create", "bb_create_policy", "bb_annotate_batch", "bb_destroy", "bb_counts", "bb_last_error"):
        assert re.search(r"fn " + name + r"\(", rs), name
    # the retry on BB_E_CAPACITY and the count of reads with rows are code, not comments
    assert rs.count("bb_annotate_batch(") >= 3 and "fn reads_with_rows" in rs
    assert "+use crate::annotate::gpu::" in patch and "annotate_collected" in patch and "+pub mod gpu;" in patch
    md = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    assert "integration/gpu.rs" in md and "integration/annotator.patch" in md and "/* retry */" not in md


@pytest.mark.skipif(not os.path.isdir(REF) or not shutil.which("patch"), reason="reference tree not present (GPU box) or no patch(1)")
def test_patch_applies_to_the_reference_seam(tmp_path):
    d = tmp_path / "src" / "annotate"
    d.mkdir(parents=True)
    for f in ("annotator.rs", "mod.rs"):
        shutil.copy(os.path.join(REF, f), d / f)
    r = subprocess.run(["patch", "-p1", "--dry-run", "-i", os.path.join(ROOT, "integration", "annotator.patch")], cwd=tmp_path, capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    subprocess.check_call(["patch", "-p1", "-s", "-i", os.path.join(ROOT, "integration", "annotator.patch")], cwd=tmp_path)
    new = (d / "annotator.rs").read_text()
    assert "demux(read_id" not in new and "GpuDemuxer::new(" in new and "pub mod gpu;" in (d / "mod.rs").read_text()
