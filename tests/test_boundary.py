"""The drop-in boundary driven as integration/annotator.patch drives it: rows must not depend on how many reads a call
carries, on how many worker threads call at once, or on the form the sequences cross PCIe in."""
import json
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "barbell_amd", "bin", "bb-boundary-bench")
pytestmark = pytest.mark.gpu


def run(*flags, env=None):
    out = subprocess.run([BIN, *map(str, flags)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900, env=dict(os.environ, **(env or {})))
    assert out.returncode == 0, out.stderr.decode(errors="replace")[-500:]
    return json.loads(out.stdout.decode())


@pytest.mark.parametrize("kit,flags,reads,lmin,lmax", [
    ("SQK-NBD114-96", ["--flank-max-errors", 3], 49152, 4000, 4000),     # configs[1]: equal reads
    ("SQK-NBD114-96", ["--flank-max-errors", 3], 16384, 100, 30000),     # reads of differing lengths: sorted by length, the long ones cut into segments (bb_len.h)
    ("SQK-RBK114-96", ["--flank-max-errors", -1, "--use-extended"], 8192, 4000, 4000),   # configs[4]: two groups, k = 20
])
def test_rows_do_not_depend_on_batch_size_threads_or_form(kit, flags, reads, lmin, lmax):
    """One hash over all rows (batch-global read indices, order-independent) for 1 k-read calls from one and from ten threads, for the batch
    size the binding uses, for one call with every read, under the classic treatment of a batch (a round trip per decision) and with the
    sequences two bases per byte."""
    common = ["--kit", kit, *flags, "--reads", reads, "--read-len", lmax, "--read-len-min", lmin, "--check"]
    base = run(*common, "--batch", f"1024,8192,{reads}", "--threads", "1,10")
    packed = run(*common, "--batch", f"1024,{reads}", "--threads", "1,10", "--packed")
    classic = run(*common, "--batch", f"1024,{reads}", "--threads", "1,10", env={"BARBELL_AMD_DEFER_MAX": "0", "BARBELL_AMD_SMALL_PFX_MAX": "0"})
    runs = base["runs"] + packed["runs"] + classic["runs"]
    assert len(runs) == 14 and all(r["reads_checked"] == reads and r["rows"] > reads // 3 for r in runs)
    assert len({(r["rows_hash"], r["rows"]) for r in runs}) == 1, [(r["batch"], r["threads"], r["rows_hash"], r["rows"]) for r in runs]
    # what the small-batch treatment promises: one wait of the host per call (two when the rows outnumber the speculative copy)
    assert all(r["host_syncs_per_call"] <= 2.0 for r in base["runs"] if r["batch"] <= 8192)
    assert all(r["host_syncs_per_call"] >= 4.0 for r in classic["runs"])
