"""bench.py's N>1 branch on the one-GPU box: `python bench.py --gpus 2` must become two ranks by itself (one process per
GPU; here both mapped onto GPU 0 with --device-mod 1), each rank annotating its own contiguous shard of the synthetic
stream, and the all-reduced per-barcode histogram must be the sum of the two shards' single-rank histograms
(annotator.rs:278-280 is the fan-out it replaces)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
COMMON = ["--reads", "200000", "--batch", "100000", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-other-configs", "--no-e2e", "--no-policy-variants",
          "--no-stress", "--no-boundary", "--print-histogram"]


def _bench(extra, env_extra=None):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(env_extra or {})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + COMMON + extra, capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


@pytest.mark.gpu
@pytest.mark.parametrize("backend", ["gloo", "nccl"])
def test_gpus_2_starts_two_ranks(backend):
    import torch

    if backend == "nccl" and torch.cuda.device_count() < 2:
        pytest.skip("RCCL needs one device per rank; one GPU visible")
    two = _bench(["--gpus", "2", "--backend", backend] + (["--device-mod", "1"] if backend == "gloo" else []))
    assert two["n_gpus"] == 2 and two["scaling"] == "weak"
    a = _bench(["--gpus", "1"])
    b = _bench(["--gpus", "1", "--first-read", "200000"])
    assert a["n_gpus"] == 1
    want = [x + y for x, y in zip(a["histogram"], b["histogram"])]
    assert two["histogram"] == want and two["histogram_total"] == a["histogram_total"] + b["histogram_total"] > 0
    # two ranks x 2 steps x 100000 reads
    assert abs(two["value"] * two["ms_per_step"] * 1e-3 * two["steps"] - 2 * 2 * 100000) < 1.0


@pytest.mark.gpu
def test_eight_ranks_as_the_scaling_run_starts_them():
    """The size SCALE runs at: `python bench.py --gpus 8` -> eight ranks (launcher, port choice, --gpus / WORLD_SIZE checks, shards, the
    histogram reduce, the line's `dist` object with every rank's device), here on the one GPU (gloo, --device-mod 1) with the reads cut
    to fit: the all-reduced histogram is the one-rank histogram of the same 8 x 40 000 reads, and the line says eight ranks on one device."""
    small = ["--reads", "40000", "--batch", "20000"]
    eight = _bench(["--gpus", "8", "--backend", "gloo", "--device-mod", "1"] + small)
    assert eight["n_gpus"] == 8 and eight["dist"]["world"] == 8 and eight["dist"]["backend"] == "gloo"
    assert [r["rank"] for r in eight["dist"]["ranks"]] == list(range(8)) and eight["dist"]["distinct_devices"] == 1
    assert all(r["reads"] == 2 * 20000 for r in eight["dist"]["ranks"])
    one = _bench(["--gpus", "1", "--reads", "320000", "--batch", "20000", "--steps", "16"])
    assert eight["histogram"] == one["histogram"] and eight["histogram_total"] > 100000
    assert abs(eight["value"] * eight["ms_per_step"] * 1e-3 * eight["steps"] - 8 * 2 * 20000) < 1.0


@pytest.mark.gpu
def test_rccl_branch_runs_at_world_size_one():
    """`--force-dist`: the exact statements of the N>1 branch — init_process_group("nccl", device_id=dev), barrier, all_reduce of the
    histogram (sum) and of the elapsed time (max), destroy — on the one visible GPU: RCCL (librccl through torch's nccl backend) is
    initialised and runs a collective; the histogram is the plain single-process one."""
    forced = _bench(["--gpus", "1", "--force-dist"], {"HSA_ENABLE_IPC_MODE_LEGACY": "0"})
    plain = _bench(["--gpus", "1"])
    assert forced["dist"]["backend"] == "nccl" and forced["dist"]["world"] == 1 and forced["dist"]["forced_at_world_1"]
    assert "dist" not in plain
    assert forced["histogram"] == plain["histogram"] and forced["histogram_total"] > 0


def test_gpus_mismatch_is_refused():
    """A launcher that sets WORLD_SIZE but passes another --gpus must not produce a line that says n_gpus: 1."""
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4"], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode != 0 and "WORLD_SIZE" in r.stderr
