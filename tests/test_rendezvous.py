"""How the processes of a `--shard R/W --rccl-id PATH` run meet (csrc/host/bb_rendezvous.cpp), driven without a GPU through
`barbell-amd rendezvous`: the files carry the run's identity — every rank's fresh nonce — so what an interrupted run left behind is never
read as this run's."""
import os
import subprocess
import time

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(ROOT, "barbell_amd", "bin", "barbell-amd")
pytestmark = pytest.mark.skipif(not os.path.exists(CLI), reason="barbell-amd not built")


def start(base, r, w, counts, delay=0.0, env=None, bus=None):
    cmd = [CLI, "rendezvous", "--rccl-id", base, "--shard", f"{r}/{w}", "--counts", ",".join(map(str, counts)), "--bus", bus or f"gpu{r}"]
    if delay:
        cmd += ["--start-delay", str(delay)]
    return subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=dict(os.environ, **(env or {})))


def finish(p, timeout=60):
    out, err = p.communicate(timeout=timeout)
    return p.returncode, out, err


def totals(out):
    return [int(x) for x in out.splitlines()[0].split()[1:]]


def leftovers(tmp_path):
    return sorted(f for f in os.listdir(tmp_path) if f.startswith("rv"))


def test_clean_run_sums_and_cleans_up(tmp_path):
    base = str(tmp_path / "rv")
    ps = [start(base, r, 3, [r + 1, 10 * (r + 1), 0]) for r in range(3)]
    res = [finish(p) for p in ps]
    assert all(rc == 0 for rc, _, _ in res), res
    assert all(totals(out) == [6, 60, 0] for _, out, _ in res)
    assert all("shared_device 0" in out for _, out, _ in res)
    assert leftovers(tmp_path) == []


def test_shared_device_is_noticed(tmp_path):
    base = str(tmp_path / "rv")
    ps = [start(base, r, 2, [5, 7], bus="0000:05:00.0@box") for r in range(2)]
    res = [finish(p) for p in ps]
    assert all(rc == 0 and totals(out) == [10, 14] and "shared_device 1" in out for rc, out, _ in res), res


def test_stale_files_of_an_interrupted_run_are_not_read(tmp_path):
    """An earlier run left every file behind (hello, info, counts, done of both ranks).  In the new run rank 1 reaches the meeting point a second
    before rank 0 has even started: it must wait for THIS run's rank 0 — not sum the old counts — and both must end with the new totals."""
    base = str(tmp_path / "rv")
    old = [start(base, r, 2, [1000, 2000], env={"BARBELL_AMD_KEEP_RENDEZVOUS": "1"}) for r in range(2)]
    assert all(finish(p)[0] == 0 for p in old)
    assert len(leftovers(tmp_path)) >= 6      # hello + info + counts of both ranks (and their done files)
    p1 = start(base, 1, 2, [3, 4])
    time.sleep(0.3)
    assert p1.poll() is None                  # waiting, not done with stale numbers
    p0 = start(base, 0, 2, [30, 40], delay=0.7)
    r0, r1 = finish(p0), finish(p1)
    assert r0[0] == 0 and r1[0] == 0, (r0, r1)
    assert totals(r0[1]) == [33, 44] and totals(r1[1]) == [33, 44]
    assert leftovers(tmp_path) == []


def test_a_rank_that_never_starts_is_a_timeout_naming_the_stale_file(tmp_path):
    base = str(tmp_path / "rv")
    old = [start(base, r, 2, [1, 2], env={"BARBELL_AMD_KEEP_RENDEZVOUS": "1"}) for r in range(2)]
    assert all(finish(p)[0] == 0 for p in old)
    t0 = time.time()
    rc, out, err = finish(start(base, 1, 2, [3, 4], env={"BARBELL_AMD_RCCL_TIMEOUT": "1.5"}))
    assert rc == 1 and 1.0 < time.time() - t0 < 30
    assert "timed out" in err and "rv.r0.info" in err and "another run's identity" in err, err


def test_stale_files_of_a_run_with_another_world_size(tmp_path):
    base = str(tmp_path / "rv")
    old = [start(base, r, 3, [1], env={"BARBELL_AMD_KEEP_RENDEZVOUS": "1"}) for r in range(3)]
    assert all(finish(p)[0] == 0 for p in old)
    p1 = start(base, 1, 2, [5])
    p0 = start(base, 0, 2, [6], delay=0.5)
    r0, r1 = finish(p0), finish(p1)
    assert r0[0] == 0 and r1[0] == 0, (r0, r1)
    assert totals(r0[1]) == [11] and totals(r1[1]) == [11]


def test_shards_with_different_queries_are_refused(tmp_path):
    base = str(tmp_path / "rv")
    ps = [start(base, 0, 2, [1, 2, 3], env={"BARBELL_AMD_RCCL_TIMEOUT": "3"}), start(base, 1, 2, [1, 2], env={"BARBELL_AMD_RCCL_TIMEOUT": "3"})]
    res = [finish(p) for p in ps]
    assert all(rc == 1 for rc, _, _ in res)
    assert any("other queries" in err or "another size" in err for _, _, err in res), res
