"""N>1 path on CPU: two gloo ranks each take their shard of the synthetic read stream, annotate it
(CPU oracle as the stand-in backend — the GPU is not available here) and all-reduce the per-barcode
histogram; the result must equal the single-process histogram of the whole stream, and the shards
must tile the stream exactly."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N_PER_RANK = 96
SEED = 0xBA7BE11 ^ 3


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from barbell_amd import annotate as A, parallel
    from oracle import pyoracle as po
    from tests.common import config_groups

    groups = config_groups("nbd96")
    first, n = parallel.shard_range(rank, world, N_PER_RANK)
    bases, offsets = A.synth_reads_host(groups, SEED, 300, 900, first, n)
    rows = po.Oracle([g.as_tuple() for g in groups]).annotate(bases, offsets, n_threads=2)
    hist = torch.from_numpy(parallel.histogram_from_rows(rows, groups))
    parallel.allreduce_histogram(hist)
    np.save(os.path.join(out_dir, f"hist{rank}.npy"), hist.numpy())
    np.save(os.path.join(out_dir, f"bases{rank}.npy"), bases)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_histogram_reduce(tmp_path):
    world = 2
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    from barbell_amd import annotate as A, parallel
    from oracle import pyoracle as po
    from tests.common import config_groups

    groups = config_groups("nbd96")
    bases, offsets = A.synth_reads_host(groups, SEED, 300, 900, 0, world * N_PER_RANK)
    rows = po.Oracle([g.as_tuple() for g in groups]).annotate(bases, offsets, n_threads=4)
    want = parallel.histogram_from_rows(rows, groups)
    h0, h1 = np.load(tmp_path / "hist0.npy"), np.load(tmp_path / "hist1.npy")
    assert (h0 == h1).all() and (h0 == want).all() and want.sum() == len(rows) > 0
    # shards tile the stream
    b0, b1 = np.load(tmp_path / "bases0.npy"), np.load(tmp_path / "bases1.npy")
    assert np.concatenate([b0, b1]).tobytes() == bases.tobytes()


def test_shard_range_and_layout():
    from barbell_amd import parallel
    from tests.common import config_groups

    assert parallel.shard_range(3, 8, 1000) == (3000, 1000)
    offs, total = parallel.histogram_layout(config_groups("dual"))
    assert offs == [0, 97] and total == 194
