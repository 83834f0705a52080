"""Multi-GPU plumbing of the annotate path: reads shard trivially (each `demux` touches only its own
read — searcher.rs:430), so rank r of W owns a contiguous block of the read stream and there is no
data-path collective.  The only exchange is one all-reduce (sum) of the per-(group, barcode)
histogram at the end — RCCL over xGMI on the GPUs (`nccl` backend), `gloo` in the
CPU tests."""
import os

import numpy as np


def effective_cpus():
    """CPUs this process can actually keep busy: the affinity mask cut by the cgroup's CPU quota.  A container may SEE every CPU of its host
    and still be throttled to a few of them (the MI355X box of round 5: 256 visible, `cpu.max` = 1600000 100000 = 16 CPUs) — a thread pool
    sized by os.cpu_count() then runs slower than one sized by the quota (the CPU checker: 112 k reads/s on 16 threads, 51-74 k on 256)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    for path, parse in (("/sys/fs/cgroup/cpu.max", lambda t: (t.split()[0], t.split()[1])),):
        try:
            q, per = parse(open(path).read())
            if q != "max" and int(per) > 0:
                n = min(n, max(1, int(int(q) / int(per) + 0.5)))
        except (OSError, ValueError, IndexError):
            pass
    try:   # cgroup v1
        q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        if q > 0 and per > 0:
            n = min(n, max(1, int(q / per + 0.5)))
    except (OSError, ValueError):
        pass
    return max(1, n)


def shard_range(rank, world, reads_per_rank):
    """(first_read, n_reads) of rank's contiguous shard of the synthetic/real read stream (weak scaling: every rank of the
    `world` holds `reads_per_rank` reads; a rank outside the world is a launch error, not an empty shard)."""
    if not 0 <= rank < max(1, world):
        raise ValueError(f"shard_range: rank {rank} outside a world of {world}")
    return rank * reads_per_rank, reads_per_rank


def histogram_layout(groups):
    """offset of each group's counters: n_seqs tag counters then one flank-only counter"""
    offs, o = [], 0
    for g in groups:
        offs.append(o)
        o += len(g.seqs) + 1
    return offs, o


def histogram_from_rows(rows, groups):
    """What bb_counts accumulates on the device, computed from returned rows (for checks)."""
    offs, total = histogram_layout(groups)
    h = np.zeros(total, dtype=np.int64)
    for gi, g in enumerate(groups):
        rg = rows[rows["group_idx"] == gi]
        idx = np.where(rg["barcode_idx"] >= 0, rg["barcode_idx"], len(g.seqs)).astype(np.int64)
        np.add.at(h, offs[gi] + idx, 1)
    return h


def allreduce_histogram(hist_tensor):
    """Sum a torch int64 histogram over all ranks (no-op without an initialised process group)."""
    import torch.distributed as dist

    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(hist_tensor)
    return hist_tensor
