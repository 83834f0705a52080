"""Multi-GPU plumbing of the annotate path: reads shard trivially (each `demux` touches only its own
read — searcher.rs:430), so rank r of W owns a contiguous block of the read stream and there is no
data-path collective.  The only exchange is one all-reduce (sum) of the per-(group, barcode)
histogram at the end — RCCL over xGMI on the GPUs (`nccl` backend), `gloo` in the
CPU tests."""
import numpy as np


def shard_range(rank, world, reads_per_rank):
    """(first_read, n_reads) of rank's contiguous shard of the synthetic/real read stream."""
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
