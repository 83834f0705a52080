#!/usr/bin/env python3
"""bench.py — throughput of the annotate hot path on MI355X.

Workload (BASELINE.json configs[1]): SQK-NBD114-96 (96 barcodes, 46-nt N-masked flank,
--flank-max-errors 3) on synthetic 4 kb reads.  Per GPU, `--reads` reads (default 10 M = the named
config) are generated on the device and stay resident in HBM; one "step" is one pass of the whole
hot path (flank scan -> trace -> barcode search/score -> collapse -> rows) over one batch of
`--batch` of those reads, step s taking batch s mod (reads/batch).  With N>1 every rank owns its own
shard of the read stream (weak scaling, no data-path collective); the only collective is one RCCL
all-reduce of the per-barcode histogram at the end of the timed region.

Prints ONE JSON line (rank 0).  `roofline` prices the dominant kernel against the HBM roofline with
the ALGORITHMIC bytes of SURVEY.md §8(d) (read length + 8 B offset + 48 B per row out); the path is
integer-VALU bound, so that fraction is small by construction — the `compute` object carries the
DP-cell rate that actually bounds it.  `cpu_baseline` is the CPU checker's bit-parallel path (64-bit Myers
words, OpenMP over reads) timed on this box's host cores on a bounded sample of the same reads; the
scalar restatement's rows on three windows of the batch are compared with the GPU's (bit-exact).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
from barbell_amd.parallel import effective_cpus  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

METRIC = "reads/s demultiplexed (SQK-NBD114-96, k≤5) at 1/2/4/8 MI355X; HBM GB/s vs roofline"
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec
SEED = 0xBA7BE11 ^ 2    # seed = 0xBA7BE11 ^ config_id


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--reads", type=int, default=10_000_000, help="reads resident per GPU")
    ap.add_argument("--batch", type=int, default=2_000_000, help="reads per step per GPU")
    ap.add_argument("--read-len", type=int, default=4000)
    ap.add_argument("--config", default="nbd96", choices=["nbd96", "dual", "rbk24", "rbk96x"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-configs", action="store_true")
    ap.add_argument("--no-e2e", action="store_true")
    ap.add_argument("--no-boundary", action="store_true")
    ap.add_argument("--no-stress", action="store_true")
    ap.add_argument("--no-policy-variants", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=10.0)
    ap.add_argument("--e2e-reads", type=int, default=4_000_000, help="reads of the end-to-end leg's FASTQ file (8 KB of text each)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="torch.distributed backend for N>1 (nccl = RCCL over xGMI; gloo only for single-GPU dry runs)")
    ap.add_argument("--device-mod", type=int, default=0,
                    help="dry-run aid: map LOCAL_RANK onto LOCAL_RANK %% device-mod GPUs (0 = one GPU per rank)")
    ap.add_argument("--force-dist", action="store_true",
                    help="run the N>1 statements (init_process_group, barrier, all_reduce of the histogram and of the time) at world size 1 too: "
                         "the RCCL branch executes on a one-GPU box")
    ap.add_argument("--print-histogram", action="store_true", help="carry the all-reduced per-barcode histogram in the JSON line")
    ap.add_argument("--first-read", type=int, default=0,
                    help="index of rank 0's first read in the synthetic stream (tests: a one-rank run of another rank's shard)")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # started as plain `python bench.py --gpus N`: become N ranks (one process per GPU) instead of silently running one
        import socket
        import subprocess

        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.call(cmd, env=env))

    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != max(1, args.gpus):
        sys.exit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch one rank per GPU "
                 f"(python -m torch.distributed.run --nproc-per-node {args.gpus} bench.py --gpus {args.gpus} ...)")
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    use_dist = world > 1 or args.force_dist
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if world == 1 and "MASTER_PORT" not in os.environ:      # --force-dist started as plain `python bench.py`: a rendezvous of one
            import socket

            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                os.environ["MASTER_PORT"] = str(sk.getsockname()[1])
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
    dev_idx = local_rank % args.device_mod if args.device_mod > 0 else local_rank
    torch.cuda.set_device(dev_idx)
    dev = torch.device("cuda", dev_idx)
    cdev = dev if args.backend == "nccl" else torch.device("cpu")  # where collective tensors live
    if use_dist:
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group("gloo")
        assert dist.get_world_size() == max(1, args.gpus), (dist.get_world_size(), args.gpus)

    from barbell_amd import annotate as A
    from tests.common import config_groups

    groups = config_groups(args.config)
    dm = A.Demuxer(device=dev_idx)
    for g in groups:
        dm.add_query_group(g)

    batch = min(args.batch, args.reads)
    n_batches = max(1, args.reads // batch)
    n_res = n_batches * batch
    L = args.read_len
    first_read = args.first_read + rank * n_res  # contiguous shard of the read stream per rank

    # ---- synthetic reads generated straight into HBM (fixed length -> offsets are i*L) ----
    d_off = torch.arange(0, n_res + 1, dtype=torch.int64, device=dev) * L
    d_bases = torch.empty(n_res * L, dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    dm.synth_dev(SEED, L, L, first_read, n_res, d_off.data_ptr(), d_bases.data_ptr())
    rows_cap = (4 if args.config == "nbd96" else 6) * batch
    d_rows = torch.empty(rows_cap * 48, dtype=torch.uint8, device=dev)
    # per-batch offsets must start at the batch's own base pointer
    d_off_b = d_off[: batch + 1].contiguous()
    torch.cuda.synchronize()

    def step(s):
        b = s % n_batches
        return dm.demux_dev(d_bases.data_ptr() + b * batch * L, d_off_b.data_ptr(), batch, d_rows.data_ptr(), rows_cap)

    for s in range(args.warmup):
        step(s)
    dm.counts_reset()
    dm.set_timing(True)
    kms, dom_launch = {}, {}
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    rows_total = last_rows = 0
    for s in range(args.steps):
        last_rows = step(s)
        rows_total += last_rows
        for k, v in dm.kernel_ms().items():
            kms[k] = kms.get(k, 0.0) + v
        dk_name, dk_ms = dm.dominant_kernel()
        dl = dom_launch.setdefault(dk_name, [0.0, 0])   # summed duration, number of steps in which this launch was the stage's longest
        dl[0] += dk_ms
        dl[1] += 1
    hist = torch.from_numpy(dm.counts().astype(np.int64)).to(cdev)
    rank_devices = None
    if use_dist:
        dist.all_reduce(hist)  # RCCL over xGMI: the only collective of the path
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    t1 = time.perf_counter()
    el = torch.tensor([t1 - t0], dtype=torch.float64, device=cdev)
    if use_dist:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
    elapsed = float(el.item())
    if use_dist:
        # which device every rank ran on (outside the timed region): a SCALE record can be checked for "the backend saw N ranks on N devices"
        mine = {"rank": rank, "local_rank": local_rank, "device": dev_idx, "pci_bus_id": _pci_bus_id(dev_idx), "reads": args.steps * batch}
        gathered = [None] * world
        dist.all_gather_object(gathered, mine)
        rank_devices = gathered
    reads_done = args.steps * batch * world
    value = reads_done / elapsed

    if rank == 0:
        # dominant kernel and its roofline
        kavg = {k: v / args.steps for k, v in kms.items()}
        dom = max(kavg, key=kavg.get)
        rows_per_launch = rows_total / args.steps
        alg_bytes = batch * (L + 8) + 48.0 * rows_per_launch  # SURVEY §8(d), per launch (= per batch)
        # the dominant KERNEL: the longest single launch inside the dominant stage, timed by its own pair of events on its own stream
        # (its rc twin runs alongside it on a second stream); stages other than the barcode stage are one kernel's time already
        stage_ms = kavg[dom]
        dname, (dms, dcount) = (max(dom_launch.items(), key=lambda kv: kv[1][0]) if dom_launch else ("", (0.0, 0)))
        if dom == "k_barcode" and dname and dms > 0.0:
            dom_kernel, dom_ms = dname, dms / dcount   # average over the steps in which it WAS the longest launch, not over all steps
        else:
            dom_kernel, dom_ms = dom, stage_ms
        achieved = alg_bytes / (dom_ms * 1e-3) / 1e9
        traffic, traffic_all, traffic_src = load_traffic(args, batch, L, dom, dom_kernel)
        gi = dm.group_info(0)
        cells_flank = 2.0 * sum(dm.group_info(g).flank_len for g in range(len(groups))) * L * batch
        out = {
            "metric": METRIC, "value": value, "unit": "reads/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u32", "data": "synthetic",
            "config": {"workload": f"{args.config}: SQK-NBD114-96 kit preset, {n_res} synthetic {L}-nt reads resident per GPU, "
                                   f"flank-max-errors {gi.flank_k}, one step = one {batch}-read batch per GPU (BASELINE.json configs[1])"
                       if args.config == "nbd96" else f"{args.config}: {n_res} synthetic {L}-nt reads per GPU, batch {batch}",
                       "reads_per_gpu": n_res, "batch_reads": batch, "read_len": L, "sharding": f"reads x{world}",
                       "rows_per_step": rows_per_launch},
            "roofline": {"bound": "hbm", "kernel": dom_kernel, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "algorithmic_bytes_per_launch": alg_bytes, "avg_launch_ms": dom_ms, "stage": dom, "stage_ms": stage_ms,
                         # the same bytes against the whole stage (the kernel's rc twin runs beside it on a second stream) and the whole step:
                         # the conservative readings of the same roofline
                         "frac_over_stage": alg_bytes / (stage_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                         "frac_over_step": alg_bytes / (elapsed / args.steps) / 1e9 / HBM_PEAK_GBS,
                         "traffic_all_kernels_per_step": traffic_all, "traffic_source": traffic_src},
            "kernel_ms_per_step": kavg,
            "compute": compute_section(args, kavg, cells_flank, batch, L),
            "histogram_total": int(hist.sum().item()),
        }
        if use_dist:
            out["dist"] = {"backend": dist.get_backend(), "world": dist.get_world_size(), "forced_at_world_1": bool(args.force_dist and world == 1),
                           "collectives": ["barrier", "all_reduce(histogram, sum)", "all_reduce(elapsed, max)"],
                           "ranks": rank_devices, "distinct_devices": len({(r or {}).get("pci_bus_id") for r in rank_devices or []})}
        if args.print_histogram:
            out["histogram"] = [int(x) for x in hist.cpu().tolist()]
        if args.config == "nbd96":
            try:
                out["filter_step"], d_v = filter_leg(dm, d_rows, int(rows_per_launch), dev)
                out["trim_step"] = trim_leg(dm, d_rows, d_v, int(rows_per_launch), d_bases.data_ptr() + ((args.steps - 1) % n_batches) * batch * L,
                                            d_off_b, batch, L, dev)
                out["ingest_step"] = ingest_leg(dm, d_bases.data_ptr() + ((args.steps - 1) % n_batches) * batch * L, batch, L, dev)
            except Exception as e:  # noqa: BLE001  (legs outside the timed region never cost the line)
                out["widened_steps_error"] = f"{type(e).__name__}: {e}"[:400]
        if world == 1 and args.config == "nbd96" and not args.no_other_configs:
            # BASELINE configs[3] / configs[4] (driver-run numbers for the other query geometries; `value` stays configs[1])
            out["other_configs"] = {c: _guarded(other_config_leg, c, dev_idx, dev, L, args) for c in ("dual", "rbk96x")}
        if world == 1 and not args.no_other_configs:
            # the same steps with the batch halved between TWO contexts of this GPU (two host threads, a stream pair each, as the product host's --streams 2):
            # one half's scan runs into the other half's barcode stage and the launches' tails fill each other; outside `value`, whose step is one context's
            out["two_contexts"] = _guarded(two_contexts_leg, args, groups, dev_idx, d_bases, batch, n_batches, L, dev, rows_per_launch)
        if world == 1 and args.config == "nbd96" and not args.no_stress:
            # the same pipeline where the filtered scan's assumption (unrelated text rarely comes within k edits of a flank window)
            # is strained; outside `value`
            out["stress"] = {name: _guarded(stress_leg, mode, dev_idx, dev, L, args) for name, mode in (("low_complexity_30pct", 1), ("prefix_decoys_every_200nt", 2), ("artefacts_50pct", 3), ("prefix_decoys_every_60nt", 4))}
        if world == 1 and args.config == "nbd96" and not args.no_stress:
            # read lengths as a real run's (heavy-tailed) against equal reads of the same mean: what the benchmark's fixed 4 kb does not show
            out["length_mix"] = {c: _guarded(length_mix_leg, c, dev_idx, dev, args) for c in ("nbd96", "rbk96x")}
        if world == 1 and args.config == "nbd96" and not args.no_policy_variants:
            # the same workload under every setting of the assumptions about sassy / cigar-lodhi-rs that Barbell's own code does not pin
            # (include/barbell_amd_policy.h): what the headline becomes if the real crates turn out to differ from the default; outside `value`
            out["policy_variants"] = _guarded(policy_variants_leg, dev_idx, dev, L, args)
        if world == 1 and args.config == "nbd96" and not args.no_e2e:
            out["e2e_step"] = _guarded(e2e_leg, d_bases, min(n_res, args.e2e_reads), L, dev)
        if world == 1 and args.config == "nbd96" and not args.no_boundary:
            # the C ABI as the reference-side binding drives it: worker threads x own context x bb_annotate_batch on pageable host memory
            # (tools/boundary_rate.py, csrc/host/boundary_bench.cpp); the resident reads are released first: the harness is a process of its own
            out["boundary_step"] = _guarded(boundary_leg)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = _guarded(cpu_baseline, args, groups, dm, d_bases, L, batch, (args.steps - 1) % n_batches, d_rows, last_rows)
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


def _pci_bus_id(dev_idx):
    try:
        p = torch.cuda.get_device_properties(dev_idx)
        return f"{p.pci_domain_id:04x}:{p.pci_bus_id:02x}:{p.pci_device_id:02x}" if hasattr(p, "pci_bus_id") else str(dev_idx)
    except Exception:  # noqa: BLE001
        return str(dev_idx)


def two_contexts_leg(args, groups, dev_idx, d_bases, batch, n_batches, L, dev, rows_per_launch, parts=2):
    """The timed loop of `value` again, every step's batch split between `parts` contexts that run side by side (a host thread each: the C ABI's calls
    block, as the reference's worker threads do — one Demuxer each, annotator.rs:88-101).  Nothing else differs: same resident reads, same steps."""
    import threading

    from barbell_amd import annotate as A

    n = batch // parts
    dms = []
    for _ in range(parts):
        dm = A.Demuxer(device=dev_idx)
        for g in groups:
            dm.add_query_group(g)
        dms.append(dm)
    d_off = torch.arange(0, n + 1, dtype=torch.int64, device=dev) * L
    cap = 4 * n if args.config == "nbd96" else 6 * n
    rows = [torch.empty(cap * 48, dtype=torch.uint8, device=dev) for _ in range(parts)]
    got = [0] * parts

    def work(i, first, k):
        for s in range(first, first + k):
            b = s % n_batches
            got[i] = dms[i].demux_dev(d_bases.data_ptr() + (b * batch + i * n) * L, d_off.data_ptr(), n, rows[i].data_ptr(), cap)

    def run(first, k):
        th = [threading.Thread(target=work, args=(i, first, k)) for i in range(parts)]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for t in th:
            t.start()
        for t in th:
            t.join()
        torch.cuda.synchronize()
        return time.perf_counter() - t0

    run(0, max(1, args.warmup))
    dt = run(0, args.steps)
    out = {"contexts": parts, "reads_per_s": args.steps * n * parts / dt, "ms_per_step": dt / args.steps * 1e3, "steps": args.steps, "batch_reads": n * parts,
           "rows_last_step": int(sum(got)), "note": "each step's batch halved between two contexts on one GPU, a host thread each; `value` is one context's"}
    for dm in dms:
        dm.close()
    return out


def other_config_leg(cfg, dev_idx, dev, L, args, n=2_000_000, steps=3):   # (n = BASELINE's step size since round 6: 1 M-read steps ran the same kernels 10-15 % below it)
    """One of the other BASELINE query sets on its own resident reads: `dual` = configs[3] (native_left/right.fasta,
    --flank-max-errors 5, two groups), `rbk96x` = configs[4] made meaningful (SQK-RBK114-96 --use-extended: two groups,
    90-nt flanks, automatic cutoff; --use-extended is a no-op for SQK-NBD114-96).  n reads x L resident, one batch,
    `steps` timed passes; a sample of the rows is compared with the oracle."""
    from barbell_amd import _abi
    from barbell_amd import annotate as A
    from oracle import pyoracle as po
    from tests.common import config_groups

    groups = config_groups(cfg)
    dm = A.Demuxer(device=dev_idx)
    for g in groups:
        dm.add_query_group(g)
    seed = 0xBA7BE11 ^ {"dual": 4, "rbk96x": 5}[cfg]
    d_off = torch.arange(0, n + 1, dtype=torch.int64, device=dev) * L
    d_bases = torch.empty(n * L, dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    dm.synth_dev(seed, L, L, 0, n, d_off.data_ptr(), d_bases.data_ptr())
    cap = 6 * n
    d_rows = torch.empty(cap * 48, dtype=torch.uint8, device=dev)
    nr = dm.demux_dev(d_bases.data_ptr(), d_off.data_ptr(), n, d_rows.data_ptr(), cap)  # warm-up
    dm.set_timing(True)
    kms = {}
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        nr = dm.demux_dev(d_bases.data_ptr(), d_off.data_ptr(), n, d_rows.data_ptr(), cap)
        for k, v in dm.kernel_ms().items():
            kms[k] = kms.get(k, 0.0) + v / steps
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    dom = max(kms, key=kms.get)
    out = {"reads_per_s": n / dt, "ms_per_step": dt * 1e3, "reads": n, "read_len": L, "steps": steps, "rows_per_step": nr, "groups": len(groups),
           "dominant_kernel": dom, "dominant_kernel_ms": kms[dom], "kernel_ms_per_step": kms}
    if not args.no_cpu_baseline:
        cores = effective_cpus()
        w = 4096 if cfg == "dual" else 2048
        full = np.frombuffer(d_rows[: nr * 48].cpu().numpy().tobytes(), dtype=_abi.ROW_DTYPE)
        orc = po.Oracle([g.as_tuple() for g in groups])
        ok, rows = True, 0
        for first in (0, n - w):
            want = orc.annotate(d_bases[first * L: (first + w) * L].cpu().numpy(), np.arange(w + 1, dtype=np.uint64) * np.uint64(L), n_threads=cores)
            got = full[(full["read_idx"] >= first) & (full["read_idx"] < first + w)].copy()
            got["read_idx"] -= first
            ok = ok and got.tobytes() == want.tobytes()
            rows += len(want)
        out["parity_on_sample"] = bool(ok)
        out["sample"] = f"first and last {w} reads of the batch against the CPU oracle ({rows} rows)"
    dm.close()
    return out


def policy_variants_leg(dev_idx, dev, L, args, n=1_000_000, steps=3):
    """SQK-NBD114-96, n resident reads, `steps` timed passes per policy, the median reported (tests/common.py::GPU_POLICIES cut to what the reference's own
    vectors leave open — tests/golden/policy_feasible.json: every open alternative of every hazard alone, the five feasible traceback classes among them,
    and two mixtures): reads/s, the barcode stage's time, which kernel decided
    the hits, and the rows of the first reads against the CPU checker under the same policy.  `min_vs_default` is over the settings Barbell's
    own code leaves open; Lodhi's p and lambda are pinned by searcher.rs:209 (Lodhi::new(3, 0.5)) and only listed."""
    from barbell_amd import _abi
    from barbell_amd import annotate as A
    from oracle import pyoracle as po
    from tests.common import GPU_POLICIES, config_groups, split_feasible

    # the settings the reference's own vectors and examples leave open (tests/golden/policy_feasible.json: cigar_parse.rs:163-176 refutes 13 of the 18
    # traceback classes, the documented dual-end example refutes rcpath=mirror); the refuted ones are listed, not run
    feasible, refuted = split_feasible(GPU_POLICIES)
    groups = config_groups("nbd96")
    d_off = torch.arange(0, n + 1, dtype=torch.int64, device=dev) * L
    d_bases = torch.empty(n * L, dtype=torch.uint8, device=dev)
    cap = 6 * n
    d_rows = torch.empty(cap * 48, dtype=torch.uint8, device=dev)
    w = 2048
    sample = None
    res, base = {}, None
    for pol in ["default"] + feasible:
        ptxt = "" if pol == "default" else pol
        dm = A.Demuxer(device=dev_idx, policy=ptxt)
        for g in groups:
            dm.add_query_group(g)
        if sample is None:
            torch.cuda.synchronize()
            dm.synth_dev(SEED, L, L, 0, n, d_off.data_ptr(), d_bases.data_ptr())
            sample = d_bases[: w * L].cpu().numpy()
        nr = dm.demux_dev(d_bases.data_ptr(), d_off.data_ptr(), n, d_rows.data_ptr(), cap)
        dm.set_timing(True)
        kms, dts = {}, []
        for _ in range(steps):  # the median step: three steps per policy, and one hiccup (a code object paged in, a clock ramp) would be a third of a mean
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            nr = dm.demux_dev(d_bases.data_ptr(), d_off.data_ptr(), n, d_rows.data_ptr(), cap)
            dts.append(time.perf_counter() - t0)
            for k, v in dm.kernel_ms().items():
                kms.setdefault(k, []).append(v)
        dt = sorted(dts)[len(dts) // 2]
        kms = {k: sorted(v)[len(v) // 2] for k, v in kms.items()}
        p = _abi.policy_from_str(ptxt)
        pinned = p.lodhi_p != 3 or p.lodhi_lambda != 0.5
        st = [dm.barcode_stats(0, s) for s in (0, 1)]
        e = {"reads_per_s": n / dt, "ms_per_step": dt * 1e3, "barcode_stage_ms": kms["k_barcode"], "rows_per_step": nr,
             "kernel": "k_barcode (any-policy)" if pinned else ("k_barcode_lane" if all(s["lane_kernel"] for s in st) else "k_barcode_pfx"),
             "undecided_hits_frac": sum(s["undecided"] for s in st) / max(1, sum(s["hits"] for s in st))}
        if pinned:
            e["pinned_by_barbell"] = "searcher.rs:209 constructs Lodhi::new(3, 0.5): not an open assumption; listed, not in min_vs_default"
        if not args.no_cpu_baseline:
            full = np.frombuffer(d_rows[: nr * 48].cpu().numpy().tobytes(), dtype=_abi.ROW_DTYPE)
            want = po.Oracle([g.as_tuple() for g in groups], policy=ptxt or None).annotate(sample, np.arange(w + 1, dtype=np.uint64) * np.uint64(L),
                                                                                    n_threads=effective_cpus(), fast=True)
            e["parity_on_sample"] = bool(full[full["read_idx"] < w].tobytes() == want.tobytes())
        dm.close()
        if pol == "default":
            base = e["reads_per_s"]
        e["vs_default"] = e["reads_per_s"] / base
        res[pol] = e
    open_ = {k: v for k, v in res.items() if "pinned_by_barbell" not in v}
    worst = min(open_, key=lambda k: open_[k]["vs_default"])
    return {"reads": n, "read_len": L, "steps": steps, "sample": f"first {w} reads against the CPU checker under the same policy", "n_policies": len(res),
            "min_vs_default": open_[worst]["vs_default"], "min_policy": worst, "all_parity": all(v.get("parity_on_sample", True) for v in res.values()),
            "range": "the settings tests/golden/policy_feasible.json leaves open (tools/policy_feasible.py)", "refuted_by_reference_not_run": refuted,
            "policies": res}


def stress_leg(mode, dev_idx, dev, L, args, n=1_000_000, steps=3):
    """SQK-NBD114-96 on reads whose bodies / read mix strain the filtered flank scan (bb_synth.h: the seed's top byte selects the mix):
    reads/s, how much of the batch the filter flagged, which scan ran (bb_last_scan_stats), the scan stage's time, and a parity sample."""
    from barbell_amd import _abi
    from barbell_amd import annotate as A
    from oracle import pyoracle as po
    from tests.common import config_groups

    groups = config_groups("nbd96")
    dm = A.Demuxer(device=dev_idx)
    for g in groups:
        dm.add_query_group(g)
    seed = (mode << 56) | (0xBA7BE11 ^ 2)
    d_off = torch.arange(0, n + 1, dtype=torch.int64, device=dev) * L
    d_bases = torch.empty(n * L, dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    dm.synth_dev(seed, L, L, 0, n, d_off.data_ptr(), d_bases.data_ptr())
    cap = 6 * n
    d_rows = torch.empty(cap * 48, dtype=torch.uint8, device=dev)
    for _ in range(3):   # untimed: the batch that learns what this mix is like (filter flags, undecided hits), the probe batch of a back-off, one more
        nr = dm.demux_dev(d_bases.data_ptr(), d_off.data_ptr(), n, d_rows.data_ptr(), cap)
    dm.set_timing(True)
    kms = {}
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        nr = dm.demux_dev(d_bases.data_ptr(), d_off.data_ptr(), n, d_rows.data_ptr(), cap)
        for k, v in dm.kernel_ms().items():
            kms[k] = kms.get(k, 0.0) + v / steps
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    st = dm.scan_stats(0)
    out = {"reads_per_s": n / dt, "ms_per_step": dt * 1e3, "reads": n, "rows_per_step": nr, "flagged_fraction": st["flagged_pieces"] / max(1, st["total_pieces"]),
           "scan": ("full scan", "filter + verification", "filter, then full scan (flags above break-even)",
                    "full scan (the group's last probed batch flagged above break-even: no filter pass for 16 batches)")[st["kind"]], "scan_stage_ms": kms["k_flank_scan"],
           "barcode_stage_ms": kms["k_barcode"]}
    if not args.no_cpu_baseline:
        w = 2048
        full = np.frombuffer(d_rows[: nr * 48].cpu().numpy().tobytes(), dtype=_abi.ROW_DTYPE)
        want = po.Oracle([g.as_tuple() for g in groups]).annotate(d_bases[: w * L].cpu().numpy(), np.arange(w + 1, dtype=np.uint64) * np.uint64(L),
                                                                  n_threads=effective_cpus(), fast=True)
        out["parity_on_sample"] = bool(full[full["read_idx"] < w].tobytes() == want.tobytes())
    dm.close()
    return out


def length_mix_leg(cfg, dev_idx, dev, args, n=1_000_000, steps=3):
    """Reads whose lengths differ as a run's do (tests/common.py::heavy_tailed_batch: 200 nt .. 120 kb, 70 % under 3 kb) against the same number of
    bases in equal reads: bases/s of the whole step and the scan stage's time for both.  The scans give a lane one read; where the lengths
    differ the lanes take segments of reads by falling length instead (bb_len.h) — without that the heavy-tailed batch ran the scan stage at
    13 x (SQK-NBD114-96) the time of the equal reads (round 5, tools/ragged_probe.py).  Host-generated reads, uploaded once; outside `value`.
    1 M reads since round 6 (100 k until then: a batch that small is half fixed costs, and DESIGN quoted a larger mix the driver never ran)."""
    from barbell_amd import _abi
    from barbell_amd import annotate as A
    from oracle import pyoracle as po
    from tests.common import config_groups, heavy_tailed_batch

    groups = config_groups(cfg)
    hb, ho = heavy_tailed_batch(groups, n)
    mean = int(len(hb) / (len(ho) - 1))
    fb, fo = A.synth_reads_host(groups, 9, mean, mean, 0, len(ho) - 1)
    out = {"reads": len(ho) - 1, "mean_len": mean, "max_len": int((ho[1:] - ho[:-1]).max())}
    for name, (b, o) in (("equal_reads", (fb, fo)), ("heavy_tailed", (hb, ho))):
        dm = A.Demuxer(device=dev_idx)
        for g in groups:
            dm.add_query_group(g)
        nr_ = len(o) - 1
        d_b = torch.from_numpy(b).to(dev)
        d_o = torch.from_numpy(o.astype(np.int64)).to(dev)
        cap = 8 * nr_
        d_rows = torch.empty(cap * 48, dtype=torch.uint8, device=dev)
        nr = dm.demux_dev(d_b.data_ptr(), d_o.data_ptr(), nr_, d_rows.data_ptr(), cap)
        dm.set_timing(True)
        kms = {}
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            nr = dm.demux_dev(d_b.data_ptr(), d_o.data_ptr(), nr_, d_rows.data_ptr(), cap)
            for k, v in dm.kernel_ms().items():
                kms[k] = kms.get(k, 0.0) + v / steps
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        ls = dm.length_stats()
        out[name] = {"gbases_per_s": len(b) / dt / 1e9, "reads_per_s": nr_ / dt, "ms_per_step": dt * 1e3, "scan_stage_ms": kms["k_flank_scan"],
                     "barcode_stage_ms": kms["k_barcode"], "rows_per_step": nr, "work_items": ls["work_items"], "lines_min_max": [ls["min_lines"], ls["max_lines"]]}
        if name == "heavy_tailed" and not args.no_cpu_baseline:
            w = 1024
            full = np.frombuffer(d_rows[: nr * 48].cpu().numpy().tobytes(), dtype=_abi.ROW_DTYPE)
            want = po.Oracle([g.as_tuple() for g in groups]).annotate(b[: int(o[w])], o[: w + 1], n_threads=effective_cpus(), fast=True)
            out[name]["parity_on_sample"] = bool(full[full["read_idx"] < w].tobytes() == want.tobytes())
        dm.close()
        del d_b, d_o, d_rows
    out["heavy_tailed_vs_equal"] = out["heavy_tailed"]["gbases_per_s"] / out["equal_reads"]["gbases_per_s"]
    return out


def boundary_leg():
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import boundary_rate

    return boundary_rate.measure()


def _guarded(leg, *a):
    """the legs outside the timed region never cost the line: a failure (a full /tmp, a missing binary) is reported in their place"""
    try:
        return leg(*a)
    except Exception as e:  # noqa: BLE001
        return {"error": f"{type(e).__name__}: {e}"[:400]}


def e2e_leg(d_bases, n, L, dev):
    """FASTQ file -> annotation.tsv through the C++ host (barbell-amd annotate): the first n resident reads are written
    as a FASTQ file (page cache), then the CLI runs with two contexts on this GPU.  Reported: the CLI's steady-state rate
    (first block requested .. last block committed; file reads, PCIe upload, GPU parse + annotate + TSV rendering, file
    writes — no process start) and the wall clock of the whole process, for both upload forms (round 5: header lines + two bases per
    byte by default, `--no-pack`: sequence lines as text) with the TSVs compared; `barbell-amd kit` on the same file; the host side alone;
    and the process wall on 16 M reads (129 GB of FASTQ in /dev/shm).  Never `value`: host- and PCIe-bound (8 KB of text read per 4 kb read)."""
    import re
    import subprocess
    import tempfile

    cli = os.path.join(ROOT, "barbell_amd", "bin", "barbell-amd")
    if not os.path.exists(cli):
        return {"error": "barbell_amd/bin/barbell-amd not built"}
    n_res = d_bases.numel() // L

    def write_fastq(path, count):
        """`count` records of the resident reads (wrapping around them, ids keep counting) as 4-line FASTQ text"""
        with open(path, "wb") as f:
            step = 250_000
            for first in range(0, count, step):
                m = min(step, count - first)
                hdr = np.tile(np.frombuffer(b"@r00000000 ch=0000 st=2024-01-01T00:00Z\n", dtype=np.uint8), (m, 1))
                idx = np.arange(first, first + m)
                for d in range(8):
                    hdr[:, 9 - d] = 48 + (idx // 10 ** d) % 10
                src0 = first % n_res
                if src0 + m <= n_res:
                    seq = d_bases[src0 * L: (src0 + m) * L].view(m, L)
                else:
                    seq = torch.cat([d_bases[src0 * L:], d_bases[: (src0 + m - n_res) * L]]).view(m, L)
                q = torch.full((m, L), 53, dtype=torch.uint8, device=dev)
                sep = torch.tensor(list(b"\n+\n"), dtype=torch.uint8, device=dev).repeat(m, 1)
                nl = torch.full((m, 1), 10, dtype=torch.uint8, device=dev)
                text = torch.cat([torch.from_numpy(hdr).to(dev), seq, sep, q, nl], dim=1).contiguous().view(-1)
                text.cpu().numpy().tofile(f)

    with tempfile.TemporaryDirectory() as td:
        fq = os.path.join(td, "e2e.fastq")
        t0 = time.perf_counter()
        write_fastq(fq, n)
        gen_s = time.perf_counter() - t0
        size = os.path.getsize(fq)
        env = dict(os.environ, BARBELL_AMD_NO_TORCH="1")
        t0 = time.perf_counter()
        cmd = [cli, "annotate", "-i", fq, "-o", os.path.join(td, "a.tsv"), "--kit", "SQK-NBD114-96", "--flank-max-errors", "3", "--streams", "2",
               "--block-bytes", str(256 << 20), "-t", "32"]
        r = subprocess.run(cmd, capture_output=True, text=True, env=env)
        wall = time.perf_counter() - t0
        # two more runs of the same command: a 4 M-read run lasts a third of a second, one sample says little (the reported figures are the
        # medians; every run is listed)
        reps = [(wall, r)]
        for _ in range(2):
            t0 = time.perf_counter()
            rr = subprocess.run(cmd, capture_output=True, text=True, env=env)
            reps.append((time.perf_counter() - t0, rr))
        # the host side alone: files -> reader threads -> blocks of whole records in the upload buffers, no GPU call.  What one host process
        # can feed: the ceiling of a node's end-to-end rate however many GPUs take the blocks
        feed = {}
        try:
            rf = subprocess.run(cmd[:5] + [os.path.join(td, "feed.tsv")] + cmd[6:-6] + ["--streams", "4", "--block-bytes", str(128 << 20), "-t", "64"], capture_output=True, text=True,
                                env=dict(env, BARBELL_AMD_FEED_ONLY="1"))   # (its own output path: it must not truncate a.tsv)
            mf = re.search(r"feed-only: (\d+) bytes of staged text in ([\d.]+) s", rf.stderr)
            if rf.returncode == 0 and mf:
                feed = {"reads_per_s": n / float(mf.group(2)), "fastq_gb_per_s": size / float(mf.group(2)) / 1e9, "seconds": float(mf.group(2)), "reader_threads": 64,
                        "note": "BARBELL_AMD_FEED_ONLY=1: no upload, no kernels; one process, page cache"}
        except Exception as e:  # noqa: BLE001
            feed = {"error": str(e)[:200]}
        if r.returncode != 0:
            return {"error": r.stderr[-400:]}
        m = re.search(r"Done: (\d+) records, (\d+) with annotations, (\d+) rows .*\(([\d.]+) s in the pipeline", r.stderr)
        pipes, walls = [], []
        for w_, r_ in reps:
            m_ = re.search(r"\(([\d.]+) s in the pipeline", r_.stderr)
            if r_.returncode == 0 and m_:
                pipes.append(float(m_.group(1)))
                walls.append(w_)
        pipe, wall = sorted(pipes)[len(pipes) // 2], sorted(walls)[len(walls) // 2]
        # the same run with the sequence lines uploaded as text (--no-pack: round 4's form, 4 KB per read over PCIe instead of 2): both forms
        # reported, annotation.tsv must be the same bytes
        text_form = {}
        try:
            import hashlib

            sha = lambda path: hashlib.sha256(open(path, "rb").read()).hexdigest()
            t0 = time.perf_counter()
            r2 = subprocess.run(cmd[:5] + [os.path.join(td, "b.tsv")] + cmd[6:] + ["--no-pack"], capture_output=True, text=True, env=env)
            wall2 = time.perf_counter() - t0
            m2 = re.search(r"\(([\d.]+) s in the pipeline", r2.stderr)
            if r2.returncode == 0 and m2:
                text_form = {"steady_state_reads_per_s": n / float(m2.group(1)), "pipeline_s": float(m2.group(1)), "process_wall_s": wall2,
                             "process_wall_reads_per_s": n / wall2, "tsv_identical_to_packed": sha(os.path.join(td, "a.tsv")) == sha(os.path.join(td, "b.tsv"))}
            else:
                text_form = {"error": r2.stderr[-300:]}
        except Exception as e:  # noqa: BLE001
            text_form = {"error": str(e)[:200]}
        # `barbell-amd kit` on the same file: annotate + inspect + filter + trim, ~6.5 KB of per-barcode FASTQ written per read
        kit = {}
        try:
            import shutil

            free = shutil.disk_usage(td).free
            if free < 1.1 * size:  # the per-barcode files are ~0.8 x the FASTQ
                raise RuntimeError(f"skipped: {free >> 30} GiB free in {td}, the kit run writes ~{int(0.8 * size) >> 30} GiB")
            t0 = time.perf_counter()
            rk = subprocess.run([cli, "kit", "-k", "SQK-NBD114-96", "-i", fq, "-o", os.path.join(td, "kit"), "--flank-max-errors", "3", "--maximize",
                                 "--streams", "3", "-t", "32"], capture_output=True, text=True, env=env)
            kwall = time.perf_counter() - t0
            mk = re.search(r"\(([\d.]+) s in the pipeline", rk.stderr)
            if rk.returncode == 0 and mk:
                out_bytes = sum(os.path.getsize(os.path.join(td, "kit", x)) for x in os.listdir(os.path.join(td, "kit")))
                kit = {"steady_state_reads_per_s": n / float(mk.group(1)), "pipeline_s": float(mk.group(1)), "process_wall_s": kwall, "output_bytes": out_bytes,
                       "command": "barbell-amd kit -k SQK-NBD114-96 --flank-max-errors 3 --maximize --streams 3 -t 32",
                       "note": "the GPU plans the trim, the file writers cut the records out of the staged text (bb_trim_plan_dev); varies with the box's page-cache write-back"}
            else:
                kit = {"error": rk.stderr[-300:]}
        except Exception as e:  # the kit run is an extra, never a reason to lose the line
            kit = {"error": str(e)[:200]}
        # gzip input as a run's files concatenated (`cat run/*.fastq.gz`): the first 200 k reads as 64 gzip members in one file — bound by zlib
        # (0.3 GB/s of text per core; the reference's reader has the same limit), the members inflated side by side against one after the other
        gzm = {}
        try:
            import zlib
            from multiprocessing.pool import ThreadPool

            n_gz, n_mem = min(n, 200_000), 64
            rec = 2 * L + 64                 # upper bound of a record's bytes; the text is cut at record starts found in the file itself
            with open(fq, "rb") as f:
                head = f.read(n_gz * rec)
            ends_, pos_ = [0], 0
            per = n_gz // n_mem
            for _ in range(n_mem):
                for _ in range(per * 4):
                    pos_ = head.index(b"\n", pos_) + 1
                ends_.append(pos_)
            def _member(ab):
                c = zlib.compressobj(1, zlib.DEFLATED, 31)
                return c.compress(head[ab[0]:ab[1]]) + c.flush()
            t0 = time.perf_counter()
            with ThreadPool(effective_cpus()) as tp:    # (zlib releases the GIL)
                blobs = tp.map(_member, list(zip(ends_[:-1], ends_[1:])))
            gzp = os.path.join(td, "members.fastq.gz")
            with open(gzp, "wb") as f:
                for b_ in blobs:
                    f.write(b_)
            gzm = {"reads": per * n_mem, "members": n_mem, "text_bytes": ends_[-1], "compressed_bytes": os.path.getsize(gzp), "compress_s": time.perf_counter() - t0}
            del head, blobs
            tsvs = []
            for key, extra_env in (("members_side_by_side", {}), ("one_after_the_other", {"BARBELL_AMD_GZ_SERIAL": "1"})):
                t0 = time.perf_counter()
                rg = subprocess.run([cli, "annotate", "-i", gzp, "-o", os.path.join(td, key + ".tsv"), "--kit", "SQK-NBD114-96", "--flank-max-errors", "3", "-t", "32"],
                                    capture_output=True, text=True, env=dict(env, **extra_env))
                w_ = time.perf_counter() - t0
                if rg.returncode != 0:
                    gzm[key] = {"error": rg.stderr[-300:]}
                    continue
                gzm[key] = {"process_wall_s": w_, "reads_per_s": per * n_mem / w_, "text_gb_per_s": ends_[-1] / w_ / 1e9}
                tsvs.append(open(os.path.join(td, key + ".tsv"), "rb").read())
            gzm["tsv_identical"] = len(tsvs) == 2 and tsvs[0] == tsvs[1]
            gzm["note"] = "process wall incl. ~0.3 s of start-up; -t 32 on the container's CPU quota"
            os.remove(gzp)
        except Exception as e:  # noqa: BLE001
            gzm = {"error": f"{type(e).__name__}: {e}"[:300]}
        # the process wall on an input long enough for start-up (~0.35 s: loader, HIP, two contexts) not to dominate: 16 M reads = 129 GB of FASTQ,
        # in /dev/shm (the scratch directory's file system is smaller than that); skipped where /dev/shm cannot hold it
        big = {}
        try:
            import shutil

            n_big = 16_000_000
            if os.path.isdir("/dev/shm") and shutil.disk_usage("/dev/shm").free > 3 * n_big * (2 * L + 48):
                fq_big = "/dev/shm/barbell_amd_e2e_16m.fastq"
                try:
                    t0 = time.perf_counter()
                    write_fastq(fq_big, n_big)
                    big["fastq_write_s"] = time.perf_counter() - t0
                    big["fastq_bytes"] = os.path.getsize(fq_big)
                    runs = []
                    for _ in range(3):
                        t0 = time.perf_counter()
                        rb = subprocess.run(cmd[:3] + [fq_big, "-o", "/dev/shm/barbell_amd_e2e_16m.tsv"] + cmd[6:], capture_output=True, text=True, env=env)
                        wb = time.perf_counter() - t0
                        mb = re.search(r"\(([\d.]+) s in the pipeline", rb.stderr)
                        if rb.returncode != 0 or not mb:
                            raise RuntimeError(rb.stderr[-300:])
                        runs.append({"process_wall_s": wb, "process_wall_reads_per_s": n_big / wb, "pipeline_s": float(mb.group(1)),
                                     "steady_state_reads_per_s": n_big / float(mb.group(1))})
                    best = min(runs, key=lambda x: x["process_wall_s"])
                    big.update({"reads": n_big, "runs": runs, "process_wall_s": best["process_wall_s"], "process_wall_reads_per_s": best["process_wall_reads_per_s"],
                                "steady_state_reads_per_s": max(x["steady_state_reads_per_s"] for x in runs), "where": "/dev/shm",
                                "note": "best of three runs of the same command (the first finds the file's pages cold in this process's view)"})
                finally:
                    for pth in (fq_big, "/dev/shm/barbell_amd_e2e_16m.tsv"):
                        if os.path.exists(pth):
                            os.remove(pth)
            else:
                big = {"skipped": "/dev/shm cannot hold 129 GB"}
        except Exception as e:  # noqa: BLE001
            big = {"error": str(e)[:300]}
        short = {"reads": n, "fastq_bytes": size, "steady_state_reads_per_s": n / pipe, "steady_state_fastq_gb_per_s": size / pipe / 1e9, "pipeline_s": pipe,
                 "process_wall_s": wall, "process_wall_reads_per_s": n / wall, "runs": [{"pipeline_s": p_, "process_wall_s": w_} for p_, w_ in zip(pipes, walls)],
                 "reported": "median of the runs"}
        # The figures at the top of this object come from the LONG input where it could be run (16 M reads: 1.1-1.4 s of pipeline; the 4 M-read
        # runs last a third of a second — one 0.05 s hiccup is 15 % — and are kept under `short_input`); medians of three runs either way.
        top = short
        if big.get("runs"):
            med = lambda key: sorted(x[key] for x in big["runs"])[len(big["runs"]) // 2]
            top = {"reads": big["reads"], "fastq_bytes": big["fastq_bytes"], "pipeline_s": med("pipeline_s"), "steady_state_reads_per_s": big["reads"] / med("pipeline_s"),
                   "steady_state_fastq_gb_per_s": big["fastq_bytes"] / med("pipeline_s") / 1e9, "process_wall_s": med("process_wall_s"),
                   "process_wall_reads_per_s": big["reads"] / med("process_wall_s"), "runs": big["runs"], "reported": "median of the runs on the 16 M-read input"}
        return {**top, "short_input": short, "kit": kit, "gzip_members": gzm, "wall_16m_reads": big, "tsv_bytes": os.path.getsize(os.path.join(td, "a.tsv")), "rows": int(m.group(3)),
                "fastq_write_s": gen_s,
                "host_feed_only": feed, "upload_form": "packed: header lines + two bases per byte (BB_FASTQ_PACKED), ~2 KB per read over PCIe",
                "text_lines_form": text_form,
                "command": "barbell-amd annotate --kit SQK-NBD114-96 --flank-max-errors 3 --streams 2 --block-bytes 256Mi -t 32",
                "note": "C++ host, FASTQ text from the page cache to annotation.tsv (8 KB of text per read read by the host, ~2 KB uploaded); not the headline value"}


def filter_leg(dm, d_rows, n_rows, dev):
    """SURVEY §8(f-1), outside the timed region: the kit's default filter patterns (kits.rs:175-236,
    --maximize set) applied to the rows of the last step while they sit in HBM.  16 B verdict per 48 B row."""
    from barbell_amd.filter import Filter, kit_patterns

    flt = Filter(dm, kit_patterns("SQK-NBD114-96", True))
    d_v = torch.empty(n_rows * 16, dtype=torch.uint8, device=dev)
    flt.verdicts_dev(d_rows.data_ptr(), n_rows, d_v.data_ptr())
    reps = 10
    t0 = time.perf_counter()
    for _ in range(reps):
        flt.verdicts_dev(d_rows.data_ptr(), n_rows, d_v.data_ptr())  # synchronous on the context's stream
    ms = (time.perf_counter() - t0) / reps * 1e3
    v = d_v.view(-1, 16)
    first = v[:, 2:4].view(torch.int16).flatten() == 0          # match_idx == 0: first row of a read
    return {"patterns": len(flt.patterns), "rows": n_rows, "ms_per_step": ms, "gb_per_s": n_rows * 64 / (ms * 1e-3) / 1e9,
            "reads_with_rows": int(first.sum().item()), "reads_kept": int((first & (v[:, 0] == 1)).sum().item())}, d_v


def trim_leg(dm, d_rows, d_v, n_rows, bases_ptr, d_off_b, batch, L, dev):
    """SURVEY §8(f-2), outside the timed region: the kit driver's trim configuration (use_kit.rs:87-99) on
    the reads, rows and verdicts of the last step, all resident in HBM: slices planned, grouped by output
    label and rendered as FASTQ text.  HBM-bound; algorithmic bytes = text written + the same bytes read."""
    from barbell_amd.trim import TrimConfig, Trimmer

    tr = Trimmer(dm, TrimConfig.for_kit())
    d_q = torch.randint(33, 74, (batch * L,), dtype=torch.uint8, device=dev)
    W = 40  # "r000001234 ch=0123 start_time=2024-01-01" style fixed-width header
    idx = np.arange(batch, dtype=np.int64)
    hdr = np.tile(np.frombuffer(b"r000000000 ch=0000 st=2024-01-01T00:00Z " [:W], dtype=np.uint8), (batch, 1))
    for d in range(9):
        hdr[:, 9 - d] = 48 + (idx // 10 ** d) % 10
    d_hdr = torch.from_numpy(hdr.reshape(-1)).to(dev)
    d_hoff = torch.arange(0, batch + 1, dtype=torch.int64, device=dev) * W
    d_idl = torch.full((batch,), 10, dtype=torch.int32, device=dev)
    d_ds = torch.full((batch,), 11, dtype=torch.int32, device=dev)
    text_cap = 2 * batch * L + 128 * batch
    d_text = torch.empty(text_cap, dtype=torch.uint8, device=dev)
    d_sl = torch.empty(2 * batch * 32, dtype=torch.uint8, device=dev)
    d_sp = torch.empty(4096 * 32, dtype=torch.uint8, device=dev)
    d_st = torch.empty(batch, dtype=torch.uint8, device=dev)
    call = lambda: tr.trim_batch_dev(d_rows.data_ptr(), d_v.data_ptr(), n_rows, bases_ptr, d_q.data_ptr(), d_off_b.data_ptr(),
                                     d_hdr.data_ptr(), d_hoff.data_ptr(), d_idl.data_ptr(), d_ds.data_ptr(), batch, d_text.data_ptr(),
                                     text_cap, d_sl.data_ptr(), 2 * batch, d_sp.data_ptr(), 4096, d_st.data_ptr())
    call()
    ms = {"plan_sort": 0.0, "render": 0.0, "total": 0.0}
    reps = 5
    for _ in range(reps):
        tl, ns, nsp = call()
        for k, v in tr.last_ms().items():
            ms[k] += v / reps
    gbs = 2.0 * tl / (ms["render"] * 1e-3) / 1e9
    return {"config": "kit driver (labels, left side only, full header)", "records": ns, "labels": nsp, "text_bytes": tl,
            "ms_plan_sort": ms["plan_sort"], "ms_render": ms["render"], "render_gb_per_s": gbs, "render_frac_of_hbm_peak": gbs / HBM_PEAK_GBS,
            "reads_trimmed": int((d_st == 1).sum().item()), "reads_failed": int((d_st == 2).sum().item())}


def ingest_leg(dm, bases_ptr, batch, L, dev):
    """SURVEY §8(f-3), outside the timed region: the batch as raw FASTQ text in HBM (fixed-width records built
    with torch from the resident reads) parsed and packed by bb_fastq_ingest_dev; the packed bases must equal
    the reads they were rendered from.  HBM-bound: text read twice (newline pass, pack pass), ~its size written."""
    from barbell_amd import fastq as Q

    W = 40
    idx = np.arange(batch, dtype=np.int64)
    hdr = np.tile(np.frombuffer(b"@r00000000 ch=0000 st=2024-01-01T00:00Z\n"[: W + 1], dtype=np.uint8), (batch, 1))
    for d in range(8):
        hdr[:, 9 - d] = 48 + (idx // 10 ** d) % 10
    reads = _view_u8(bases_ptr, batch * L, dev).view(batch, L)
    quals = torch.randint(33, 74, (batch, L), dtype=torch.uint8, device=dev)
    sep = torch.tensor(list(b"\n+\n"), dtype=torch.uint8, device=dev).repeat(batch, 1)
    nl = torch.full((batch, 1), 10, dtype=torch.uint8, device=dev)
    text = torch.cat([torch.from_numpy(hdr).to(dev), reads, sep, quals, nl], dim=1).contiguous().view(-1)
    del quals, sep, nl
    n_text = text.numel()
    info, b = Q.ingest(dm, n_text, True, device_ptr=text.data_ptr())
    ok = int(info.n_records) == batch and int(info.n_bases) == batch * L and \
        torch.equal(_view_u8(b.d_bases, batch * L, dev), reads.reshape(-1))
    ms = 0.0
    reps = 5
    for _ in range(reps):
        Q.ingest(dm, n_text, True, device_ptr=text.data_ptr())
        ms += lib_ms(dm) / reps
    moved = 2.0 * n_text + 2.0 * batch * L + batch * W  # newline pass + pack pass reads, packed arrays written
    return {"records": int(info.n_records), "text_bytes": n_text, "ms": ms, "text_gb_per_s": n_text / (ms * 1e-3) / 1e9,
            "hbm_gb_per_s": moved / (ms * 1e-3) / 1e9, "frac_of_hbm_peak": moved / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "round_trip_ok": bool(ok)}


def lib_ms(dm):
    from barbell_amd._lib import lib

    return float(lib().bb_fastq_last_ms(dm._ctx()))


def _view_u8(ptr, n, dev):
    """torch uint8 view of n bytes of device memory at ptr (no copy)"""
    class _A:  # __cuda_array_interface__ carrier
        pass

    a = _A()
    a.__cuda_array_interface__ = {"shape": (n,), "typestr": "|u1", "data": (int(ptr), False), "version": 2}
    return torch.as_tensor(a, device=dev)


def compute_section(args, kavg, cells_flank, batch, L):
    """What actually bounds the path: integer VALU issue.  Ceilings are MEASURED (tools/valu_ceiling.hip ->
    profiles/valu_ceiling.json: G wave64-instructions/s of the whole chip per instruction class, 1-8 waves per SIMD);
    per-kernel VALU instruction counts per launch come from the committed SQ counter passes (tools/profile_round.sh ->
    tools/collect_valu.py -> profiles/valu_<config>.json; they cannot be counted live).  valu_issue_frac = achieved
    wave-instructions/s over the ceiling of the class mix's two ends: a kernel of only full-rate instructions could reach
    the first, one of only half-rate instructions the second."""
    out = {"flank_gcups": cells_flank / (kavg.get("k_flank_scan", 0.0) * 1e-3 + 1e-12) / 1e9,
           "note": "integer VALU issue bound (bit-parallel Myers); no MFMA. Ceilings measured per instruction class: profiles/valu_ceiling.json"}
    try:
        ceil = json.load(open(os.path.join(ROOT, "profiles", "valu_ceiling.json")))
        peak = lambda k: max(x["G"] for x in ceil["classes"][k]["ind"].values())
        full, half = peak("v_add_u32"), peak("v_lshl_or_b32")
        out["valu_ceiling_G_wave_instr_per_s"] = {"full_rate_classes": full, "half_rate_classes": half, "source": "profiles/valu_ceiling.json"}
        v = json.load(open(os.path.join(ROOT, "profiles", f"valu_{args.config}.json")))
        if v.get("batch_reads") == batch and v.get("read_len") == L:
            slot = {"k_flank_scan": ["k_flank_scan2", "k_flank_filter", "k_flank_verify"], "k_flank_trace": ["k_flank_trace"], "k_barcode": ["k_bar_prefix", "k_barcode_lane", "k_barcode_pfx", "k_barcode_reg", "k_rows"]}
            per = {}
            for name, pre in slot.items():
                n = sum(e.get("SQ_INSTS_VALU", 0.0) for k, e in v["kernels"].items() if any(k.startswith(p) for p in pre))
                ms = kavg.get(name, 0.0)
                if n and ms:
                    g = n / (ms * 1e-3) / 1e9
                    per[name] = {"valu_wave_instr_per_launch": n, "G_wave_instr_per_s": g, "valu_issue_frac_of_full_rate_ceiling": g / full,
                                 "valu_issue_frac_of_half_rate_ceiling": g / half}
            out["kernels"] = per
            out["counts_source"] = f"profiles/valu_{args.config}.json"
    except Exception as e:  # the profile files are evidence, not a dependency of the measurement
        out["ceiling_error"] = str(e)[:200]
    return out


def load_traffic(args, batch, L, dom, dom_kernel):
    """HBM bytes per launch from the committed PMC passes (tools/collect_traffic.py), if they were
    collected on this workload shape; otherwise null.  They cannot be measured live."""
    path = os.path.join(ROOT, "profiles", f"traffic_{args.config}.json")
    try:
        t = json.load(open(path))
    except Exception:
        return None, None, None
    if t.get("batch_reads") != batch or t.get("read_len") != L:
        return None, None, None
    ks = t["kernels"]
    # the k_barcode timing slot covers k_bar_prefix + k_barcode_lane | k_barcode_pfx (both strands) + k_rows + the exact kernel on undecided hits
    # the dominant kernel's own bytes per launch (profile keys carry the template arguments rocprofv3 prints; older files lack the third)
    pre = (", ".join(dom_kernel.split(", ")[:2]),) if dom_kernel != dom else (dom,)
    dom_bytes = sum(v["hbm_bytes"] for k, v in ks.items() if k.startswith(pre))
    return (dom_bytes or None, sum(v["hbm_bytes"] * v.get("launches_per_step", 1) for v in ks.values() if v.get("in_step", True)),
            os.path.relpath(path, ROOT))


def cpu_baseline(args, groups, dm, d_bases, L, batch, last_batch, d_rows, last_rows):
    """CPU baseline on a bounded sample of the SAME reads, in the same run.

    * A real `barbell` binary on PATH / $BARBELL_BIN (SURVEY §8d's preferred baseline; none exists in the build
      image): timed on the sample written as FASTQ, kind "reference", and its annotation.tsv is diffed against the
      HIP rows (tools/ref_diff.py) -> "reference_parity".  Otherwise "reference_parity": "unpinned beyond KATs".
    * The CPU checker is always run on three windows of the LAST timed batch — its first reads, its middle and its last
      reads, whose byte offsets lie beyond 4 GiB — with its scalar restatement (what every parity test compares the GPU with),
      and each window's rows are compared bit-exact with the full-batch device rows restricted to it.
    * The reported CPU rate is the checker's timing path on a larger window of the same batch, OpenMP over reads, the same rows
      (tests/test_oracle_fast.py): kind "port-simd" where the host has AVX-512 (round 6: the flank scan text-parallel as sassy's `search` is
      — chunks of the read in the lanes of a vector with an m + k overlap —, the barcode pass pattern-parallel as its
      `search_encoded_patterns`, Lodhi scores eight candidates per vector: oracle/bb_oracle_simd.h), "port-bitparallel" (64-bit Myers /
      Hyyro words) elsewhere; the scalar restatement's rate is carried as `scalar_value`.  Real Barbell runs SIMD sassy: a scalar O(m n)
      loop would understate what a CPU does by an order of magnitude."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import ref_diff
    from barbell_amd import _abi
    from oracle import pyoracle as po

    po.build()
    cores = effective_cpus()
    orc = po.Oracle([g.as_tuple() for g in groups])
    probe = min(max(64, 8 * cores), 4096, batch)
    base0 = last_batch * batch  # first resident read of the last timed batch

    def sample(first, n):
        bases = d_bases[(base0 + first) * L: (base0 + first + n) * L].cpu().numpy()
        return bases, (np.arange(n + 1, dtype=np.uint64) * np.uint64(L))

    b, o = sample(0, probe)
    t = time.perf_counter()
    orc.annotate(b, o, n_threads=cores)
    rate = probe / (time.perf_counter() - t)
    n = int(min(max(probe, rate * args.cpu_seconds * 0.5), batch))
    w = max(1, n // 3)
    full = np.frombuffer(d_rows[: last_rows * 48].cpu().numpy().tobytes(), dtype=_abi.ROW_DTYPE)
    windows, total_dt, total_rows, sample_bases = {}, 0.0, 0, None
    for name, first in (("head", 0), ("middle", max(0, batch // 2 - w // 2)), ("tail", batch - w)):
        b, o = sample(first, w)
        if name == "head":
            sample_bases = b
        t = time.perf_counter()
        want = orc.annotate(b, o, n_threads=cores)
        total_dt += time.perf_counter() - t
        got = full[(full["read_idx"] >= first) & (full["read_idx"] < first + w)].copy()
        got["read_idx"] -= first
        windows[name] = {"first_read": int(first), "reads": int(w), "first_byte_offset": int(first) * L, "rows": int(len(want)),
                         "parity": bool(got.tobytes() == want.tobytes())}
        total_rows += len(want)
    # the reported rate: the bit-parallel path on a larger head window of the same batch (sized by a probe to ~half the budget)
    b, o = sample(0, probe)
    t = time.perf_counter()
    orc.annotate(b, o, n_threads=cores, fast=True)
    frate = probe / (time.perf_counter() - t)
    nf = int(min(max(probe, frate * args.cpu_seconds * 0.5), batch))
    b, o = sample(0, nf)
    t = time.perf_counter()
    fwant = orc.annotate(b, o, n_threads=cores, fast=True)
    fdt = time.perf_counter() - t
    fgot = full[full["read_idx"] < nf]
    visible = os.cpu_count() or 1
    simd = po.Oracle.fast_is_simd()
    out = {"value": nf / fdt, "unit": "reads/s", "cores": cores, "kind": "port-simd" if simd else "port-bitparallel", "cpus_visible": visible,
           "per_core": nf / fdt / cores,
           "simd": ("AVX-512: flank scan text-parallel (8 chunks of the read in the 64-bit lanes of a vector, m + k columns of overlap, valleys replayed by the "
                    "local-minimum machine), barcode pass pattern-parallel (8 padded barcodes per vector), 8 walks back in lockstep, Lodhi scores 8 candidates per vector; "
                    "round 5's 64-bit words on the same box: 6.8 k reads/s per core" if simd else "none (no AVX-512 on this host): 64-bit Myers words"),
           "cores_note": (f"{cores} = this container's CPU quota (cgroup cpu.max) of the {visible} CPUs it sees; one pinned OpenMP worker per quota CPU "
                          f"(a pool of {visible} floating threads ran slower: throttled)" if cores < visible else f"all {cores} CPUs, one pinned OpenMP worker each"),
           "sample": f"first {nf} reads of the last timed {batch}-read batch of the same synthetic stream, CPU checker's timing path "
                     f"(bbo_annotate_batch_fast, OpenMP over reads, {cores} threads), {fdt:.1f} s wall; parity windows: 3 x {w} reads (head, middle, tail) with "
                     f"the scalar restatement, {total_dt:.1f} s wall",
           "scalar_value": 3 * w / total_dt, "bitparallel_rows_equal_gpu": bool(fgot.tobytes() == fwant.tobytes()),
           "parity_on_sample": all(v["parity"] for v in windows.values()), "parity_windows": windows, "rows_on_sample": int(total_rows),
           "reference_parity": ref_diff.UNPINNED}
    bin_ = ref_diff.find_barbell()
    if bin_:  # real Barbell on the same box: the baseline SURVEY §8d prefers, and the parity check §8c promised
        import tempfile

        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import ref_export
        from barbell_amd import annotate as A

        with tempfile.TemporaryDirectory() as td:
            cid, _, flags, files = ref_export.CONFIGS[args.config]
            for f in files:
                import shutil
                shutil.copy(os.path.join(ref_export.EX, f), td)
            nref = min(batch, max(w, int(200_000)))
            rb, ro = sample(0, nref)
            ref_export.write_fastq(os.path.join(td, "reads.fastq"), rb, ro)
            json.dump({"config": args.config, "n_reads": nref, "barbell_args": flags}, open(os.path.join(td, "manifest.json"), "w"))
            got = full[full["read_idx"] < nref]
            ids = [f"r{i}" for i in range(nref)]
            lines = A.format_rows(got, ids, groups)
            with open(os.path.join(td, "ours.tsv"), "w") as f:
                if lines:
                    f.write(A.TSV_HEADER + "\n" + "\n".join(lines) + "\n")
            try:
                rep, secs = ref_diff.reference_check(td, bin_, os.path.join(td, "ours.tsv"), threads=cores)
                out.update({"value": nref / secs, "kind": "reference", "cores": cores, "port_value": nf / fdt,
                            "sample": f"real barbell ({bin_}) annotate -t {cores} on the first {nref} reads of the last timed batch written as FASTQ, "
                                      f"{secs:.1f} s wall incl. its file IO",
                            "reference_parity": {k: rep[k] for k in ("reference_parity", "mismatch_rate", "bucket_rates", "hazards", "rows_ref", "rows_ours")}})
            except Exception as e:  # a binary that does not run here is reported, not fatal
                out["reference_error"] = str(e)[:400]
    return out


if __name__ == "__main__":
    main()
