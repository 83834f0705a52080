import itertools
import os

from barbell_amd import _abi, kits

EX = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "examples")


def config_groups(name):
    """Query groups of the BASELINE.json configs."""
    if name == "nbd96":  # configs[1]/[2]: SQK-NBD114-96, --flank-max-errors 3
        return kits.groups_from_kit("SQK-NBD114-96", flank_max_errors=3)
    if name == "rbk24":  # configs[0]: SQK-RBK114-24, automatic flank cutoff
        return kits.groups_from_kit("SQK-RBK114-24")
    if name == "dual":  # configs[3]: custom dual-end, --flank-max-errors 5
        return [kits.group_from_fasta(os.path.join(EX, "native_left.fasta"), _abi.BB_FTAG, 5),
                kits.group_from_fasta(os.path.join(EX, "native_right.fasta"), _abi.BB_RTAG, 5)]
    if name == "rbk96x":  # configs[4] made meaningful: SQK-RBK114-96 --use-extended (2 groups)
        return kits.groups_from_kit("SQK-RBK114-96", use_extended=True)
    if name == "nbd96x":  # configs[4] literally: --use-extended is a no-op for SQK-NBD114-96
        return kits.groups_from_kit("SQK-NBD114-96", use_extended=True, flank_max_errors=3)
    raise KeyError(name)


def noisy_reads(cfg, seed, n, lmin, lmax, rate=0.08):
    """synthetic reads of a config with `rate` substitutions everywhere (scores land near the thresholds, equal-cost
    alternatives appear): (groups, bases, offsets)"""
    import numpy as np

    from barbell_amd import annotate as A

    groups = config_groups(cfg)
    bases, offsets = A.synth_reads_host(groups, seed, lmin, lmax, 0, n)
    rng = np.random.default_rng(seed)
    b = bases.copy()
    pos = rng.random(len(b)) < rate
    b[pos] = rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), int(pos.sum()))
    return groups, b, offsets


# ---- the policy space (include/barbell_amd_policy.h): what tests/test_policy.py checks and bench.py's policy_variants leg times ----
# The 18 distinguishable traceback orders (barbell_amd/csrc/bb_prio.h): Match and Sub exclude each other, so orders that differ by
# swapping ADJACENT M and S take the same op at every cell; a class is named by its order with M before S where adjacent.
TRACE_CLASSES = ["MISD"] + sorted("".join(p) for p in itertools.permutations("MSID") if "SM" not in "".join(p) and "".join(p) != "MISD")
assert len(TRACE_CLASSES) == 18
ALTERNATIVES = {   # hazard -> the non-default settings (the default is the first value of each key in POLICY_DEFAULT)
    "H1": ["lm=left", "lm=strict"],
    "H2": ["rc=fwd"],
    "H3": ["trace=" + c for c in TRACE_CLASSES[1:]] + ["trace=SMID"],   # SMID: the non-canonical spelling of MSID's class
    "H5": ["rcpath=mirror"],
    "H4": ["ovh=ceil", "ovh=near", "ovh=floor:f64", "ovh=ceil:f64"],
    "H7": ["tie=last"],
    "H8": ["lodhi=3:0.5:2211", "lodhi=3:0.5:1110", "lodhi=3:0.5:2131", "lodhi=2:0.5:1111", "lodhi=3:0.7:1111", "lodhi=4:0.5:1111"],
}
GPU_POLICIES = [a for alts in ALTERNATIVES.values() for a in alts] + [
    "lm=left,rc=fwd,trace=MSID,ovh=ceil,tie=last,lodhi=3:0.5:2211,rcpath=mirror",      # everything at once, register-resident Lodhi family
    "lm=left,rc=fwd,trace=MSID,ovh=ceil,tie=last,lodhi=3:0.5:2211",                    # the same without the refuted rcpath=mirror
    "lm=strict,tie=last,lodhi=3:0.5:1121",
]


# ---- what the reference's own vectors and invariants still allow (tools/policy_feasible.py -> tests/golden/policy_feasible.json) ----
def policy_feasible():
    import json

    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "policy_feasible.json")) as f:
        return json.load(f)


def is_feasible(policy_text):
    """every field the (possibly partial) policy text names holds a value the reference's KATs (cigar_parse.rs:104-176) and no-panic /
    windowing invariants (searcher.rs:388, :445-456) leave open; `lodhi` is unconstrained by them"""
    f = policy_feasible()
    full = dict(tok.split("=", 1) for tok in _abi.policy_to_str(_abi.policy_from_str(policy_text)).split(","))
    full["trace"] = full["trace"].replace("SM", "MS")
    keys = [k for k in f["space"]]
    if not f["feasible_is_product_of_fields"]:
        return ",".join(f"{k}={full[k]}" for k in keys) in f["feasible_joint"]
    return all(full[k] in f["feasible"][k] for k in keys)


def split_feasible(policies):
    """(feasible, refuted) halves of a list of policy texts, order kept"""
    ok = [p for p in policies if is_feasible(p)]
    return ok, [p for p in policies if p not in ok]


def long_batch(groups, seed, long_lens=(300_000, 1_000_000, 2_500_000), n_short=400):
    """a batch with megabase reads (glued from synthetic constructs: flank hits all along them) first, in the middle and last among ordinary reads"""
    import numpy as np

    from barbell_amd import annotate as A

    bases, offsets = A.synth_reads_host(groups, seed, 200, 4000, 0, n_short + sum(long_lens) // 2000 + 8)
    reads = [bases[int(offsets[i]):int(offsets[i + 1])] for i in range(len(offsets) - 1)]
    out, k = [], 0
    for L in long_lens:
        parts, tot = [], 0
        while tot < L:
            parts.append(reads[k]); tot += len(reads[k]); k += 1
        out.append(np.concatenate(parts)[:L])
    short = reads[k:k + n_short]
    # long reads first, in the middle and last
    seq = [out[0]] + short[: n_short // 2] + [out[1]] + short[n_short // 2:] + [out[2]]
    offs = np.zeros(len(seq) + 1, dtype=np.uint64)
    offs[1:] = np.cumsum([len(r) for r in seq])
    return np.concatenate(seq), offs


def heavy_tailed_batch(groups, n, seed=3, scale=1.0):
    """reads whose lengths differ as a nanopore run's do — 70 % 200..3000 nt, 25 % 3..12 kb, 4.5 % 12..40 kb, 0.5 % 40..120 kb (x scale) —
    in random order; synthetic constructs at the ends of every read as in the benchmark's reads"""
    import numpy as np

    from barbell_amd import annotate as A

    parts = []
    for frac, lo, hi in ((0.70, 200, 3000), (0.25, 3000, 12000), (0.045, 12000, 40000), (0.005, 40000, 120000)):
        k = max(1, int(n * frac))
        parts.append(A.synth_reads_host(groups, seed + len(parts), max(1, int(lo * scale)), max(2, int(hi * scale)), 0, k))
    reads = [(p, i) for p, (b, o) in enumerate(parts) for i in range(len(o) - 1)]
    np.random.default_rng(seed).shuffle(reads)
    lens = np.array([int(parts[p][1][i + 1] - parts[p][1][i]) for p, i in reads], dtype=np.uint64)
    offs = np.zeros(len(reads) + 1, dtype=np.uint64)
    offs[1:] = np.cumsum(lens)
    out = np.empty(int(offs[-1]), dtype=np.uint8)
    for j, (p, i) in enumerate(reads):
        b, o = parts[p]
        out[int(offs[j]):int(offs[j + 1])] = b[int(o[i]):int(o[i + 1])]
    return out, offs
