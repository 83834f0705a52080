#!/usr/bin/env python3
"""ref_export.py — export the parity inputs so that anyone with a Rust box can pin this repo against real Barbell.

The reference's search/score arithmetic lives in crates that are absent from this image (sassy 0.2.1,
cigar-lodhi-rs 0.1.0; /root/reference/Cargo.toml:20,36), so GPU == oracle is all that can be proven here
(oracle/README.md, hazards H1-H8).  This tool writes, for each BASELINE query set, a deterministic synthetic FASTQ
(the generator of bench.py / the tests: bb_synth_reads_host, seed 0xBA7BE11 ^ config id) together with the exact
`barbell annotate` command line (flags of /root/reference/bin/main.rs:64-112) that produces the golden
annotation.tsv.  No GPU is needed (read synthesis is host code of libbarbell_amd.so).

  tools/ref_export.py OUT_DIR [--reads 10000] [--configs rbk24,nbd96,dual,rbk96x,nbd96x,nbd96n,dualn]

OUT_DIR/<config>/reads.fastq         the reads, ids r0 .. r{n-1}, constant quality
OUT_DIR/<config>/manifest.json       {"barbell_args": [...], "n_reads", "seed", "read_len": [lo, hi], ...}
OUT_DIR/<config>/run_reference.sh    barbell annotate <args> -i reads.fastq -o ref.tsv -t $THREADS
OUT_DIR/<config>/*.fasta             query files of the custom (-q) configs

Then: tools/ref_diff.py OUT_DIR/<config> --barbell /path/to/barbell   (see that file).
"""
import argparse
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

EX = os.path.join(ROOT, "tests", "golden", "examples")

# config -> (config id for the seed, read length range, barbell annotate flags, query files to copy)
CONFIGS = {
    "rbk24": (1, (600, 4000), ["--kit", "SQK-RBK114-24"], []),
    "nbd96": (2, (4000, 4000), ["--kit", "SQK-NBD114-96", "--flank-max-errors", "3"], []),
    "dual": (4, (4000, 4000), ["-q", "native_left.fasta", "native_right.fasta", "-b", "Ftag", "Rtag", "--flank-max-errors", "5"],
             ["native_left.fasta", "native_right.fasta"]),
    "rbk96x": (5, (4000, 4000), ["--kit", "SQK-RBK114-96", "--use-extended"], []),
    "nbd96x": (6, (4000, 4000), ["--kit", "SQK-NBD114-96", "--use-extended", "--flank-max-errors", "3"], []),
    # the same two query sets on short reads with 8 % substitutions everywhere: on clean synthetic reads the decisions sit far
    # from the score thresholds and the Lodhi / tie assumptions (include/barbell_amd_policy.h, H7, H8) never show; here they do
    "nbd96n": (7, (300, 1500), ["--kit", "SQK-NBD114-96", "--flank-max-errors", "3"], []),
    "dualn": (8, (300, 1500), ["-q", "native_left.fasta", "native_right.fasta", "-b", "Ftag", "Rtag", "--flank-max-errors", "5"],
              ["native_left.fasta", "native_right.fasta"]),
}
NOISE = {"nbd96n": 0.08, "dualn": 0.08}


def config_groups(name):
    from barbell_amd import _abi, kits

    if name in ("nbd96", "nbd96n"):
        return kits.groups_from_kit("SQK-NBD114-96", flank_max_errors=3)
    if name == "rbk24":
        return kits.groups_from_kit("SQK-RBK114-24")
    if name in ("dual", "dualn"):
        return [kits.group_from_fasta(os.path.join(EX, "native_left.fasta"), _abi.BB_FTAG, 5),
                kits.group_from_fasta(os.path.join(EX, "native_right.fasta"), _abi.BB_RTAG, 5)]
    if name == "rbk96x":
        return kits.groups_from_kit("SQK-RBK114-96", use_extended=True)
    if name == "nbd96x":
        return kits.groups_from_kit("SQK-NBD114-96", use_extended=True, flank_max_errors=3)
    raise KeyError(name)


def write_fastq(path, bases, offsets, first_id=0):
    """reads as 4-line FASTQ records with ids r{i}; quality 'I' throughout (annotate ignores it, annotator.rs:125-127)"""
    import numpy as np

    with open(path, "wb") as f:
        n = len(offsets) - 1
        for i in range(n):
            s = bases[int(offsets[i]): int(offsets[i + 1])].tobytes()
            f.write(b"@r%d\n" % (first_id + i) + s + b"\n+\n" + b"I" * len(s) + b"\n")
    return n


def export_config(name, out_dir, n_reads):
    from barbell_amd import annotate as A

    cid, (lo, hi), flags, files = CONFIGS[name]
    d = os.path.join(out_dir, name)
    os.makedirs(d, exist_ok=True)
    groups = config_groups(name)
    seed = 0xBA7BE11 ^ cid
    bases, offsets = A.synth_reads_host(groups, seed, lo, hi, 0, n_reads)
    if name in NOISE:
        import numpy as np

        rng = np.random.default_rng(seed)
        pos = rng.random(len(bases)) < NOISE[name]
        bases = bases.copy()
        bases[pos] = rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), int(pos.sum()))
    write_fastq(os.path.join(d, "reads.fastq"), bases, offsets)
    for f in files:
        shutil.copy(os.path.join(EX, f), os.path.join(d, f))
    man = {"config": name, "n_reads": n_reads, "seed": seed, "read_len": [lo, hi], "barbell_args": flags,
           "reference": "rickbeeloo/barbell v0.3.3 (sassy 0.2.1, cigar-lodhi-rs 0.1.0, pa-types 1.2.0)",
           "command": "barbell annotate " + " ".join(flags) + " -i reads.fastq -o ref.tsv -t $THREADS",
           "note": "run inside this directory; reads are 'r<i>' in input order; ref_diff.py restores input order before comparing"}
    json.dump(man, open(os.path.join(d, "manifest.json"), "w"), indent=1)
    with open(os.path.join(d, "run_reference.sh"), "w") as f:
        f.write("#!/bin/sh\n# produces the golden annotation of this read set with real Barbell (v0.3.3)\ncd \"$(dirname \"$0\")\"\n"
                "exec \"${BARBELL_BIN:-barbell}\" annotate " + " ".join(flags) + " -i reads.fastq -o ref.tsv -t \"${THREADS:-8}\"\n")
    os.chmod(os.path.join(d, "run_reference.sh"), 0o755)
    return d


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("out_dir")
    ap.add_argument("--reads", type=int, default=10000)
    ap.add_argument("--configs", default="rbk24,nbd96,dual,rbk96x,nbd96x,nbd96n,dualn")
    a = ap.parse_args()
    for c in a.configs.split(","):
        print(export_config(c, a.out_dir, a.reads))


if __name__ == "__main__":
    main()
