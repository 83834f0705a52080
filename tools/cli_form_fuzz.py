#!/usr/bin/env python3
"""dev aid (GPU box): `barbell-amd annotate` on random FASTQ layouts (tools/stage_fuzz.py's generator with real kit constructs in some reads) in its
three upload forms — packed (default), --no-pack (sequence lines as text), --no-compact (whole records) — must succeed alike and write the
same annotation.tsv.  usage: cli_form_fuzz.py FIRST_SEED N_SEEDS"""
import gzip
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from barbell_amd import kits  # noqa: E402

CLI = os.path.join(ROOT, "barbell_amd", "bin", "barbell-amd")
D = "/tmp/barbell_cli_fuzz"
os.makedirs(D, exist_ok=True)
groups = kits.groups_from_kit("SQK-NBD114-96", flank_max_errors=3)
bad = 0
for seed in range(int(sys.argv[1]), int(sys.argv[1]) + int(sys.argv[2])):
    rng = np.random.default_rng(seed)
    nl = b"\r\n" if rng.random() < 0.3 else b"\n"
    files = []
    for f in range(int(rng.integers(1, 4))):
        recs = []
        for i in range(int(rng.integers(0, 60))):
            L = int(rng.choice([0, 1, 2, 31, 32, 33, 64, int(rng.integers(0, 400)), int(rng.integers(0, 3000))]))
            alpha = rng.choice([b"ACGT", b"ACGTNacgtn", b"ACGTRYKMSWBDHVNU*-"], p=[0.6, 0.3, 0.1])
            s = rng.choice(np.frombuffer(bytes(alpha), dtype=np.uint8), L).tobytes()
            if rng.random() < 0.5:
                s = bytes(groups[0].seqs[int(rng.integers(96))]) + s
            recs.append(((b"r%d_%d" % (f, i)) + (b" x y" if i % 3 == 0 else b""), s))
        text = b"".join(b"@" + h + nl + s + nl + b"+" + nl + b"I" * len(s) + nl for h, s in recs)
        final_nl = rng.random() < 0.8
        if not final_nl and recs and len(recs[-1][1]) > 0:
            text = text[: -len(nl)]
        if final_nl and recs:
            text += [b"", b"", nl, nl + nl][int(rng.integers(0, 4))]
        p = f"{D}/f{f}.fq" + (".gz" if rng.random() < 0.25 else "")
        if p.endswith(".gz") and rng.random() < 0.5 and len(text) > 10:   # several gzip members, cut anywhere in the text (`cat a.gz b.gz`)
            cuts = [0] + sorted(int(x) for x in rng.integers(0, len(text), int(rng.integers(1, 12)))) + [len(text)]
            with open(p, "wb") as fh:
                for a, b in zip(cuts[:-1], cuts[1:]):
                    fh.write(gzip.compress(text[a:b], compresslevel=int(rng.choice([0, 1, 6]))))
        else:
            with (gzip.open(p, "wb") if p.endswith(".gz") else open(p, "wb")) as fh:
                fh.write(text)
        files.append(p)
    block = int(rng.choice([257, 1000, 4096, 70000, 1 << 20]))
    env = dict(os.environ, BARBELL_AMD_NO_TORCH="1")
    # round 5: gzip input inflated in pieces and by ranges of members (tiny ones), the scans' segments (nearly every read cut)
    if rng.random() < 0.6: env["BARBELL_AMD_GZ_PIECE"] = str(int(rng.choice([64, 500, 4000, 100000])))
    if rng.random() < 0.6: env["BARBELL_AMD_GZ_RANGE"] = str(int(rng.choice([64, 300, 2000, 20000])))
    if rng.random() < 0.4: env["BARBELL_AMD_SEG_LINES"] = str(int(rng.choice([4, 8, 16])))
    outs = {}
    for name, extra in (("packed", []), ("text", ["--no-pack"]), ("whole", ["--no-compact"])):
        r = subprocess.run([CLI, "annotate", "-i"] + files + ["-o", f"{D}/{name}.tsv", "--kit", "SQK-NBD114-96", "--flank-max-errors", "3", "--block-bytes", str(block),
                            "-t", str(int(rng.integers(1, 6)))] + extra, capture_output=True, text=True, env=env, timeout=300)
        outs[name] = (r.returncode, open(f"{D}/{name}.tsv", "rb").read() if r.returncode == 0 else r.stderr[-200:])
    if not (outs["packed"] == outs["text"] == outs["whole"]) or outs["whole"][0] != 0:
        bad += 1
        print(f"seed {seed} block {block} nl {nl!r}: " + ", ".join(f"{k}: rc {v[0]} {len(v[1]) if v[0] == 0 else v[1]!r}" for k, v in outs.items()))
    for p in files:
        os.remove(p)
print(int(sys.argv[2]), "seeds", bad, "bad")
sys.exit(1 if bad else 0)
