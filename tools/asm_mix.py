#!/usr/bin/env python3
"""Instruction mix of a gfx950 kernel from hipcc's assembly, priced with the measured issue rates of
profiles/valu_ceiling.json (tools/valu_ceiling.hip): a wave64 VALU instruction of a "full-rate" class occupies its
SIMD for ~2.2 cycles, a "half-rate" one for ~4.2.

  hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -S --cuda-device-only -o k.s barbell_amd/csrc/barbell_amd.hip
  tools/asm_mix.py k.s k_barcode_pfxILi48        # whole kernel
  tools/asm_mix.py k.s k_flank_scan2ILi2 --blocks  # per basic block (pick the hot loop by size)
"""
import collections
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# classes measured at >= 1000 G wave-instr/s (2 waves per SIMD and up); everything else VALU measured ~580 G
FULL = {"v_and_b32", "v_or_b32", "v_xor_b32", "v_not_b32", "v_add_u32", "v_sub_u32", "v_subrev_u32", "v_mov_b32", "v_lshrrev_b32",
        "v_bitop3_b32", "v_accvgpr_write_b32", "v_accvgpr_read_b32"}


def rate_table():
    try:
        d = json.load(open(os.path.join(ROOT, "profiles", "valu_ceiling.json")))
        peak = lambda k: max(x["G"] for x in d["classes"][k]["ind"].values())
        full = peak("v_add_u32")
        half = peak("v_lshl_or_b32")
        simd_cycles = d["cus"] * 4 * 2.4  # G SIMD-cycles/s at the nominal clock
        return simd_cycles / full, simd_cycles / half
    except Exception:
        return 2.2, 4.2


def base(op):
    op = re.sub(r"_(e32|e64|sdwa|dpp)$", "", op)
    return op


def main():
    path, pat = sys.argv[1], sys.argv[2]
    per_block = "--blocks" in sys.argv
    c_full, c_half = rate_table()
    lines = open(path).read().split("\n")
    start = None
    for i, l in enumerate(lines):
        if re.match(r"^_Z\w*:", l) and pat in l:
            start = i
            break
    if start is None:
        sys.exit(f"no kernel matching {pat}")
    blocks = collections.OrderedDict()
    cur = "entry"
    blocks[cur] = collections.Counter()
    for l in lines[start + 1:]:
        if l.startswith(".Lfunc_end"):
            break
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            cur = m.group(1)
            blocks[cur] = collections.Counter()
            continue
        t = l.strip()
        if not t or t.startswith(";") or t.startswith("."):
            continue
        op = t.split()[0]
        sd = "_sdwa" in op
        op = base(op)
        blocks[cur][op + ("(sdwa)" if sd else "")] += 1

    def show(name, cnt):
        valu = {k: v for k, v in cnt.items() if k.startswith("v_")}
        n_full = sum(v for k, v in valu.items() if k in FULL)
        n_half = sum(valu.values()) - n_full
        other = sum(v for k, v in cnt.items() if not k.startswith("v_"))
        cyc = n_full * c_full + n_half * c_half
        print(f"{name}: {sum(cnt.values())} instr, VALU {n_full + n_half} (full-rate {n_full}, half-rate {n_half}), other {other}; "
              f"VALU issue ~{cyc:.0f} SIMD-cycles")
        for k, v in sorted(cnt.items(), key=lambda kv: -kv[1])[:24]:
            tag = "F" if k in FULL else ("h" if k.startswith("v_") else " ")
            print(f"    {v:6d} {tag} {k}")

    if per_block:
        for b, cnt in blocks.items():
            if sum(cnt.values()) >= 40:
                show(b, cnt)
    else:
        tot = collections.Counter()
        for cnt in blocks.values():
            tot.update(cnt)
        show(pat, tot)


if __name__ == "__main__":
    main()
