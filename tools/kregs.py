#!/usr/bin/env python3
"""VGPR / SGPR / spills / scratch / LDS of every kernel in libbarbell_amd.so: pulls the gfx950 code objects out of the
.hip_fatbin section (clang offload bundles) and reads their metadata notes.  usage: tools/kregs.py [so] [name filter]"""
import os
import re
import struct
import subprocess
import sys
import tempfile

so = sys.argv[1] if len(sys.argv) > 1 and os.path.exists(sys.argv[1]) else os.path.join(os.path.dirname(__file__), "..", "barbell_amd", "libbarbell_amd.so")
flt = sys.argv[-1] if len(sys.argv) > 1 and not os.path.exists(sys.argv[-1]) else ""
LLVM = "/opt/rocm/lib/llvm/bin"
with tempfile.TemporaryDirectory() as td:
    fb = os.path.join(td, "fatbin")
    subprocess.check_call([f"{LLVM}/llvm-objcopy", "-O", "binary", "--only-section=.hip_fatbin", so, fb])
    blob = open(fb, "rb").read()
    MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"
    rows = []
    pos = 0
    while True:
        pos = blob.find(MAGIC, pos)
        if pos < 0:
            break
        n = struct.unpack_from("<Q", blob, pos + len(MAGIC))[0]
        q = pos + len(MAGIC) + 8
        for _ in range(n):
            off, size, tl = struct.unpack_from("<QQQ", blob, q)
            triple = blob[q + 24:q + 24 + tl].decode()
            q += 24 + tl
            if "gfx950" in triple and size:
                co = os.path.join(td, "dev.co")
                open(co, "wb").write(blob[pos + off:pos + off + size])
                txt = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", co], capture_output=True, text=True).stdout
                for blk in re.split(r"\n\s*- \.agpr_count:", txt)[1:]:
                    g = lambda k: (re.search(r"\." + k + r":\s*(\S+)", blk) or [None, "?"])[1]
                    rows.append((g("name"), g("vgpr_count"), blk.split()[0], g("sgpr_count"), g("vgpr_spill_count"), g("private_segment_fixed_size"), g("group_segment_fixed_size")))
        pos += len(MAGIC)
    names = subprocess.run(["c++filt"], input="\n".join(r[0] for r in rows), capture_output=True, text=True).stdout.splitlines()
    for nm, r in sorted(zip(names, rows)):
        nm = re.sub(r"\(.*", "", nm.replace("void ", ""))
        if flt in nm:
            print(f"{nm[:70]:70s} vgpr {r[1]:>4s} agpr {r[2]:>3s} sgpr {r[3]:>4s} spill {r[4]:>3s} scratch {r[5]:>5s} lds {r[6]:>6s}")
