"""barbell_amd/csrc/bb_prio.h: the compile-time move-plane truth tables of every traceback order (policy [H3]) against the rule they
encode — at every cell the first applicable op of the order (Match: eq, Sub: ~d0, Ins: ph, Del: pvn) — over every combination of the
four bits a DP cell can show, and the class table (24 orders, 18 classes).  Host-only: the header's constexpr part compiles with g++."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = r'''
#include <cstdio>
#include "barbell_amd/csrc/bb_prio.h"
template <uint32_t P> struct TT { static constexpr uint32_t lo = bb_prio_tt(P, 0), hi = bb_prio_tt(P, 1); static constexpr int kind = bb_prio_kind(P); };
int main() {
    int bad = 0, orders = 0;
    bool seen[BB_PRIO_CLASSES] = {};
    for (uint32_t prio = 0; prio < 256; ++prio) {
        if (!bb_prio_valid(prio)) continue;
        ++orders;
        const int cls = bb_prio_class(prio);
        if (cls < 0 || cls >= BB_PRIO_CLASSES) { ++bad; continue; }
        seen[cls] = true;
        const uint32_t canon = BB_PRIO_TABLE.cls[cls];   // the kernels are instantiated on the class's canonical order
        const int kind = bb_prio_kind(canon);
        const uint32_t tl = bb_prio_tt(canon, 0), th = bb_prio_tt(canon, 1);
        for (int bits = 0; bits < 16; ++bits) {
            const bool d0 = bits & 1, eq = bits & 2, ph = bits & 4, pvn = bits & 8;
            if (eq && !d0) continue;                      // matching characters: the diagonal neighbour has the same value
            if (d0 && !eq && !ph && !pvn) continue;      // a cell equal to its diagonal neighbour without a match got there by Ins or Del
            const bool ap[4] = {eq, !d0, ph, pvn};       // BB_OP_MATCH, SUB, INS, DEL
            int op = -1;
            for (int q = 0; q < 4 && op < 0; ++q) if (ap[(prio >> (2 * q)) & 3]) op = (prio >> (2 * q)) & 3;   // the ORIGINAL order's rule
            bool a, b, c;
            switch (kind) {
                case BB_PK_D0_EQ_PH: a = d0; b = eq; c = ph; break;
                case BB_PK_D0_EQ_PVN: a = d0; b = eq; c = pvn; break;
                case BB_PK_D0_PH_PVN: a = d0; b = ph; c = pvn; break;
                default: a = eq; b = ph; c = pvn; break;
            }
            const int idx = (a << 2) | (b << 1) | c;
            const int got = ((tl >> idx) & 1) | (((th >> idx) & 1) << 1);   // 0 Match, 1 Sub, 2 Ins, 3 Del
            if (got != op) { ++bad; printf("prio %02x class %d bits %x: table %d rule %d\n", prio, cls, bits, got, op); }
        }
    }
    int n = 0;
    for (bool s : seen) n += s;
    printf("orders %d classes %d bad %d default %02x\n", orders, n, bad, BB_PRIO_TABLE.cls[0]);
    return bad != 0;
}
'''


def test_truth_tables_encode_the_rule(tmp_path):
    src = tmp_path / "prio.cpp"
    src.write_text(SRC)
    exe = tmp_path / "prio"
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-I", ROOT, "-o", str(exe), str(src)])
    out = subprocess.run([str(exe)], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout
    assert out.stdout.strip().endswith("orders 24 classes 18 bad 0 default d8")   # M | I << 2 | S << 4 | D << 6 = 0xD8
