#!/usr/bin/env python3
"""Summarises a rocprofv3 rocpd SQLite database (rocprofv3 --kernel-trace --stats -d DIR) as a
per-kernel table: calls, total/avg/min/max duration (ns), share, VGPRs, LDS, scratch.
usage: rocpd_summary.py results.db [> profiles/rNN_kernel_stats.txt]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = db.execute(
    "select name, count(*), sum(duration), avg(duration), min(duration), max(duration), max(vgpr_count), max(lds_size), "
    "max(scratch_size), max(grid_x), max(workgroup_x) from kernels group by name order by sum(duration) desc").fetchall()
tot = sum(r[2] for r in rows) or 1
print(f"{'kernel':72s} {'calls':>6s} {'total_ns':>14s} {'avg_ns':>12s} {'min_ns':>12s} {'max_ns':>12s} {'pct':>6s} {'vgpr':>5s} {'lds':>7s} {'scratch':>8s} {'grid':>10s} {'wg':>5s}")
for r in rows:
    name = r[0] if len(r[0]) <= 72 else r[0][:69] + "..."
    print(f"{name:72s} {r[1]:6d} {r[2]:14d} {r[3]:12.0f} {r[4]:12d} {r[5]:12d} {100.0 * r[2] / tot:6.2f} {r[6]:5d} {r[7]:7d} {r[8]:8d} {r[9]:10d} {r[10]:5d}")
