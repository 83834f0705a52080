#!/bin/bash
# per-kernel device time of ANY command (runs on the GPU box): tools/kstats_cmd.sh <command ...>
export TMPDIR=/tmp; D=/tmp/kt_$$; cd /tmp
rocprofv3 --kernel-trace --stats -d $D -o t --output-format csv -- "$@" > $D.log 2>&1
f=$(find $D -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:20]:
    name = re.sub(r"[(].*", "", r["Name"])[:58]
    print(f'{name:60s} calls {r["Calls"]:>6s} avg_us {float(r["AverageNs"]) / 1e3:8.1f} pct {r["Percentage"]}')
PY
rm -rf $D
