#!/bin/bash
# per-kernel device time of one bench configuration (runs on the GPU box): tools/kstats.sh TAG [bench args]
TAG=${1:-ks}; shift
REPO=$(cd "$(dirname "$0")/.." && pwd); OUT=$REPO/gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace --stats -d "$OUT/trace" -o t --output-format csv -- python $REPO/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-other-configs --no-policy-variants --no-boundary "$@" > "$OUT/bench.json" 2> "$OUT/trace.log"
f=$(find "$OUT/trace" -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" "$OUT/kernel_stats.csv"
python - "$OUT/kernel_stats.csv" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:14]:
    print(f'{r["Name"][:60]:60s} calls {r["Calls"]:>5s} avg_us {float(r["AverageNs"])/1e3:10.1f} total_ms {float(r["TotalDurationNs"])/1e6:9.2f} {r["Percentage"]}%')
PY
rm -rf "$OUT/trace"
