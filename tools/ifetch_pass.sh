#!/bin/bash
# Runs on the GPU box (through gpurun): instruction-fetch counters of the default bench workload, the two
# k_barcode_lane launches overlapped (default) and one after the other (BARBELL_AMD_NO_SIDE_STREAM=1).
#   tools/ifetch_pass.sh TAG [bench args...]
# Output: gpurun_out/TAG/ifetch_{overlap,serial}_{1,2}.txt  (counters in their own runs, never with a trace)
TAG=${1:-ifetch}; shift
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
BENCH="python $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-other-configs --no-e2e --no-stress --no-policy-variants --no-boundary $*"
cd /tmp
for mode in overlap serial; do
  if [ $mode = serial ]; then export BARBELL_AMD_NO_SIDE_STREAM=1; else unset BARBELL_AMD_NO_SIDE_STREAM; fi
  i=0
  for set in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" \
             "SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VALU GRBM_GUI_ACTIVE"; do
    i=$((i+1))
    rocprofv3 --pmc $set -d "$OUT/$mode$i" -o p --output-format csv -- $BENCH > "$OUT/$mode$i.log" 2>&1
    python $REPO/tools/pmc_summary.py "$OUT/$mode$i" > "$OUT/ifetch_${mode}_$i.txt" 2>&1
    find "$OUT/$mode$i" -name "*.csv" -size +20M -delete
    rm -rf "$OUT/$mode$i"/*/*.db 2>/dev/null
  done
done
du -sh "$OUT" | tail -1
