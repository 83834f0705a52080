#!/bin/bash
# Runs on the GPU box (through gpurun): kernel-trace stats + PMC passes of the default bench workload.
#   tools/profile_round.sh TAG [bench args...]
# Output: gpurun_out/TAG/{bench.json, kernel_stats.csv, pmc_*.txt, counters.txt}
# Counters are collected in their own runs (never together with a trace), FETCH_SIZE and WRITE_SIZE in separate
# passes (TCC slots), as /opt/skills/guides/MI355X_MICROARCH.md prescribes.
TAG=${1:-prof}; shift
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
BENCH="python $REPO/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-other-configs --no-e2e --no-stress --no-policy-variants --no-boundary $*"
cd /tmp
$BENCH > "$OUT/bench.json" 2> "$OUT/bench.err"
rocprofv3 --kernel-trace --stats -d "$OUT/trace" -o t --output-format csv -- $BENCH > "$OUT/trace.log" 2>&1
f=$(find "$OUT/trace" -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" "$OUT/kernel_stats.csv"
rocprofv3 -L > "$OUT/counters.txt" 2>&1
i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" \
           "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM SQ_ACTIVE_INST_SCA" \
           "SQ_INST_CYCLES_VMEM SQ_INSTS_SMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --pmc $set -d "$OUT/pmc$i" -o p --output-format csv -- $BENCH > "$OUT/pmc$i.log" 2>&1
  python $REPO/tools/pmc_summary.py "$OUT/pmc$i" > "$OUT/pmc_$i.txt" 2>&1
  find "$OUT/pmc$i" -name "*.csv" -size +20M -delete
done
cat "$OUT"/pmc_*.txt > "$OUT/pmc_all.txt"
rm -rf "$OUT/trace"/*/*.db 2>/dev/null
du -sh "$OUT" | tail -1
