#!/bin/bash
# Runs on the GPU box (through gpurun): the full profile round of ONE bench config at BASELINE size (10 M resident reads, 2 M-read steps) —
# bench line (20 steps), kernel-trace stats, SQ and traffic PMC passes — and the per-config evidence files the bench line reads back:
#   gpurun_out/TAG/{bench_20.json, kernel_stats.csv, pmc_all.txt, traffic_CONFIG.json, valu_CONFIG.json}
#   tools/profile_config.sh CONFIG TAG      (CONFIG = nbd96 | dual | rbk96x | rbk24)
CONFIG=${1:?config}; TAG=${2:?tag}
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp
python $REPO/bench.py --config $CONFIG --steps 20 --warmup 2 --no-cpu-baseline --no-other-configs --no-e2e --no-stress --no-policy-variants --no-boundary > "$OUT/bench_20.json" 2> "$OUT/bench_20.err"
bash $REPO/tools/profile_round.sh $TAG --config $CONFIG > "$OUT/profile_round.log" 2>&1
CMD="python bench.py --config $CONFIG --steps 4 --warmup 1 --no-cpu-baseline --no-other-configs --no-e2e --no-stress --no-policy-variants --no-boundary"
python $REPO/tools/collect_traffic.py "$OUT" "$OUT/traffic_$CONFIG.json" 2000000 4000 "$CMD" > "$OUT/collect.log" 2>&1
python $REPO/tools/collect_valu.py "$OUT/pmc1" "$OUT/pmc2" "$OUT/pmc3" "$OUT/valu_$CONFIG.json" 2000000 4000 >> "$OUT/collect.log" 2>&1
find "$OUT" -name "*.csv" -size +8M -delete
du -sh "$OUT" | tail -1
