#!/bin/bash
# dev aid (runs on the GPU box): A/B of the in-tree library against experimental builds exp/lib*.so on the same box, alternating
#   tools/ab_bench.sh [reps] [configs...]
cd "$(dirname "$0")/.."
REPS=${1:-3}; shift; CFGS=${*:-nbd96}
for rep in $(seq $REPS); do
for so in barbell_amd/libbarbell_amd.so exp/lib*.so; do
  for cfg in $CFGS; do
  BARBELL_AMD_SO=$PWD/$so python bench.py --config $cfg --steps 10 --warmup 3 --no-cpu-baseline --no-other-configs --no-policy-variants --no-boundary --no-e2e --no-stress 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']
print('$so'.split('/')[-1], '$cfg', 'reads/s %.1fM' % (d['value']/1e6), 'scan %.2f trace %.2f barcode %.2f' % (k['k_flank_scan'], k['k_flank_trace'], k['k_barcode']), d['roofline'].get('kernel'), '%.2f' % d['roofline'].get('avg_launch_ms'))"
  done
done
done
