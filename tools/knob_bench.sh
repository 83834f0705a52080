#!/bin/bash
# dev aid (GPU box): bench the default library under several settings of one environment knob: knob_bench.sh VAR v1 v2 ...
cd "$(dirname "$0")/.."
VAR=$1; shift
for v in "$@"; do
  env $VAR=$v python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-other-configs --no-e2e --no-policy-variants --no-boundary --reads 4000000 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']
print('$VAR=$v', 'reads/s %.1fM' % (d['value']/1e6), 'scan %.2f trace %.2f barcode %.2f' % (k['k_flank_scan'], k['k_flank_trace'], k['k_barcode']))"
done
