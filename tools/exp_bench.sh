#!/bin/bash
# dev aid (runs on the GPU box): times the barcode stage of experimental library builds barbell_amd/exp/lib_*.so
cd "$(dirname "$0")/.."
for so in barbell_amd/libbarbell_amd.so barbell_amd/exp/lib_*.so; do
  BARBELL_AMD_SO=$PWD/$so python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-other-configs --no-policy-variants --no-boundary --no-e2e --no-stress --reads 4000000 "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']
print('$so'.split('/')[-1], 'reads/s %.1fM' % (d['value']/1e6), 'scan %.2f trace %.2f barcode %.2f' % (k['k_flank_scan'], k['k_flank_trace'], k['k_barcode']))"
done
