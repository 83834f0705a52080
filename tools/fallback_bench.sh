#!/bin/bash
# Runs on the GPU box: what the fallback kernels cost on the headline workload (reachable through odd geometry, env knobs, the lane
# kernel's back-off): reads/s and stage times of bench.py with each forced.  -> stdout, one line per variant
cd "$(dirname "$0")/.."
run() {
  env "$@" python bench.py --steps 3 --warmup 1 --reads 4000000 --no-cpu-baseline --no-other-configs --no-e2e --no-stress --no-policy-variants --no-boundary 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']
print('%-34s %6.1f M reads/s  scan %5.2f trace %5.2f barcode %6.2f  dominant %s' % ('$*', d['value']/1e6, k['k_flank_scan'], k['k_flank_trace'], k['k_barcode'], d['roofline']['kernel']))"
}
run BARBELL_AMD_X=default
run BARBELL_AMD_LANE=0
run BARBELL_AMD_NO_FAST=1
run BARBELL_AMD_NO_PFX=1
run BARBELL_AMD_GENERIC=1
run BARBELL_AMD_SCAN_FILTER=0
run BARBELL_AMD_TRACE_FULL=1
run BARBELL_AMD_NO_SIDE_STREAM=1
