#!/bin/bash
# Runs on the GPU box (through gpurun): only the FETCH_SIZE / WRITE_SIZE passes of tools/profile_round.sh.
#   tools/traffic_pass.sh TAG [bench args...]   ->  gpurun_out/TAG/{pmc_4.txt, pmc_5.txt, pmc4/, pmc5/}
TAG=${1:-traffic}; shift
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
BENCH="python $REPO/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-other-configs --no-e2e --no-stress --no-policy-variants --no-boundary $*"
cd /tmp
i=3
for set in "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --pmc $set -d "$OUT/pmc$i" -o p --output-format csv -- $BENCH > "$OUT/pmc$i.log" 2>&1
  python $REPO/tools/pmc_summary.py "$OUT/pmc$i" > "$OUT/pmc_$i.txt" 2>&1
  find "$OUT/pmc$i" -name "*.csv" -size +20M -delete
done
python $REPO/tools/collect_traffic.py "$OUT" "$OUT/traffic.json" 2000000 4000 "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- $BENCH"
python - "$OUT/traffic.json" <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); t=0
for k,v in sorted(d['kernels'].items(), key=lambda kv:-kv[1]['hbm_bytes']*kv[1]['launches_per_step']):
    b=v['hbm_bytes']*v['launches_per_step']; t+=b
    print('%-40s %8.3f GB (fetch %.3f, write %.3f)'%(k,b/1e9,2*v['FETCH_SIZE_KiB']*1024/1e9*v['launches_per_step'],v['WRITE_SIZE_KiB']*1024/1e9*v['launches_per_step']))
print('total %.2f GB'%(t/1e9))
PY
