#!/bin/bash
# dev aid (runs on the GPU box): HBM fetch bytes per kernel (FETCH_SIZE pass) of experimental library builds barbell_amd/exp/lib_*.so
cd "$(dirname "$0")/.."; export TMPDIR=/tmp
for so in barbell_amd/libbarbell_amd.so barbell_amd/exp/lib_*.so; do
  n=$(basename $so .so); rm -rf /tmp/tr_$n
  ( cd /tmp && BARBELL_AMD_SO=$OLDPWD/$so rocprofv3 --pmc FETCH_SIZE -d /tmp/tr_$n -o p --output-format csv -- python $OLDPWD/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-other-configs --no-policy-variants --no-boundary --no-e2e --no-stress --reads 4000000 > /dev/null 2>&1 )
  echo "== $n"; python tools/pmc_summary.py /tmp/tr_$n | grep -A1 -E "k_flank_verify|k_flank_trace|k_flank_filter" | grep -v "^--"
done
