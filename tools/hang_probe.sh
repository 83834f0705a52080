#!/bin/bash
# dev aid: reruns the gzip-parts kit command until it hangs, then dumps the threads' stacks
cd "$(dirname "$0")/.."
python - <<'PY'
import os, sys
sys.path.insert(0, ".")
from barbell_amd import annotate as A, kits
sys.path.insert(0, "tests")
from test_cli import write_fastq
groups = kits.groups_from_kit("SQK-NBD114-96", flank_max_errors=3)
n = 1300
bases, offsets = A.synth_reads_host(groups, 555, 300, 2200, 0, n)
ids = [f"r{i}" for i in range(n)]
os.makedirs("/tmp/hp", exist_ok=True)
cuts = [0, 200, 200, 650, 651, 1000, n]
for k in range(len(cuts) - 1):
    a, b = cuts[k], cuts[k + 1]
    write_fastq(f"/tmp/hp/part{k}.fastq.gz", ids[a:b], bases[int(offsets[a]):int(offsets[b])], offsets[a:b + 1] - offsets[a], gz=True)
write_fastq("/tmp/hp/all.fastq", ids, bases, offsets)
PY
for i in $(seq 1 30); do
  for inp in "/tmp/hp/all.fastq" "/tmp/hp/part0.fastq.gz /tmp/hp/part1.fastq.gz /tmp/hp/part2.fastq.gz /tmp/hp/part3.fastq.gz /tmp/hp/part4.fastq.gz /tmp/hp/part5.fastq.gz"; do
  rm -rf /tmp/hp/out
  BARBELL_AMD_NO_TORCH=1 barbell_amd/bin/barbell-amd kit -k SQK-NBD114-96 -i $inp -o /tmp/hp/out --flank-max-errors 3 --maximize -t 4 --batch-reads 100 > /tmp/hp/log.txt 2>&1 &
  pid=$!
  for s in $(seq 1 100); do sleep 0.2; kill -0 $pid 2>/dev/null || break; done
  if kill -0 $pid 2>/dev/null; then
    echo "HANG at iteration $i input=$inp"
    gdb -batch -ex "thread apply all bt 12" -p $pid 2>/dev/null | grep -E "^Thread|^#" | grep -v "^#1[0-9]" | head -150
    kill -9 $pid
    exit 0
  fi
  wait $pid || { echo "rc=$? at $i"; tail -3 /tmp/hp/log.txt; }
  done
done
echo "no hang in 30 iterations"
