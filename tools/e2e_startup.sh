#!/bin/bash
# Runs on the GPU box: where the wall clock of `barbell-amd annotate` goes on a 4 M-read file (process start .. exit), next to its
# steady-state figure.  Usage: tools/e2e_startup.sh [n_reads]   -> stdout
N=${1:-4000000}
REPO=$(cd "$(dirname "$0")/.." && pwd)
export TMPDIR=/tmp
python $REPO/tools/e2e_rate.py $N 4000 --only-kit --json /tmp/e2e.json > /tmp/e2e_rate.log 2>&1   # writes /tmp/e2e.fastq
CLI=$REPO/barbell_amd/bin/barbell-amd
TIMEFORMAT="wall %R s user %U sys %S"
echo "== baseline: process + kit listing (no GPU)"; time $CLI kits > /dev/null
head -c 8040000 /tmp/e2e.fastq > /tmp/tiny.fastq
echo "== baseline: 1000-read file, 1 context"; { time env BARBELL_AMD_PROFILE=1 $CLI annotate -i /tmp/tiny.fastq -o /tmp/t.tsv --kit SQK-NBD114-96 --flank-max-errors 3 --streams 1 ; } 2>&1 | grep -E "profile: start|wall"
for mode in default pinned; do
  for cfg in "2 128" "2 256"; do
    set -- $cfg
    if [ $mode = pinned ]; then export BARBELL_AMD_PINNED_SLOTS=1; else unset BARBELL_AMD_PINNED_SLOTS; fi
    echo "== $mode streams $1 block $2 MiB"
    for i in 1 2; do
      { time env BARBELL_AMD_PROFILE=1 BARBELL_AMD_NO_TORCH=1 $CLI annotate -i /tmp/e2e.fastq -o /tmp/a.tsv --kit SQK-NBD114-96 --flank-max-errors 3 --streams $1 --block-bytes $(($2<<20)) -t 32 ; } 2>&1 | grep -E "profile: start|profile: pipeline|profile: feeder|profile: [0-9.]* s between|wall" | cut -c1-150
    done
  done
done
