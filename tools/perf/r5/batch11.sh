#!/bin/bash
# Round 5, batch 11: decode_records_persistent with a balanced grid (every wavefront the same number of tiles) on the
# 65 k - 131 k class, cached and uncached, against the one-tile kernel there.
set -o pipefail
out=gpurun_out/r5_batch11
mkdir -p $out
export MEMB_SYNTH_DEVICE=0 AB3_ROUNDS=4
for model in "4 2196017" "6 1999995" "2 2196017"; do
    set -- $model
    AB3='balanced:balance_grid=1,onetile:persistent=0' AB3_BITS=$1 AB3_WORDS=$2 AB3_CASES=70k,rot70k,85k,rot85k,100k,rot100k,120k,rot120k,130k,rot130k \
        timeout -k 10 500 python tools/perf/ab3.py > $out/balance_$1bit.txt 2>&1 || { tail -30 $out/balance_$1bit.txt; exit 1; }
    echo "$1-bit"; sed -n '/--- median/,$p' $out/balance_$1bit.txt | grep -v "^---\|A/A\|base2"
done
