#!/bin/bash
# Round 5: decode_records_persistent's class (2 R < tiles <= 4 R) again, now that the one-tile kernel holds 28 wavefronts per
# CU: rule (base) against one tile per wavefront always (one), cached and with rotating buffers.
set -o pipefail
out=gpurun_out/r5_records_class
mkdir -p $out
export MEMB_SYNTH_DEVICE=0 AB3_ROUNDS=3
cases=70k,85k,100k,114k,120k,130k,rot70k,rot85k,rot100k,rot114k,rot120k,rot130k
for model in "4 2196017" "6 1999995" "2 2196017"; do
    set -- $model
    AB3='one:persistent=0,rec:persistent=2' AB3_BITS=$1 AB3_WORDS=$2 AB3_CASES=$cases \
        timeout -k 10 500 python tools/perf/ab3.py > $out/$1bit.txt 2>&1 || { tail -30 $out/$1bit.txt; exit 1; }
    echo "== $1-bit"; sed -n '/--- median/,$p' $out/$1bit.txt | grep -v "^---\|A/A\|base2"
done
