#!/bin/bash
# Round 5: the finer segment index against the usual one over batch sizes 20k..130k (the rule's upper edge).
set -o pipefail
out=gpurun_out/r5_fine_sweep
mkdir -p $out
export MEMB_SYNTH_DEVICE=0 AB3_ROUNDS=4
for model in "4 2196017" "6 1999995"; do
    set -- $model
    AB3='off:fine_lanes=1,on:fine_lanes=2' AB3_BITS=$1 AB3_WORDS=$2 AB3_CASES=${FINE_SWEEP_CASES:-16k,20k,24k,28k,32k,36k,40k,50k,56k,60k,65k,70k,80k,90k,110k,130k} \
        timeout -k 10 500 python tools/perf/ab3.py > $out/fine_$1bit.txt 2>&1 || { tail -30 $out/fine_$1bit.txt; exit 1; }
    echo "fine index, $1-bit"; grep -v "A/A\|base2" $out/fine_$1bit.txt | tail -120
done
