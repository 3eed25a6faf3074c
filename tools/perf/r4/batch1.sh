#!/bin/bash
# Round 4, batch 1: which kernel owns which configuration.
# {default, one tile per wavefront, records pipeline, general persistent} x {4-bit, 2-bit, 6-bit, byte-key 4-bit, student-t 4-bit}
# x {key order, shuffled, 100 k, 500 k random rows}, on ONE Reader per model (tools/perf/ab3.py, A/A control included).
set -o pipefail
out=gpurun_out/r4_batch1
mkdir -p $out
export MEMB_SYNTH_DEVICE=0
export AB3='onetile:persistent=0,records:persistent=2;pipeline=1,general:persistent=2;pipeline=0'
export AB3_CASES=sorted,random,100k,500k AB3_ROUNDS=4
for model in "4 1234 normal 2196017" "2 1234 normal 2196017" "6 1234 normal 1999995" "4 99 normal 2196017" "4 1234 student 2196017"; do
    set -- $model
    AB3_BITS=$1 AB3_SEED=$2 AB3_DIST=$3 AB3_WORDS=$4 timeout -k 10 240 python tools/perf/ab3.py > $out/kernels_$1bit_seed$2_$3.txt 2>&1 || exit 1
    tail -42 $out/kernels_$1bit_seed$2_$3.txt
done
