#!/bin/bash
# Round 4, batch 18: block sizes between four and sixteen wavefronts for the dumps (one Reader per model).
set -o pipefail
out=gpurun_out/r4_batch18
mkdir -p $out
export MEMB_SYNTH_DEVICE=0 AB3_ROUNDS=4
for model in "4 1234" "2 1234"; do
    set -- $model
    AB3='w4:persistent=0;waves_per_block=4,w5:persistent=0;waves_per_block=5,w6:persistent=0;waves_per_block=6,w7:persistent=0;waves_per_block=7,w10:persistent=0;waves_per_block=10,w12:persistent=0;waves_per_block=12' \
        AB3_BITS=$1 AB3_SEED=$2 AB3_CASES=sorted,random timeout -k 10 300 python tools/perf/ab3.py > $out/blocks_$1bit.txt 2>&1 || exit 1
    echo "$1-bit (base = the rule: eight)"; sed -n '/--- median/,$p' $out/blocks_$1bit.txt | grep -v "^---\|A/A"
done
