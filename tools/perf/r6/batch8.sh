#!/bin/bash
# Round 6, batch 8: block sizes of the split union (configs[4]: 500 000 words, two 4-bit models) -- four (the rule), six, seven, eight -- repeated buffer and nothing cached.
set -o pipefail
out=gpurun_out/r6_batch8
mkdir -p $out
export MEMB_SYNTH_DEVICE=0
AB3='w7:waves_per_block=7,w8:waves_per_block=8,w6:waves_per_block=6,w5:waves_per_block=5' AB3_CASES=union,hbmunion AB3_ROUNDS=4 timeout -k 10 600 python tools/perf/ab3.py > $out/union.txt 2>&1 || { tail -30 $out/union.txt; exit 1; }
grep -A8 "^case" $out/union.txt | grep -v "^--"
