#!/bin/bash
# Round 5: the two-model union's rule for tiles per wavefront (T = 2 from 16 000 words on) with nothing cached between
# launches, by size: rule / T = 1 / T = 2 / blocks of eight.
set -o pipefail
out=gpurun_out/r5_union_hbm
mkdir -p $out
export MEMB_SYNTH_DEVICE=0 AB3_ROUNDS=3
for words in 5000 10000 16000 30000 60000 100000 250000; do
    AB3='t1:tiles_per_wave=1,t2:tiles_per_wave=2,t3:tiles_per_wave=3' AB3_UNION_WORDS=$words AB3_CASES=union,hbmunion \
        timeout -k 10 300 python tools/perf/ab3.py > $out/union_$words.txt 2>&1 || { tail -20 $out/union_$words.txt; exit 1; }
    echo "== $words words"; sed -n '/--- median/,$p' $out/union_$words.txt | grep -v "^---\|A/A\|base2"
done
