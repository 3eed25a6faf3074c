#!/bin/bash
# Round 5: does the union's T follow rounds? W = 7 168 resident wavefronts, tiles = words / 4; tile-times = T x ceil(tiles / T / W).
# Predicted: T = 1 wins at 58k, 70k, 80k, 115k words (3 against 4, 3 / 4, 3 / 4, 5 / 6), T = 2 wins or ties at 45k, 90k (2 / 2, 4 / 4).
set -o pipefail
out=gpurun_out/r5_union_hbm
mkdir -p $out
export MEMB_SYNTH_DEVICE=0 AB3_ROUNDS=3
for words in 45000 58000 70000 80000 90000 115000; do
    AB3='t1:tiles_per_wave=1,t2:tiles_per_wave=2' AB3_UNION_WORDS=$words AB3_CASES=hbmunion \
        timeout -k 10 300 python tools/perf/ab3.py > $out/union2_$words.txt 2>&1 || { tail -20 $out/union2_$words.txt; exit 1; }
    echo "== $words words"; sed -n '/--- median/,$p' $out/union2_$words.txt | grep -v "^---\|A/A\|base2\|case"
done
