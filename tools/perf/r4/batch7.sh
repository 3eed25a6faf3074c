#!/bin/bash
# Round 4, batch 7: decode_union_split with the NEXT tile's row regions in flight during this tile's decode and stores
# (build/measure = the tree) against the tree before it (build/prev), both measurement builds, same box, alternating.
set -o pipefail
out=gpurun_out/r4_batch7
mkdir -p $out
export MEMB_SYNTH_DEVICE=0 AB3_ROUNDS=4
for round in 1 2; do
    for root in prev measure; do
        for words in 250000 500000 1000000; do
            MEMB_PACKAGE_ROOT=build/$root AB3='t2:tiles_per_wave=2,t3:tiles_per_wave=3,t4:tiles_per_wave=4' AB3_UNION_WORDS=$words AB3_CASES=union \
                timeout -k 10 300 python tools/perf/ab3.py > $out/union_${root}_${words}_$round.txt 2>&1 || exit 1
            echo "round $round build/$root union of $words words"; sed -n '/--- median/,$p' $out/union_${root}_${words}_$round.txt | grep -v "A/A\|^---\|case"
        done
    done
done
