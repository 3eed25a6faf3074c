#!/bin/bash
# Round 5, batch 31: decode_union_split's tiles per wavefront a GRID apart (the two-tile pattern's stride) instead of next to
# each other (measurement build, debug = 0x8000), T = 2 and 3, repeated and new batches, by size.
set -o pipefail
out=gpurun_out/r5_union_stride
mkdir -p $out
export MEMB_SYNTH_DEVICE=0 AB3_ROUNDS=3 MEMB_PACKAGE_ROOT=build/measure
for words in 100000 250000 500000 1000000; do
    AB3='apart:debug=0x8000,t3:tiles_per_wave=3,t3apart:tiles_per_wave=3;debug=0x8000' AB3_UNION_WORDS=$words AB3_CASES=union,hbmunion \
        timeout -k 10 300 python tools/perf/ab3.py > $out/union_$words.txt 2>&1 || { tail -20 $out/union_$words.txt; exit 1; }
    echo "== $words words"; sed -n '/--- median/,$p' $out/union_$words.txt | grep -v "^---\|A/A\|base2"
done
