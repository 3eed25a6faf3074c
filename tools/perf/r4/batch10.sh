#!/bin/bash
# Round 4, batch 10: batch 9 again with a baseline -- build/base = the copy in front of the tile's loads (the tree),
# build/v1 = copy overlapped with the loads, inside the T loop, build/v3 = overlapped, straight-line code for T = 1.
set -o pipefail
out=gpurun_out/r4_batch10
mkdir -p $out
export MEMB_SYNTH_DEVICE=0 AB3_ROUNDS=3
for round in 1 2; do
    for root in base v1 v3; do
        MEMB_PACKAGE_ROOT=build/$root AB3_CASES=sorted,random,500k,10k,1k MEMB_HIP_PERSISTENT=0 \
            timeout -k 10 300 python tools/perf/ab3.py > $out/4bit_${root}_$round.txt 2>&1 || exit 1
        echo "round $round build/$root 4-bit"; sed -n '/--- median/,$p' $out/4bit_${root}_$round.txt | grep "case\|base "
    done
done
for root in base v1 v3 base v1 v3; do
    MEMB_PACKAGE_ROOT=build/$root AB3_BITS=6 AB3_WORDS=1999995 AB3_CASES=sorted,random,10k MEMB_HIP_PERSISTENT=0 \
        timeout -k 10 300 python tools/perf/ab3.py > $out/6bit_${root}.txt 2>&1 || exit 1
    echo "build/$root 6-bit"; sed -n '/--- median/,$p' $out/6bit_${root}.txt | grep "case\|base "
done
