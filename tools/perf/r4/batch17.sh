#!/bin/bash
# Round 4, batch 17: pieces a lane gathers before it stores them back to back in outputTile (MEMB_HIP_OUTPUT_BURST; 5 in the
# tree): builds with 3, 4, 6, 8 against the tree, alternating processes on one box.
set -o pipefail
out=gpurun_out/r4_batch17
mkdir -p $out
export MEMB_SYNTH_DEVICE=0 AB3_ROUNDS=3
for round in 1 2; do
    for root in measure burst3 burst4 burst6 burst8; do
        MEMB_PACKAGE_ROOT=build/$root AB3_CASES=sorted,random,500k,100k timeout -k 10 300 python tools/perf/ab3.py > $out/${root}_$round.txt 2>&1 || exit 1
        echo "round $round build/$root: $(sed -n '/--- median/,$p' $out/${root}_$round.txt | grep "base " | awk '{printf "%s ", $2}')"
    done
done
