#!/bin/bash
# Round 4, batch 9: decode_trained with the table copy overlapped with the tile's dependent loads (build/measure = the tree)
# against the tree before it (build/prev), both measurement builds, same box, alternating; full GPU suite first.
set -o pipefail
out=gpurun_out/r4_batch9
mkdir -p $out
export MEMB_SYNTH_DEVICE=0 AB3_ROUNDS=4
timeout -k 10 800 python -m pytest tests -m gpu -q -x --timeout=600 > $out/tests.log 2>&1 || { tail -30 $out/tests.log; exit 1; }
tail -2 $out/tests.log
for round in 1 2; do
    for root in prev measure; do
        MEMB_PACKAGE_ROOT=build/$root AB3='onetile:persistent=0' AB3_CASES=sorted,random,500k,100k,10k,1k \
            timeout -k 10 300 python tools/perf/ab3.py > $out/4bit_${root}_$round.txt 2>&1 || exit 1
        echo "round $round build/$root 4-bit"; sed -n '/--- median/,$p' $out/4bit_${root}_$round.txt | grep "case\|base \|onetile"
    done
done
for root in prev measure prev measure; do
    MEMB_PACKAGE_ROOT=build/$root AB3='onetile:persistent=0' AB3_BITS=6 AB3_WORDS=1999995 AB3_CASES=sorted,random,500k,10k \
        timeout -k 10 300 python tools/perf/ab3.py > $out/6bit_${root}.txt 2>&1 || exit 1
    echo "build/$root 6-bit"; sed -n '/--- median/,$p' $out/6bit_${root}.txt | grep "case\|base \|onetile"
    MEMB_PACKAGE_ROOT=build/$root AB3='onetile:persistent=0' AB3_BITS=2 AB3_CASES=sorted,random,10k \
        timeout -k 10 300 python tools/perf/ab3.py > $out/2bit_${root}.txt 2>&1 || exit 1
    echo "build/$root 2-bit"; sed -n '/--- median/,$p' $out/2bit_${root}.txt | grep "case\|base \|onetile"
done
