#!/bin/bash
# Round 4, batch 12: T tiles per wavefront of the split union again, now that the copy is overlapped with the first tile's loads.
set -o pipefail
out=gpurun_out/r4_batch12
mkdir -p $out
export MEMB_SYNTH_DEVICE=0 AB3_ROUNDS=4
for words in 30000 100000 160000 250000 500000 1000000; do
    AB3='t1:tiles_per_wave=1,t2:tiles_per_wave=2,t3:tiles_per_wave=3,t4:tiles_per_wave=4' AB3_UNION_WORDS=$words AB3_CASES=union \
        timeout -k 10 300 python tools/perf/ab3.py > $out/union_$words.txt 2>&1 || exit 1
    echo "union of $words words"; sed -n '/--- median/,$p' $out/union_$words.txt | grep -v "^---\|case"
done
