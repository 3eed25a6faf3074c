#!/bin/bash
# Round 5, batch 28: T tiles per wavefront of the one-tile kernel again (seven wavefronts per SIMD now), by block size.
set -o pipefail
out=gpurun_out/r5_t2_now
mkdir -p $out
export MEMB_SYNTH_DEVICE=0 AB3_ROUNDS=3
for model in "4 2196017" "6 1999995"; do
    set -- $model
    AB3='w4:waves_per_block=4,t2w4:tiles_per_wave=2;waves_per_block=4,t2w8:tiles_per_wave=2;waves_per_block=8,t3w4:tiles_per_wave=3;waves_per_block=4,t3w8:tiles_per_wave=3;waves_per_block=8' \
        AB3_BITS=$1 AB3_WORDS=$2 AB3_CASES=sorted,random timeout -k 10 400 python tools/perf/ab3.py > $out/$1bit.txt 2>&1 || { tail -20 $out/$1bit.txt; exit 1; }
    echo "== $1-bit"; sed -n '/--- median/,$p' $out/$1bit.txt | grep -v "^---\|A/A\|base2"
done
