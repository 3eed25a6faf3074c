#!/bin/bash
# Round 4, batch 5: (a) the split union's default against T = 1 / 2 forced (is the rule in effect?),
# (b) the 65 k - 131 k class with NOTHING cached between launches: four batches round-robin into four output buffers
#     ('rot<N>k' cases of tools/perf/ab3.py) -- does decode_records_persistent still own it when rows and output are in HBM?
set -o pipefail
out=gpurun_out/r4_batch5
mkdir -p $out
export MEMB_SYNTH_DEVICE=0 AB3_ROUNDS=4
AB3='t1:tiles_per_wave=1,t2:tiles_per_wave=2,t3:tiles_per_wave=3' AB3_CASES=union timeout -k 10 300 python tools/perf/ab3.py > $out/union_default.txt 2>&1 || exit 1
sed -n '/--- median/,$p' $out/union_default.txt
for model in "4 2196017" "6 1999995" "2 2196017"; do
    set -- $model
    AB3='onetile:persistent=0,records:persistent=2,onetile_w8:persistent=0;waves_per_block=8,onetile_w2:persistent=0;waves_per_block=2' \
        AB3_BITS=$1 AB3_WORDS=$2 AB3_CASES=100k,rot100k,rot70k,rot130k,rot200k timeout -k 10 300 python tools/perf/ab3.py > $out/rotating_$1bit.txt 2>&1 || exit 1
    sed -n '/--- median/,$p' $out/rotating_$1bit.txt
done
