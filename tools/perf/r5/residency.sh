#!/bin/bash
# Round 5: resident wavefronts per CU of decode_trained by unused LDS (option lds_pad, measurement builds), in the build
# held to 80 scalar registers (up to 32 per CU) and the one with 100 (up to 24).
set -o pipefail
out=gpurun_out/r5_residency
mkdir -p $out
export MEMB_SYNTH_DEVICE=0 AB3_ROUNDS=3
# blocks of four (15.6 KB): 8, 7, 6, 5, 4 blocks per CU; blocks of eight (25 KB): 4, 3, 2
variants='b4x8:waves_per_block=4,b4x7:waves_per_block=4;lds_pad=7400,b4x6:waves_per_block=4;lds_pad=11400,b4x5:waves_per_block=4;lds_pad=16400,b4x4:waves_per_block=4;lds_pad=24400,b8x4:waves_per_block=8,b8x3:waves_per_block=8;lds_pad=25000,b8x2:waves_per_block=8;lds_pad=55000'
for build in measure measure100; do
    for model in "4 2196017" "6 1999995"; do
        set -- $model
        MEMB_PACKAGE_ROOT=build/$build AB3="$variants" AB3_BITS=$1 AB3_WORDS=$2 AB3_CASES=sorted,random,1000k,250k timeout -k 10 400 python tools/perf/ab3.py > $out/${build}_$1bit.txt 2>&1 || { tail -20 $out/${build}_$1bit.txt; exit 1; }
        echo "== $build, $1-bit"; sed -n '/--- median/,$p' $out/${build}_$1bit.txt | grep -v "^---\|A/A"
    done
done
