#!/bin/bash
# Round 4, batch 4 (another box): full GPU suite on the T-loop kernels, then
# (a) blocks of 8 / 16 wavefronts for the one-tile kernel on dumps (batch 3: -4 % on the 4-bit key-order dump, +1 % shuffled),
# (b) every XCD one contiguous run of the batch (measurement build, debug bit 15),
# (c) T tiles per wavefront in the split union by batch size.
set -o pipefail
out=gpurun_out/r4_batch4
mkdir -p $out
export MEMB_SYNTH_DEVICE=0
if [ -z "$SKIP_TESTS" ]; then
timeout -k 10 800 python -m pytest tests -m gpu -q -x --timeout=600 > $out/tests.log 2>&1 || { tail -30 $out/tests.log; exit 1; }
tail -2 $out/tests.log
fi
export AB3_ROUNDS=4
for model in "4 1234" "2 1234" "4 99"; do
    set -- $model
    AB3='onetile:persistent=0,w8:persistent=0;waves_per_block=8,w16:persistent=0;waves_per_block=16,w2:persistent=0;waves_per_block=2' \
        AB3_BITS=$1 AB3_SEED=$2 AB3_CASES=sorted,random,1000k timeout -k 10 300 python tools/perf/ab3.py > $out/blocks_$1bit_seed$2.txt 2>&1 || exit 1
    sed -n '/--- median/,$p' $out/blocks_$1bit_seed$2.txt
done
AB3='onetile:persistent=0,w8:persistent=0;waves_per_block=8,w16:persistent=0;waves_per_block=16' \
    AB3_BITS=6 AB3_WORDS=1999995 AB3_CASES=sorted,random timeout -k 10 300 python tools/perf/ab3.py > $out/blocks_6bit.txt 2>&1 || exit 1
sed -n '/--- median/,$p' $out/blocks_6bit.txt
for words in 30000 100000 250000 500000 1000000; do
    AB3='t2:tiles_per_wave=2,t3:tiles_per_wave=3,t4:tiles_per_wave=4' AB3_UNION_WORDS=$words AB3_CASES=union \
        timeout -k 10 300 python tools/perf/ab3.py > $out/union_tiles_$words.txt 2>&1 || exit 1
    echo "union of $words words"; sed -n '/--- median/,$p' $out/union_tiles_$words.txt
done
python tools/perf/build_measure.py > $out/build.txt 2>&1 || exit 1
MEMB_PACKAGE_ROOT=build/measure AB3='xcd:persistent=0;debug=0x8000,w8:persistent=0;waves_per_block=8,w8xcd:persistent=0;waves_per_block=8;debug=0x8000,onetile:persistent=0' \
    AB3_CASES=sorted,random timeout -k 10 300 python tools/perf/ab3.py > $out/xcd_4bit.txt 2>&1 || exit 1
sed -n '/--- median/,$p' $out/xcd_4bit.txt
