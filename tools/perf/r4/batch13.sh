#!/bin/bash
# Round 4, batch 13: decode_records_persistent with the copy of table and codebook overlapped with the first tile's row
# regions (the tree) against the copy in front (measurement build, debug bit 15); parity tests of the kernel first.
set -o pipefail
out=gpurun_out/r4_batch13
mkdir -p $out
export MEMB_SYNTH_DEVICE=0 AB3_ROUNDS=4
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_full_size.py -q -x -k "every_kernel or default_path or ragged or randomized or layouts" > $out/tests.log 2>&1 || { tail -30 $out/tests.log; exit 1; }
tail -2 $out/tests.log
python tools/perf/build_measure.py > $out/build.txt 2>&1 || exit 1
for model in "4 2196017" "6 1999995" "2 2196017"; do
    set -- $model
    MEMB_PACKAGE_ROOT=build/measure AB3='front:debug=0x8000' AB3_BITS=$1 AB3_WORDS=$2 AB3_CASES=100k,rot100k,70k,130k \
        timeout -k 10 300 python tools/perf/ab3.py > $out/records_$1bit.txt 2>&1 || exit 1
    echo "$1-bit"; sed -n '/--- median/,$p' $out/records_$1bit.txt | grep -v "^---"
done
