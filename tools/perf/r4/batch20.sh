#!/bin/bash
# Round 4, batch 20: four copies of the lookup table in LDS, one per quarter of the wavefront (option table_copies), in
# decode_trained and decode_records_persistent -- round 2 found bank copies useless for dumps (the decode is hidden there);
# the classes in which the decode is NOT hidden (10 k - 130 k rows) had never been asked.
set -o pipefail
out=gpurun_out/r4_batch20
mkdir -p $out
export MEMB_SYNTH_DEVICE=0 AB3_ROUNDS=4
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -q -x -k "every_kernel or ragged" > $out/tests.log 2>&1 || { tail -20 $out/tests.log; exit 1; }
for model in "4 2196017" "6 1999995" "2 2196017"; do
    set -- $model
    AB3='copies:table_copies=1' AB3_BITS=$1 AB3_WORDS=$2 AB3_CASES=10k,50k,100k,rot100k,250k,500k,sorted,random \
        timeout -k 10 300 python tools/perf/ab3.py > $out/copies_$1bit.txt 2>&1 || exit 1
    echo "$1-bit"; sed -n '/--- median/,$p' $out/copies_$1bit.txt | grep -v "^---\|A/A\|base2"
done
