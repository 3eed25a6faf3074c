#!/bin/bash
# Round 5: the two rules of the round on another box -- the finer segment index by batch size, block sizes by order.
set -o pipefail
out=gpurun_out/r5_confirm
mkdir -p $out
export MEMB_SYNTH_DEVICE=0 AB3_ROUNDS=4
for model in "4 2196017" "6 1999995"; do
    set -- $model
    AB3='off:fine_lanes=1,on:fine_lanes=2' AB3_BITS=$1 AB3_WORDS=$2 AB3_CASES=1k,10k,20k,30k,50k,100k \
        timeout -k 10 400 python tools/perf/ab3.py > $out/fine_$1bit.txt 2>&1 || { tail -30 $out/fine_$1bit.txt; exit 1; }
    echo "fine index, $1-bit"; sed -n '/--- median/,$p' $out/fine_$1bit.txt | grep -v "^---\|A/A\|base2"
done
run() {
    label=$1; shift
    env "$@" AB3='w4:waves_per_block=4,w8:waves_per_block=8' AB3_CASES=sorted,random,1000k,500k timeout -k 10 400 python tools/perf/ab3.py > $out/$label.txt 2>&1 || { tail -20 $out/$label.txt; exit 1; }
    echo "block size, $label"; sed -n '/--- median/,$p' $out/$label.txt | grep -v "^---\|A/A\|base2"
}
run 4bit AB3_BITS=4
run 6bit AB3_BITS=6 AB3_WORDS=1999995
run 4bit_bytekeys AB3_BITS=4 AB3_SEED=99
