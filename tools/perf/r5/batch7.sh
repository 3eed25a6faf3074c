#!/bin/bash
# Round 5, batch 7: the block size of decode_trained for full-size batches, both orders, every model kind (VERDICT r4
# item 2: one table, and a rule that text, code and test agree on).
set -o pipefail
out=gpurun_out/r5_batch7
mkdir -p $out
export MEMB_SYNTH_DEVICE=0 AB3_ROUNDS=4
timeout -k 10 600 python tools/perf/r5/block_sizes.py > $out/geometry.txt 2>&1 || { tail -20 $out/geometry.txt; exit 1; }
cat $out/geometry.txt
run() {   # label, environment...
    label=$1; shift
    env "$@" AB3='w4:waves_per_block=4,w8:waves_per_block=8' AB3_CASES=sorted,random,1000k,500k timeout -k 10 400 python tools/perf/ab3.py > $out/$label.txt 2>&1 || { tail -20 $out/$label.txt; exit 1; }
    echo "$label"; sed -n '/--- median/,$p' $out/$label.txt | grep -v "^---\|A/A\|base2"
}
run 4bit AB3_BITS=4
run 2bit AB3_BITS=2
run 6bit AB3_BITS=6 AB3_WORDS=1999995
run 4bit_bytekeys AB3_BITS=4 AB3_SEED=99
run 4bit_student AB3_BITS=4 AB3_DIST=student
run 8bit AB3_BITS=8
