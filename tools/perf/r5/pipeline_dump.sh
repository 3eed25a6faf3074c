#!/bin/bash
# Round 5, batch 28: the records pipeline NOT as a resident grid -- every wavefront takes K tiles a grid apart and exits
# (option pipeline_tiles) -- against the one-tile kernel on large batches.
set -o pipefail
out=gpurun_out/r5_pipeline_dump
mkdir -p $out
export MEMB_SYNTH_DEVICE=0 AB3_ROUNDS=3
run() {
    label=$1; shift
    env "$@" AB3='p2:persistent=2;pipeline_tiles=2,p3:persistent=2;pipeline_tiles=3,p4:persistent=2;pipeline_tiles=4,p2w8:persistent=2;pipeline_tiles=2;waves_per_block=8,res:persistent=2' \
        AB3_CASES=sorted,random,1000k,500k,250k timeout -k 10 400 python tools/perf/ab3.py > $out/$label.txt 2>&1 || { tail -20 $out/$label.txt; exit 1; }
    echo "== $label"; sed -n '/--- median/,$p' $out/$label.txt | grep -v "^---\|A/A\|base2"
}
run 4bit AB3_BITS=4
run 6bit AB3_BITS=6 AB3_WORDS=1999995
run 2bit AB3_BITS=2
