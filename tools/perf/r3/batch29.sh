#!/bin/bash
# round 3, batch 29: which kernel for which batch size, 20 k .. 250 k random rows (shipped build, burst timing):
# o = one tile per wavefront, r = decode_records_persistent, g = the general persistent kernel; base = the choice by size
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3
export AB3_ROUNDS=4 AB3_REPS=40 AB3_BURST=1 MEMB_HIP_AUTOTUNE=0
for bits in 4 2 6; do
AB3_BITS=$bits AB3='o:persistent=0,r:persistent=2;pipeline=1,g:persistent=2;pipeline=0' AB3_CASES=20k,35k,50k,65k,80k,100k,130k,160k,250k timeout -k 10 400 python3 tools/perf/ab3.py > gpurun_out/r3/b29_sizes_bits$bits.log 2>&1; echo "bits $bits"; sed -n '/^---/,$p' gpurun_out/r3/b29_sizes_bits$bits.log | grep -v "A/A"
done
