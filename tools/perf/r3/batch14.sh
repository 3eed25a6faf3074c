#!/bin/bash
# round 3, batch 14: one-tile kernel with its loads issued before the table copy (default) against behind it (debug bit 18)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3
export AB3_ROUNDS=4 AB3_REPS=20 MEMB_PACKAGE_ROOT=build/measure MEMB_HIP_PERSISTENT=0
for bits in 2 4 6; do
AB3_BITS=$bits AB3='late:debug=0x40000' AB3_CASES=sorted,random,100k,10k,1k timeout -k 10 300 python3 tools/perf/ab3.py > gpurun_out/r3/b14_early_bits$bits.log 2>&1; echo "bits $bits"; sed -n '/^---/,$p' gpurun_out/r3/b14_early_bits$bits.log | grep -v "A/A"
done
