#!/bin/bash
# round 3, batch 16: one-tile kernel, only the row ids before the table copy (debug bit 20) against the default
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3
export AB3_ROUNDS=4 AB3_REPS=20 MEMB_PACKAGE_ROOT=build/measure MEMB_HIP_PERSISTENT=0
for bits in 2 4; do
AB3_BITS=$bits AB3='rowearly:debug=0x100000,allearly:debug=0x40000' AB3_CASES=sorted,random,100k,10k timeout -k 10 300 python3 tools/perf/ab3.py > gpurun_out/r3/b16_rowearly_bits$bits.log 2>&1; echo "bits $bits"; sed -n '/^---/,$p' gpurun_out/r3/b16_rowearly_bits$bits.log | grep -v "A/A"
done
