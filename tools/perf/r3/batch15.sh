#!/bin/bash
# round 3, batch 15: persistent kernels with the row ids loaded before the table copy (default since batch 2) against behind it
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3
export AB3_ROUNDS=4 AB3_REPS=20 MEMB_PACKAGE_ROOT=build/measure MEMB_HIP_AUTOTUNE=0
for bits in 4 2; do
AB3_BITS=$bits AB3='rowslate:debug=0x80000' AB3_CASES=sorted,random,250k,100k,50k timeout -k 10 300 python3 tools/perf/ab3.py > gpurun_out/r3/b15_rows_bits$bits.log 2>&1; echo "bits $bits"; sed -n '/^---/,$p' gpurun_out/r3/b15_rows_bits$bits.log | grep -v "A/A"
done
