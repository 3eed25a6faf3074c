#!/bin/bash
# round 3, batch 41: multi-symbol table entries for nibble-key lookups (every code that ends inside the next 8 bits per LDS
# lookup) against the build before (build/old_src), same box, one process each, twice; GPU suite first
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3
timeout -k 10 900 python3 -m pytest tests -x -q -m gpu > gpurun_out/r3/b41_pytest.log 2>&1 || { tail -30 gpurun_out/r3/b41_pytest.log; exit 1; }
tail -2 gpurun_out/r3/b41_pytest.log
export AB3_ROUNDS=3 AB3_REPS=40 AB3_BURST=1 MEMB_HIP_AUTOTUNE=0
for bits in 4 2; do
for root in build/old_src . build/old_src .; do
echo "== $bits-bit, package root $root"
AB3_BITS=$bits MEMB_PACKAGE_ROOT=$root AB3='o:persistent=0' AB3_CASES=20k,50k,100k,250k,sorted,union timeout -k 10 300 python3 tools/perf/ab3.py > gpurun_out/r3/b41_tmp.log 2>&1; sed -n '/^---/,$p' gpurun_out/r3/b41_tmp.log | grep -v "A/A\|base2\|^---"; cat gpurun_out/r3/b41_tmp.log >> gpurun_out/r3/b41_multi.log
done
done
