#!/bin/bash
# round 2, batch 10: final packed-table build -- parity suite, then all models
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 1100 python3 -m pytest tests -m gpu -x -q > gpurun_out/r2_gpu_tests_b.log 2>&1 || { tail -40 gpurun_out/r2_gpu_tests_b.log; exit 1; }
tail -3 gpurun_out/r2_gpu_tests_b.log
export AB2_ROUNDS=2 AB2_REPS=12 AB2_CASES=sorted,random,100k
for bits in 6 8; do
  export AB2_BITS=$bits
  AB2='default:0,nodecode:1,nooutput:2' timeout -k 10 400 python3 tools/perf/ab2.py > gpurun_out/r2_batch10_bits${bits}.log 2>&1 || { tail gpurun_out/r2_batch10_bits${bits}.log; exit 1; }
  echo "bits $bits"; tail -4 gpurun_out/r2_batch10_bits${bits}.log
done
