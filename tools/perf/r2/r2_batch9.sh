#!/bin/bash
# round 2, batch 9: bank copies of the first-level table and of the codebook, separately
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export AB2_ROUNDS=3 AB2_REPS=12 AB2_CASES=sorted,random,100k
for bits in 6 8; do
  export AB2_BITS=$bits
  AB2='default:0,t1:0:MEMB_HIP_TABLE_COPIES=1,k1:0:MEMB_HIP_CODEBOOK_COPIES=1,t1k1:0:MEMB_HIP_TABLE_COPIES=1;MEMB_HIP_CODEBOOK_COPIES=1,t1k4:0:MEMB_HIP_TABLE_COPIES=1;MEMB_HIP_CODEBOOK_COPIES=4' timeout -k 10 400 python3 tools/perf/ab2.py > gpurun_out/r2_batch9_bits${bits}.log 2>&1 || { tail gpurun_out/r2_batch9_bits${bits}.log; exit 1; }
  echo "bits $bits"; head -6 gpurun_out/r2_batch9_bits${bits}.log | cut -c1-120; tail -6 gpurun_out/r2_batch9_bits${bits}.log
done
