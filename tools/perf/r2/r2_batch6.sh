#!/bin/bash
# round 2, batch 6: one-shot kernel (52 VGPRs, 32 waves/CU) vs the persistent pipeline on byte-key models
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export AB2_ROUNDS=2 AB2_REPS=12 AB2_CASES=sorted,random,100k
for bits in 6 8 4; do
  export AB2_BITS=$bits
  AB2='base:0,oneshot:0:MEMB_HIP_PERSISTENT=0,oneshot4:0:MEMB_HIP_PERSISTENT=0;MEMB_HIP_WAVES=4,l16:0:MEMB_HIP_LANES=16,oneshotl16:0:MEMB_HIP_PERSISTENT=0;MEMB_HIP_LANES=16,l4:0:MEMB_HIP_LANES=4' timeout -k 10 400 python3 tools/perf/ab2.py > gpurun_out/r2_batch6_bits${bits}.log 2>&1 || { tail gpurun_out/r2_batch6_bits${bits}.log; exit 1; }
  echo "bits $bits"; tail -7 gpurun_out/r2_batch6_bits${bits}.log
done
