#!/bin/bash
# round 2, batch 8: first-level width of the packed table (no second level where it fits), copies
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export AB2_ROUNDS=2 AB2_REPS=12 AB2_CASES=sorted,random,100k
for bits in 6 8; do
  export AB2_BITS=$bits
  AB2='default:0,r12:0:MEMB_HIP_BYTE_ROOT_BITS=12,r11:0:MEMB_HIP_BYTE_ROOT_BITS=11,c1:0:MEMB_HIP_TABLE_COPIES=1,w4:0:MEMB_HIP_WAVES=4,nodecode:1,nooutput:2' timeout -k 10 400 python3 tools/perf/ab2.py > gpurun_out/r2_batch8_bits${bits}.log 2>&1 || { tail gpurun_out/r2_batch8_bits${bits}.log; exit 1; }
  echo "bits $bits"; head -8 gpurun_out/r2_batch8_bits${bits}.log | cut -c1-120; tail -8 gpurun_out/r2_batch8_bits${bits}.log
done
