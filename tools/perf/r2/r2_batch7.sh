#!/bin/bash
# round 2, batch 7: packed 4-byte table with bank-replicated first level + codebook (byte-key models)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py -x -q > gpurun_out/r2_batch7_parity.log 2>&1 || { tail -30 gpurun_out/r2_batch7_parity.log; exit 1; }
tail -3 gpurun_out/r2_batch7_parity.log
export AB2_ROUNDS=2 AB2_REPS=12 AB2_CASES=sorted,random,100k
for bits in 6 8; do
  export AB2_BITS=$bits
  AB2='c32:0,c1:0:MEMB_HIP_TABLE_COPIES=1,c8:0:MEMB_HIP_TABLE_COPIES=8,r10:0:MEMB_HIP_BYTE_ROOT_BITS=10,r11c1:0:MEMB_HIP_BYTE_ROOT_BITS=11;MEMB_HIP_TABLE_COPIES=1,r6:0:MEMB_HIP_BYTE_ROOT_BITS=6,nodecode:1,nooutput:2' timeout -k 10 400 python3 tools/perf/ab2.py > gpurun_out/r2_batch7_bits${bits}.log 2>&1 || { tail gpurun_out/r2_batch7_bits${bits}.log; exit 1; }
  echo "bits $bits"; grep DIFFERS gpurun_out/r2_batch7_bits${bits}.log; tail -9 gpurun_out/r2_batch7_bits${bits}.log
done
