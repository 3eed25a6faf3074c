#!/bin/bash
# round 2, batch 19: row records on the 2-bit model (33 % padding: 128-byte regions for rows of 1..3 bits per weight)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export AB2_ROUNDS=3 AB2_REPS=15 AB2_CASES=sorted,coldsorted,random,100k AB2_BITS=2
AB2='compact:0,records:0:MEMB_HIP_RECORD_PADDING_PERCENT=40' timeout -k 10 500 python3 tools/perf/ab2.py > gpurun_out/r2_batch19.log 2>&1 || { tail gpurun_out/r2_batch19.log; exit 1; }
tail -3 gpurun_out/r2_batch19.log
