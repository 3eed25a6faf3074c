#!/bin/bash
# round 2, batch 17: tile size and block barrier again, on the row-record layout (steady state)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export AB2_ROUNDS=3 AB2_REPS=15 AB2_CASES=sorted,coldsorted,random,100k
AB2='base:0,sync:8,l4:0:MEMB_HIP_LANES=4,l4sync:8:MEMB_HIP_LANES=4,w4:0:MEMB_HIP_WAVES=4,w4sync:8:MEMB_HIP_WAVES=4,nodecode:1,outonly:5' timeout -k 10 600 python3 tools/perf/ab2.py > gpurun_out/r2_batch17.log 2>&1 || { tail gpurun_out/r2_batch17.log; exit 1; }
tail -9 gpurun_out/r2_batch17.log
