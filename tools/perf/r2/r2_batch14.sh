#!/bin/bash
# round 2, batch 14: the adopted / rejected variants again, every timing run in for 20 ms (steady power state)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export AB2_ROUNDS=3 AB2_REPS=15 AB2_CASES=sorted,coldsorted,random,100k
AB2='base:0,sc1nt:96,sc0sc1:48,nt:64,sc1:32,ldnt:256,sync:8,l4:0:MEMB_HIP_LANES=4,l4sync:8:MEMB_HIP_LANES=4,nodecode:1,outonly:5' timeout -k 10 700 python3 tools/perf/ab2.py > gpurun_out/r2_batch14_steady.log 2>&1 || { tail gpurun_out/r2_batch14_steady.log; exit 1; }
tail -12 gpurun_out/r2_batch14_steady.log
