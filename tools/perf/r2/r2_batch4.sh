#!/bin/bash
# round 2, batch 4: what the reads cost inside the real kernel (bit 2: no global reads), warm vs cold caches
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export AB2_ROUNDS=3 AB2_REPS=12 AB2_CASES=sorted,coldsorted,random,coldrandom
AB2='base:0,sc1:32,sc1nt:96,auto0:224,noloads:4,noloads_sc1nt:100,nodecode:1,nodecode_sc1nt:97,nooutput:2' \
  timeout -k 10 600 python3 tools/perf/ab2.py > gpurun_out/r2_batch4_split.log 2>&1 || { tail -20 gpurun_out/r2_batch4_split.log; exit 1; }
tail -11 gpurun_out/r2_batch4_split.log
