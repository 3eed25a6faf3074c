#!/bin/bash
# round 2, batch 2: full matrix of store cache policies x non-temporal stream loads; per-tile policy choice
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export AB2_ROUNDS=3 AB2_REPS=15
AB2='base:0,sc0:16,sc1:32,sc0sc1:48,nt:64,sc0nt:80,sc1nt:96,all:112,L:256,Lsc0:272,Lsc1:288,Lsc0sc1:304,Lnt:320,Lsc0nt:336,Lsc1nt:352,Lall:368,auto0:224,auto3:3296,Lauto0:480,M:512,LM:768' \
  timeout -k 10 800 python3 tools/perf/ab2.py > gpurun_out/r2_batch2_policies.log 2>&1 || exit 1
tail -24 gpurun_out/r2_batch2_policies.log
