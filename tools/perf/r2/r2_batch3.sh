#!/bin/bash
# round 2, batch 3: line touches ahead of the real bitstream loads (key-order dump), uniform and in bursts
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export AB2_ROUNDS=3 AB2_REPS=15 AB2_CASES=sorted
AB2='base:0,sc1nt:96,st1f2:2162784,st1f3:3211360,st1f4:4259936,st1f6:6357088,st4f2:2359392,st6f2:2490464,st6f4:4587616,st12f2:2883680,pt1f2:2162688,pt6f2:2490368' \
  timeout -k 10 600 python3 tools/perf/ab2.py > gpurun_out/r2_batch3_touch.log 2>&1 || { tail -20 gpurun_out/r2_batch3_touch.log; exit 1; }
tail -14 gpurun_out/r2_batch3_touch.log
