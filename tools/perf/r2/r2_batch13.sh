#!/bin/bash
# round 2, batch 13: larger tiles (fewer lanes per word) x block barrier
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export AB2_ROUNDS=3 AB2_REPS=12 AB2_CASES=sorted,random,100k
AB2='base:0,l4:0:MEMB_HIP_LANES=4,l2:0:MEMB_HIP_LANES=2,l4sync:8:MEMB_HIP_LANES=4,l2sync:8:MEMB_HIP_LANES=2,l4w4:0:MEMB_HIP_LANES=4;MEMB_HIP_WAVES=4,l2w4:0:MEMB_HIP_LANES=2;MEMB_HIP_WAVES=4,l1w4:0:MEMB_HIP_LANES=1;MEMB_HIP_WAVES=4' timeout -k 10 500 python3 tools/perf/ab2.py > gpurun_out/r2_batch13_tiles.log 2>&1 || { tail gpurun_out/r2_batch13_tiles.log; exit 1; }
head -9 gpurun_out/r2_batch13_tiles.log | cut -c1-110; tail -9 gpurun_out/r2_batch13_tiles.log
