#!/bin/bash
# round 2, batch 1: cache policies of the output stores / bitstream loads, occupancy via register bounds
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export AB2_ROUNDS=3 AB2_REPS=15
AB2='base:0,nt:16,sc1:32,sc0sc1:48,sc1nt:64,sc0:80,ldnt:256,sc1ldnt:288,allnt:800,w4:0:MEMB_HIP_WAVES=4' \
  timeout -k 10 500 python3 tools/perf/ab2.py > gpurun_out/r2_batch1_policies.log 2>&1 || exit 1
cp memb_amd/libmemb_hip.so /tmp/libmemb_hip_main.so
cp tools/perf/variants/libmemb_hip_b256x5.so memb_amd/libmemb_hip.so
AB2='b5w4:0:MEMB_HIP_WAVES=4,b5w4sc1:32:MEMB_HIP_WAVES=4' AB2_CASES=sorted,random \
  timeout -k 10 300 python3 tools/perf/ab2.py > gpurun_out/r2_batch1_bounds.log 2>&1
cp /tmp/libmemb_hip_main.so memb_amd/libmemb_hip.so
tail -15 gpurun_out/r2_batch1_policies.log; tail -4 gpurun_out/r2_batch1_bounds.log
