#!/bin/bash
# round 2, batch 5: byte-key models (6-bit, 8-bit): where the time goes, and occupancy via register bounds
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export AB2_ROUNDS=2 AB2_REPS=12 AB2_CASES=sorted,random
cp memb_amd/libmemb_hip.so /tmp/libmemb_hip_main.so
for bits in 6 8; do
  export AB2_BITS=$bits
  cp /tmp/libmemb_hip_main.so memb_amd/libmemb_hip.so
  AB2='base:0,noloads:4,nodecode:1,nooutput:2,w4:0:MEMB_HIP_WAVES=4' timeout -k 10 400 python3 tools/perf/ab2.py > gpurun_out/r2_batch5_bits${bits}_main.log 2>&1 || { tail gpurun_out/r2_batch5_bits${bits}_main.log; exit 1; }
  tail -6 gpurun_out/r2_batch5_bits${bits}_main.log
  for w in 5 6; do
    cp tools/perf/variants/libmemb_hip_b256x$w.so memb_amd/libmemb_hip.so
    AB2="b${w}w4:0:MEMB_HIP_WAVES=4" timeout -k 10 300 python3 tools/perf/ab2.py > gpurun_out/r2_batch5_bits${bits}_b$w.log 2>&1
    tail -1 gpurun_out/r2_batch5_bits${bits}_b$w.log
  done
done
cp /tmp/libmemb_hip_main.so memb_amd/libmemb_hip.so
