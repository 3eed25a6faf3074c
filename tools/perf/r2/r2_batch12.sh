#!/bin/bash
# round 2, batch 12: full on/off matrix (decode, loads, block barrier) + write-pattern calibration on the same box
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(cd tools/perf && timeout -k 10 120 ./wpattern 2>&1 | head -12) > gpurun_out/r2_batch12_wpattern.log
cat gpurun_out/r2_batch12_wpattern.log
export AB2_ROUNDS=3 AB2_REPS=12 AB2_CASES=sorted,random
AB2='base:0,nodecode:1,noloads:4,outonly:5,sync:8,nodecsync:9,noloadsync:12,outonlysync:13' timeout -k 10 500 python3 tools/perf/ab2.py > gpurun_out/r2_batch12_matrix.log 2>&1 || { tail gpurun_out/r2_batch12_matrix.log; exit 1; }
tail -9 gpurun_out/r2_batch12_matrix.log
