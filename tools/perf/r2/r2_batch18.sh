#!/bin/bash
# round 2, batch 18: is the output phase itself slower than its store pattern? (bit 13: stores of constants, no LDS reads)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(cd tools/perf && timeout -k 10 200 ./worder | tail -8) > gpurun_out/r2_batch18_worder.log 2>&1; cat gpurun_out/r2_batch18_worder.log
export AB2_ROUNDS=3 AB2_REPS=15 AB2_CASES=sorted,random
AB2='base:0,nogather:8192,outonly:5,outonly_nogather:8197,nodecode:1,nodecode_nogather:8193,outonly_w4:5:MEMB_HIP_WAVES=4,outonly_nogather_w4:8197:MEMB_HIP_WAVES=4' timeout -k 10 600 python3 tools/perf/ab2.py > gpurun_out/r2_batch18.log 2>&1 || { tail gpurun_out/r2_batch18.log; exit 1; }
tail -9 gpurun_out/r2_batch18.log
