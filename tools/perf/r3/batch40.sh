#!/bin/bash
# round 3, batch 40: skewed start in decode_records_persistent (measurement build): half the wavefronts begin their first
# decode 0.8 / 1.7 / 2.5 / 3.4 us late (w.: the odd wavefronts of a block = bit 16, b.: the odd blocks = bit 17), 65 k - 130 k rows
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3
export AB3_ROUNDS=4 AB3_REPS=40 AB3_BURST=1 MEMB_HIP_AUTOTUNE=0 MEMB_PACKAGE_ROOT=build/measure
AB3='w1:debug=65536,w2:debug=81920,w3:debug=98304,w4:debug=114688,b1:debug=131072,b2:debug=147456,b3:debug=163840,b4:debug=180224' AB3_CASES=80k,100k,130k timeout -k 10 500 python3 tools/perf/ab3.py > gpurun_out/r3/b40_skew.log 2>&1; sed -n '/^---/,$p' gpurun_out/r3/b40_skew.log | grep -v "A/A\|differs"
