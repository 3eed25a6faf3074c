#!/bin/bash
# round 3, batch 28: where the time of the small batches goes (100 k / 10 k / 1 k random rows): measurement build,
# phases switched off in turn (debug bits: 1 = no decode, 2 = no output, 4 = no loads / row ids, 0x2000 = no LDS reads in the output)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3
export AB3_ROUNDS=4 AB3_REPS=40 AB3_BURST=1 MEMB_HIP_AUTOTUNE=0 MEMB_PACKAGE_ROOT=build/measure
AB3='nodecode:debug=1,nooutput:debug=2,loadsonly:debug=3,noloads:debug=4,outputonly:debug=5,nothing:debug=7,storesonly:debug=8197,o:persistent=0,p:persistent=2' AB3_CASES=100k,50k,10k,1k timeout -k 10 500 python3 tools/perf/ab3.py > gpurun_out/r3/b28_small_bounds.log 2>&1; sed -n '/^variant/p;/^---/,$p' gpurun_out/r3/b28_small_bounds.log | grep -v "A/A\|differs"
