#!/bin/bash
# round 3, batch 26: what bounds the union kernel -- measurement build, decode / output / loads switched off in turn
# (debug bits: 1 = no decode, 2 = no output, 4 = no loads and no row ids, 0x2000 = output without its LDS reads)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3
export AB3_ROUNDS=4 AB3_REPS=20 MEMB_HIP_AUTOTUNE=0 MEMB_PACKAGE_ROOT=build/measure
AB3='nodecode:debug=1,nooutput:debug=2,loadsonly:debug=3,noloads:debug=4,outputonly:debug=5,decodeonly:debug=6,nothing:debug=7,nogather:debug=8192,storesonly:debug=8197,o:persistent=0' AB3_CASES=union,500k timeout -k 10 500 python3 tools/perf/ab3.py > gpurun_out/r3/b26_union_bounds.log 2>&1; sed -n '/^variant/p;/^---/,$p' gpurun_out/r3/b26_union_bounds.log | grep -v "A/A"
