#!/bin/bash
# round 3, batch 34: store width. The guide quotes 6.0-6.2 TB/s for dword stores (256 B per wave instruction); the output
# phase stores 1 KiB per instruction (dwordx4). Measurement build, key-order dump and 500 k rows: the same bytes as
# four 256-byte (w1: debug bit 14) or two 512-byte (w2: bit 15) store instructions, with the whole kernel (..) and
# with stores only (s..: + no loads, no decode, no LDS reads)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3
export AB3_ROUNDS=4 AB3_REPS=20 MEMB_HIP_AUTOTUNE=0 MEMB_PACKAGE_ROOT=build/measure
AB3='w1:debug=16384,w2:debug=32768,s4:debug=8197,s1:debug=24581,s2:debug=40965,o:persistent=0,ow1:persistent=0;debug=16384,os4:persistent=0;debug=8197,os1:persistent=0;debug=24581' AB3_CASES=sorted,500k timeout -k 10 500 python3 tools/perf/ab3.py > gpurun_out/r3/b34_store_width.log 2>&1; sed -n '/^---/,$p' gpurun_out/r3/b34_store_width.log | grep -v "A/A\|differs"
