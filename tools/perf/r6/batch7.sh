#!/bin/bash
# Round 6, batch 7: the 8-bit model (33 KiB of tables: 16 resident wavefronts per CU in blocks of eight, 14 in blocks of seven, 12 in blocks of four) in both
# orders -- the rule keeps eight for it whatever the order; and tools/perf/r6/dumps.py on the final tree (what each model's order memory picks).
set -o pipefail
out=gpurun_out/r6_batch7
mkdir -p $out
export MEMB_SYNTH_DEVICE=0
AB3_BITS=8 AB3_DIST=student AB3='w7:waves_per_block=7,w4:waves_per_block=4,w8:waves_per_block=8' AB3_CASES=sorted,random,1000k AB3_ROUNDS=3 timeout -k 10 600 python tools/perf/ab3.py > $out/dump_8bit.txt 2>&1 || { tail -30 $out/dump_8bit.txt; exit 1; }
grep "^variant" $out/dump_8bit.txt | cut -c1-150; grep -A7 "^case" $out/dump_8bit.txt | grep -v "^--"
timeout -k 10 200 python tools/perf/r6/dumps.py 2>&1 | grep -v amdgpu.ids | tee $out/dumps.txt
