#!/bin/bash
# round 3, batch 31: the fused union at other batch sizes: one tile per wavefront (o) against the persistent form (p);
# base = the choice by size (persistent from 2 tiles per resident wavefront = 65 k words on)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3
export AB3_ROUNDS=4 AB3_REPS=30 AB3_BURST=1 MEMB_HIP_AUTOTUNE=0
for words in 30000 60000 100000 200000 1000000; do
echo "union of $words words"
AB3_UNION_WORDS=$words AB3='o:persistent=0,p:persistent=2' AB3_CASES=union timeout -k 10 200 python3 tools/perf/ab3.py > gpurun_out/r3/b31_tmp.log 2>&1; sed -n '/^case/,$p' gpurun_out/r3/b31_tmp.log | grep -v "A/A"; { echo "# union of $words words"; cat gpurun_out/r3/b31_tmp.log; } >> gpurun_out/r3/b31_union_sizes.log
done
