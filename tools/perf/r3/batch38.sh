#!/bin/bash
# round 3, batch 38: decode_union_split for byte-key pairs (two 6-bit models, 4-byte PACKED tables) against the forms
# before (n: union_split=0 by size, no: one tile per wavefront), 30 k / 100 k / 500 k words
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3
export AB3_ROUNDS=4 AB3_REPS=30 AB3_BURST=1 MEMB_HIP_AUTOTUNE=0 AB3_BITS=6 AB3_UNION_BITS=6 MEMB_SYNTH_DEVICE=0
for words in 30000 100000 500000; do
echo "union of $words words, two 6-bit models"
AB3_UNION_WORDS=$words AB3='s:union_split=1,n:union_split=0,no:union_split=0;persistent=0' AB3_CASES=union timeout -k 10 400 python3 tools/perf/ab3.py > gpurun_out/r3/b38_tmp.log 2>&1; sed -n '/^case/,$p' gpurun_out/r3/b38_tmp.log | grep -v "A/A"; { echo "# union of $words words, two 6-bit models"; cat gpurun_out/r3/b38_tmp.log; } >> gpurun_out/r3/b38_union_split_bytes.log
done
