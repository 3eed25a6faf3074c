#!/bin/bash
# round 3, batch 32: decode_union_split (the wavefront's word slots divided between two nibble-key models) against the
# persistent union (n: union_split=0, by size) and the one-tile union (no), at several batch sizes; base = s = split
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_full_size.py -x -q -m gpu -k "union" > gpurun_out/r3/b32_pytest.log 2>&1 || { tail -40 gpurun_out/r3/b32_pytest.log; exit 1; }
tail -2 gpurun_out/r3/b32_pytest.log
export AB3_ROUNDS=4 AB3_REPS=30 AB3_BURST=1 MEMB_HIP_AUTOTUNE=0
for words in 10000 30000 100000 500000 1000000; do
echo "union of $words words"
AB3_UNION_WORDS=$words AB3='s:union_split=2,n:union_split=0,no:union_split=0;persistent=0' AB3_CASES=union timeout -k 10 200 python3 tools/perf/ab3.py > gpurun_out/r3/b32_tmp.log 2>&1; sed -n '/^case/,$p' gpurun_out/r3/b32_tmp.log | grep -v "A/A"; { echo "# union of $words words"; cat gpurun_out/r3/b32_tmp.log; } >> gpurun_out/r3/b32_union_split.log
done
