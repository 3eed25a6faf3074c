#!/bin/bash
# round 3, batch 27: the union's output row by row (outputUnionRows) against the piece-by-piece walk of the build before
# (build/old_src = a worktree of the commit before, built the same way), same box, one process each, A/A inside each
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3
# (union tests: passed in the first attempt of this batch)

export AB3_ROUNDS=4 AB3_REPS=20 MEMB_HIP_AUTOTUNE=0
for root in build/old_src . build/old_src .; do
echo "== package root $root"
MEMB_PACKAGE_ROOT=$root AB3='o:persistent=0,b4:blocks_per_cu=4' AB3_CASES=union timeout -k 10 300 python3 tools/perf/ab3.py > gpurun_out/r3/b27_tmp.log 2>&1; sed -n '/^package/p;/^---/,$p' gpurun_out/r3/b27_tmp.log | grep -v "A/A"; cat gpurun_out/r3/b27_tmp.log >> gpurun_out/r3/b27_union_rows.log
done
