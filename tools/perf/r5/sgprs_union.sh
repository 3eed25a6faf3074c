#!/bin/bash
# Round 5: the two-model union (decode_union_split) under the scalar-register budget, alternating processes; then the
# finer segment index over batch sizes with the budget in place.
set -o pipefail
out=gpurun_out/r5_sgprs
mkdir -p $out
export MEMB_SYNTH_DEVICE=0 AB3_ROUNDS=3
for pass in 1 2 3; do
    for build in new old; do
        root=""; [ $build = old ] && root=build/sgpr100
        MEMB_PACKAGE_ROOT=$root AB3= AB3_CASES=union timeout -k 10 300 python tools/perf/ab3.py > $out/union_${build}_$pass.txt 2>&1 || { tail -20 $out/union_${build}_$pass.txt; exit 1; }
        echo "union $build pass $pass: $(sed -n '/--- median/,$p' $out/union_${build}_$pass.txt | grep '  base ')"
    done
done
bash tools/perf/r5/fine_sweep.sh > $out/fine_sweep.txt 2>&1 || { tail -20 $out/fine_sweep.txt; exit 1; }
