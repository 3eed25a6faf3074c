#!/bin/bash
# Round 6, batch 10: several symbols per table lookup for nibble-key models (build/multi: the tree compiled with -DMEMB_HIP_MULTI_SYMBOL=1).
# (1) parity: the soak's lookups of every kind against the CPU checker with that build; (2) timing: the tree and build/multi as alternating processes.
set -o pipefail
out=gpurun_out/r6_multi
mkdir -p $out
export MEMB_SYNTH_DEVICE=0
repo=$(pwd)
( cd /tmp && PYTHONPATH=$repo/build/multi:$repo SOAK_SECONDS=150 SOAK_SEED=77 timeout -k 10 400 python $repo/tools/perf/soak.py > $repo/$out/soak_multi.txt 2>&1 ) || { tail -20 $out/soak_multi.txt; exit 1; }
tail -2 $out/soak_multi.txt
for round in 1 2; do
  for root in "" build/multi; do
    for bits in 4 2; do
      echo "== package '${root:-tree}' ${bits}-bit, round $round"
      MEMB_PACKAGE_ROOT=$root AB3_BITS=$bits AB3='' AB3_CASES=sorted,random,hbm100k,100k,hbm60k,20k,10k,1k,500k AB3_ROUNDS=2 timeout -k 10 300 python tools/perf/ab3.py 2>&1 | grep "^  base \|^case" | paste - - | awk '{print $2, $4}' | tr '\n' ' ' | tee -a $out/ab_${bits}bit.txt
      echo | tee -a $out/ab_${bits}bit.txt
    done
  done
done
