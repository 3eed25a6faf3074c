#!/bin/bash
# Round 5, batch 28: the records pipeline as a non-resident grid (pipeline_tiles) at five / six / seven wavefronts per SIMD
# (builds build/recb5, the tree, build/recb3) against the one-tile kernel (the same in all three builds), 4-bit model.
set -o pipefail
out=gpurun_out/r5_pipeline_dump
mkdir -p $out
export MEMB_SYNTH_DEVICE=0 AB3_ROUNDS=3
for build in recb5 tree recb3; do
    root=""; [ $build != tree ] && root=build/$build
    MEMB_PACKAGE_ROOT=$root AB3='p2:persistent=2;pipeline_tiles=2,p3:persistent=2;pipeline_tiles=3,p4:persistent=2;pipeline_tiles=4' AB3_CASES=sorted,random \
        timeout -k 10 300 python tools/perf/ab3.py > $out/waves_$build.txt 2>&1 || { tail -20 $out/waves_$build.txt; exit 1; }
    echo "== $build"; sed -n '/--- median/,$p' $out/waves_$build.txt | grep -v "^---\|A/A\|base2"
done
