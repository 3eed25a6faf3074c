#!/bin/bash
# Round 5, batch 28: where does the records pipeline (two tiles per wavefront, non-resident grid) lose what its pattern gains?
# Measurement build: the one-tile kernel and the pipeline with the decode skipped (debug = 1) and with constants stored (0x2000).
set -o pipefail
out=gpurun_out/r5_pipeline_phases
mkdir -p $out
export MEMB_SYNTH_DEVICE=0 AB3_ROUNDS=3 MEMB_PACKAGE_ROOT=build/measure
AB3='nodecode:debug=1,nogather:debug=0x2001,p2:persistent=2;pipeline_tiles=2,p2nodecode:persistent=2;pipeline_tiles=2;debug=1,p2nogather:persistent=2;pipeline_tiles=2;debug=0x2001,w4nodecode:waves_per_block=4;debug=1' \
    AB3_CASES=sorted,random timeout -k 10 400 python tools/perf/ab3.py > $out/4bit.txt 2>&1 || { tail -20 $out/4bit.txt; exit 1; }
sed -n '/--- median/,$p' $out/4bit.txt | grep -v "^---\|A/A\|base2"
