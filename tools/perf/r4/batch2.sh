#!/bin/bash
# Round 4, batch 2: where the time of the split union and of the one-tile kernel goes, by subtraction
# (measurement build: phases switched off through the `debug` option; tools/perf/ab3.py, A/A control included).
#   1 = no decode, 2 = no output, 4 = no loads (row ids, row records), 0x4000 = no table / codebook copy into LDS
set -o pipefail
out=gpurun_out/r4_batch2
mkdir -p $out
export MEMB_SYNTH_DEVICE=0
python tools/perf/build_measure.py > $out/build.txt 2>&1 || exit 1
export MEMB_PACKAGE_ROOT=build/measure
export AB3='nodecode:debug=1,nooutput:debug=2,noloads:debug=4,nocopy:debug=0x4000,stores_only:debug=5,stores_only_nocopy:debug=0x4005,loads_stores_nocopy:debug=0x4001,loads_only:debug=3'
AB3_CASES=union AB3_ROUNDS=4 timeout -k 10 300 python tools/perf/ab3.py > $out/union_phases.txt 2>&1 || exit 1
tail -16 $out/union_phases.txt
MEMB_HIP_PERSISTENT=0 AB3_CASES=sorted,random,500k AB3_ROUNDS=4 timeout -k 10 300 python tools/perf/ab3.py > $out/onetile_phases_4bit.txt 2>&1 || exit 1
tail -48 $out/onetile_phases_4bit.txt
