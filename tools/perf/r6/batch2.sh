#!/bin/bash
# Round 6, batch 2: (a) blocks of 7 and 14 wavefronts (28 resident per CU in whole blocks) against 4 and 8, both orders, 4- and 6-bit;
# (b) the term table of the headline kernel by subtraction (measurement build: no decode / no table copy / row ids not loaded);
# (c) the memory patterns at 100 000 rows with nothing cached, and of the dump, on the same box.
set -o pipefail
out=gpurun_out/r6_batch2
mkdir -p $out
export MEMB_SYNTH_DEVICE=0
for bits in 4 6; do
AB3_BITS=$bits AB3_WORDS=$([ $bits = 6 ] && echo 1999995 || echo 2196017) AB3='w4:waves_per_block=4,w7:waves_per_block=7,w14:waves_per_block=14,w6:waves_per_block=6' AB3_CASES=sorted,random,1000k AB3_ROUNDS=4 \
    timeout -k 10 600 python tools/perf/ab3.py > $out/blocks_${bits}bit.txt 2>&1 || { tail -30 $out/blocks_${bits}bit.txt; exit 1; }
grep -A8 "^case" $out/blocks_${bits}bit.txt
done
MEMB_PACKAGE_ROOT=build/measure AB3='nodec:debug=1,nocopy:debug=0x4000,noids:debug=0x10000,nodec_nocopy:debug=0x4001,nodec_noids:debug=0x10001,none:debug=0x14001,noids_nocopy:debug=0x14000,w4:waves_per_block=4,w4none:waves_per_block=4;debug=0x14001' \
    AB3_CASES=sorted,random AB3_ROUNDS=4 timeout -k 10 600 python tools/perf/ab3.py > $out/terms_4bit.txt 2>&1 || { tail -30 $out/terms_4bit.txt; exit 1; }
grep -A13 "^case" $out/terms_4bit.txt
timeout -k 10 300 python tools/perf/r6/rot_ceilings.py > $out/rot_ceilings.txt 2>&1 || { tail -30 $out/rot_ceilings.txt; exit 1; }
cat $out/rot_ceilings.txt
