#!/bin/bash
# round 3, batch 7: staggered start of the persistent wavefronts (measurement build): are the one-tile kernel's wins on
# the 2- and 6-bit dumps a matter of phases in step?
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3
export AB3_ROUNDS=4 AB3_REPS=20 MEMB_PACKAGE_ROOT=build/measure
V='stagw:debug=0x18000,stagbw:debug=0x30000,stagbw2:debug=0x34000,onetile:persistent=0'
for bits in 2 4 6; do
AB3_BITS=$bits AB3=$V AB3_CASES=sorted,random,250k timeout -k 10 400 python3 tools/perf/ab3.py > gpurun_out/r3/b7_stagger_bits$bits.log 2>&1 || { tail -30 gpurun_out/r3/b7_stagger_bits$bits.log; exit 1; }
echo "bits $bits"; sed -n '/^---/,$p' gpurun_out/r3/b7_stagger_bits$bits.log | grep -v "A/A"
done
