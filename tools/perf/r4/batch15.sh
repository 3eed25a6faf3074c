#!/bin/bash
# Round 4, batch 15: instruction priority (s_setprio 3) for the output phase / the decode of decode_trained (measurement
# build, debug bits 16 / 17), one Reader.
set -o pipefail
out=gpurun_out/r4_batch15
mkdir -p $out
export MEMB_SYNTH_DEVICE=0 AB3_ROUNDS=4 MEMB_PACKAGE_ROOT=build/measure
AB3='prio_out:debug=0x10000,prio_dec:debug=0x20000,prio_both:debug=0x30000' AB3_CASES=sorted,random,500k,10k \
    timeout -k 10 300 python tools/perf/ab3.py > $out/prio_4bit.txt 2>&1 || exit 1
sed -n '/--- median/,$p' $out/prio_4bit.txt | grep -v "A/A"
AB3='prio_out:debug=0x10000,prio_dec:debug=0x20000' AB3_BITS=6 AB3_WORDS=1999995 AB3_CASES=sorted,random \
    timeout -k 10 300 python tools/perf/ab3.py > $out/prio_6bit.txt 2>&1 || exit 1
sed -n '/--- median/,$p' $out/prio_6bit.txt | grep -v "A/A"
