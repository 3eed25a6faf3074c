#!/bin/bash
# Round 6, batch 3: blocks of seven wavefronts (four whole blocks = the 28 resident wavefronts of a CU) on a second box, every
# model kind, both orders, and down the batch sizes; the records pipeline's class in blocks of six (24 resident = four blocks).
set -o pipefail
out=gpurun_out/r6_batch3
mkdir -p $out
export MEMB_SYNTH_DEVICE=0
run() {   # name, extra env...
    name=$1; shift
    env "$@" timeout -k 10 500 python tools/perf/ab3.py > $out/$name.txt 2>&1 || { tail -30 $out/$name.txt; exit 1; }
    echo "== $name"; grep -A7 "^case" $out/$name.txt | grep -v "^--"
}
run sizes_4bit AB3='w7:waves_per_block=7,w4:waves_per_block=4,w8:waves_per_block=8,one7:persistent=0;waves_per_block=7,rec6:persistent=2;waves_per_block=6' AB3_CASES=sorted,random,500k,250k,hbm130k,hbm100k,hbm60k,100k AB3_ROUNDS=3
run dump_2bit AB3_BITS=2 AB3='w7:waves_per_block=7,w4:waves_per_block=4' AB3_CASES=sorted,random,500k AB3_ROUNDS=3
run dump_bytekeys AB3_SEED=99 AB3='w7:waves_per_block=7,w4:waves_per_block=4,w6:waves_per_block=6' AB3_CASES=sorted,random,500k AB3_ROUNDS=3
run dump_student AB3_DIST=student AB3='w7:waves_per_block=7,w4:waves_per_block=4' AB3_CASES=sorted,random AB3_ROUNDS=3
run dump_6bit AB3_BITS=6 AB3_WORDS=1999995 AB3='w7:waves_per_block=7,w6:waves_per_block=6,w4:waves_per_block=4,w5:waves_per_block=5' AB3_CASES=sorted,random,500k,hbm100k AB3_ROUNDS=3
