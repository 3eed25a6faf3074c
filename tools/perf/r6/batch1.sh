#!/bin/bash
# Round 6, batch 1 (ran on commit 7d9a76a; the kernel and its option value are gone since: this file is the record of how it was measured):
# decode_two_tiles (option persistent = 3) -- parity first, then against the rule's kernels:
# key-order dump, shuffled dump, 100 000 rows (cached / nothing cached), blocks of 4 and 8.
set -o pipefail
out=gpurun_out/r6_batch1
mkdir -p $out
export MEMB_SYNTH_DEVICE=0
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "every_kernel_of_a_model" > $out/parity.txt 2>&1 || { tail -30 $out/parity.txt; exit 1; }
tail -3 $out/parity.txt
AB3='two:persistent=3,two8:persistent=3;waves_per_block=8,two4:persistent=3;waves_per_block=4,one4:waves_per_block=4' AB3_CASES=sorted,random,100k,hbm100k,hbm60k,500k AB3_ROUNDS=4 \
    timeout -k 10 900 python tools/perf/ab3.py > $out/ab_4bit.txt 2>&1 || { tail -30 $out/ab_4bit.txt; exit 1; }
tail -25 $out/ab_4bit.txt
