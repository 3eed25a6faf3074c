#!/bin/bash
# Round 5, batch 4: decode_union_split -- nibble keys through the 4-byte tables (option union_compact: 6 KiB of LDS image
# per block instead of 8, one ds_read_b32 per symbol), blocks of eight wavefronts, both.
set -o pipefail
out=gpurun_out/r5_batch4
mkdir -p $out
export MEMB_SYNTH_DEVICE=0 AB3_ROUNDS=4
MEMB_HIP_UNION_COMPACT=1 timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_words.py -x -q -k "union" > $out/tests.log 2>&1 || { tail -40 $out/tests.log; exit 1; }
tail -2 $out/tests.log
for words in 100000 500000 1000000; do
    AB3='compact:union_compact=1,w8:waves_per_block=8,compact_w8:union_compact=1;waves_per_block=8,w2:waves_per_block=2' AB3_UNION_WORDS=$words AB3_CASES=union \
        timeout -k 10 300 python tools/perf/ab3.py > $out/union_$words.txt 2>&1 || { tail -20 $out/union_$words.txt; exit 1; }
    echo "union of $words words"; sed -n '/--- median/,$p' $out/union_$words.txt | grep -v "^---\|case"
done
