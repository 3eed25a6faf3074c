#!/bin/bash
# Round 5, batch 14: threads of the str walk by batch size (one per 32 768 words, 8 .. 64) against fixed pools.
set -o pipefail
out=gpurun_out/r5_batch14
mkdir -p $out
export MEMB_SYNTH_DEVICE=0
for run in 1 2; do
for threads in 64 32 16; do
    echo "MEMB_PACK_THREADS=$threads (pool size; active threads by batch size)"
    MEMB_PACK_THREADS=$threads timeout -k 10 300 python tools/perf/r5/words.py --repeats 7 > $out/words_t${threads}_$run.txt 2>&1 || { tail -20 $out/words_t${threads}_$run.txt; exit 1; }
    grep "all\|random" $out/words_t${threads}_$run.txt | cut -c1-150
done
done
