#!/bin/bash
# Round 5, batch 2: the device word search, second form -- words written once into pinned job regions that the lookup
# kernel reads over PCIe, lookups of finished runs of jobs overlapping the filling of later ones.
set -o pipefail
out=gpurun_out/r5_batch2
mkdir -p $out
export MEMB_SYNTH_DEVICE=0
timeout -k 10 900 python -m pytest tests/test_gpu_words.py -x -q > $out/tests.log 2>&1 || { tail -40 $out/tests.log; exit 1; }
tail -3 $out/tests.log
for threads in 32 16 64 128; do
    echo "MEMB_PACK_THREADS=$threads"
    MEMB_PACK_THREADS=$threads timeout -k 10 300 python tools/perf/r5/words.py --repeats 7 > $out/words_t$threads.txt 2>&1 || { tail -20 $out/words_t$threads.txt; exit 1; }
    grep "stage words\|all\|random" $out/words_t$threads.txt
done
