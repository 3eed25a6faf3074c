#!/bin/bash
# Round 5, batch 1: the device word search (new GPU tests, host vs device timings with phases) and a first look at
# more lanes per word on the small batch classes (layouts that exist today: 13 lanes per word on the compact layout
# against 8 lanes with and without row records).
set -o pipefail
out=gpurun_out/r5_batch1
mkdir -p $out
export MEMB_SYNTH_DEVICE=0
timeout -k 10 900 python -m pytest tests/test_gpu_words.py -x -q > $out/tests.log 2>&1 || { tail -40 $out/tests.log; exit 1; }
tail -3 $out/tests.log
timeout -k 10 300 python tools/perf/r5/words.py > $out/words.txt 2>&1 || { tail -30 $out/words.txt; exit 1; }
cat $out/words.txt
for threads in 8 16 64; do
    echo "MEMB_PACK_THREADS=$threads MEMB_HIP_PACK_THREADS=$threads"
    MEMB_PACK_THREADS=$threads MEMB_HIP_PACK_THREADS=$threads timeout -k 10 300 python tools/perf/r5/words.py --repeats 5 > $out/words_t$threads.txt 2>&1 || exit 1
    grep "all\|100 000" $out/words_t$threads.txt
done
AB3='c8:!MEMB_HIP_ROW_RECORDS=0,l16:!MEMB_HIP_LANES=16' AB3_CASES=1k,10k,50k,100k,rot100k AB3_ROUNDS=4 \
    timeout -k 10 400 python tools/perf/ab3.py > $out/lanes_4bit.txt 2>&1 || { tail -30 $out/lanes_4bit.txt; exit 1; }
sed -n '/--- median/,$p' $out/lanes_4bit.txt
