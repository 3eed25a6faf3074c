#!/bin/bash
# Round 4, batch 11: decode_union_split with the copy of tables and codebooks overlapped with the first tile's loads (the
# tree) against the copy in front (measurement build, debug bit 15), one Reader, by batch size; union parity tests first.
set -o pipefail
out=gpurun_out/r4_batch11
mkdir -p $out
export MEMB_SYNTH_DEVICE=0 AB3_ROUNDS=4
timeout -k 10 400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_full_size.py -q -x -k "union" > $out/tests.log 2>&1 || { tail -30 $out/tests.log; exit 1; }
tail -2 $out/tests.log
python tools/perf/build_measure.py > $out/build.txt 2>&1 || exit 1
for words in 10000 30000 100000 250000 500000 1000000; do
    MEMB_PACKAGE_ROOT=build/measure AB3='front:debug=0x8000' AB3_UNION_WORDS=$words AB3_CASES=union \
        timeout -k 10 300 python tools/perf/ab3.py > $out/union_$words.txt 2>&1 || exit 1
    echo "union of $words words"; sed -n '/--- median/,$p' $out/union_$words.txt | grep -v "^---\|case"
done
