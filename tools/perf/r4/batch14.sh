#!/bin/bash
# Round 4, batch 14: the tree as it stands -- full GPU suite, one bench.py run, rocprofv3 summaries of the kernels that changed
# since batch 6 (split union, records pipeline) and of the headline.
out=gpurun_out/r4_batch14
mkdir -p $out
export MEMB_SYNTH_DEVICE=0
timeout -k 10 800 python -m pytest tests -m gpu -q -x --timeout=600 > $out/tests.log 2>&1 || { tail -30 $out/tests.log; exit 1; }
tail -2 $out/tests.log
timeout -k 10 400 python bench.py > $out/bench.json 2> $out/bench.err || { tail -5 $out/bench.err; exit 1; }
bash tools/perf/prof.sh r04b_union decode_union_split --workload union-concat-500k > gpurun_out/prof_r04b_union.txt 2>&1 || exit 1
bash tools/perf/prof.sh r04b_100k decode_records_persistent --workload glove840b-300d-4bit-100k > gpurun_out/prof_r04b_100k.txt 2>&1 || exit 1
bash tools/perf/prof.sh r04b_headline "decode_trained<" > gpurun_out/prof_r04b_headline.txt 2>&1 || exit 1
for t in union 100k headline; do grep -E "^(trace|traffic_over|hbm_traffic_bytes)" gpurun_out/prof_r04b_$t.txt; done
