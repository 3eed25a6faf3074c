#!/bin/bash
# Round 4, batch 6: rocprofv3 summaries of the final tree -- headline (with the box-ceiling kernels in the same trace),
# configs[2] (6-bit dump), configs[3] (2-bit dump), configs[4] (union), configs[1] (100 k rows).
out=gpurun_out
export MEMB_SYNTH_DEVICE=0
bash tools/perf/prof.sh r04_headline "decode_trained<" > $out/prof_r04_headline.txt 2>&1 || exit 1
tail -4 $out/prof_r04_headline.txt
bash tools/perf/prof.sh r04_6bit "decode_trained<" --workload fasttext2m-300d-6bit-fullvocab > $out/prof_r04_6bit.txt 2>&1 || exit 1
bash tools/perf/prof.sh r04_union decode_union_split --workload union-concat-500k > $out/prof_r04_union.txt 2>&1 || exit 1
bash tools/perf/prof.sh r04_2bit "decode_trained<" --workload glove840b-300d-2bit-fullvocab > $out/prof_r04_2bit.txt 2>&1 || exit 1
bash tools/perf/prof.sh r04_100k decode_records_persistent --workload glove840b-300d-4bit-100k > $out/prof_r04_100k.txt 2>&1 || exit 1
for t in headline 6bit union 2bit 100k; do grep -E "^(trace|traffic_over|hbm_traffic_bytes|lds_conflict)" $out/prof_r04_$t.txt; done
