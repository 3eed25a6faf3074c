#!/bin/bash
# Round 5: the rocprofv3 passes (kernel trace + counters) of every workload, for profiles/r05_* (tools/perf/collect_profiles.py
# copies them there).
set -o pipefail
out=gpurun_out/r5_profiles
mkdir -p $out
export MEMB_SYNTH_DEVICE=0
for spec in "r05_headline decode_trained<false,+2,+true> " "r05_100k decode_records_persistent --workload+glove840b-300d-4bit-100k" \
            "r05_union decode_union_split --workload+union-concat-500k" "r05_6bit decode_trained<false,+2,+false> --workload+fasttext2m-300d-6bit-fullvocab" \
            "r05_2bit decode_trained<false,+2,+true> --workload+glove840b-300d-2bit-fullvocab" "r05_uniform dequant_uniform_tile --workload+uniform-8bit-500k"; do
    set -- $spec
    tag=$1; kernel=${2//+/ }; shift 2   # ('+' stands for a space inside a word of the list above)
    args=${*//+/ }
    echo "== prof $tag ($kernel) $args"
    timeout -k 10 700 bash tools/perf/prof.sh $tag "$kernel" $args > $out/prof_$tag.txt 2>&1 || { tail -20 $out/prof_$tag.txt; exit 1; }
    grep "AverageNs\|traffic_over_algorithmic\|lds_conflict_share\|hbm_traffic_bytes\|'frac'" $out/prof_$tag.txt | head -8
done
