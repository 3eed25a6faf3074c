#!/bin/bash
# usage: tools/perf/profiles.sh <round tag, e.g. r06> [workload tags: headline 100k union 6bit 2bit uniform]      (repo root, GPU box)
# The rocprofv3 passes (kernel trace + counters, tools/perf/prof.sh) of every workload that has a bench.py --workload, into
# gpurun_out/prof_<round>_<tag>/; `python tools/perf/collect_profiles.py <round> <commit>` then copies the summaries to
# profiles/<round>_* and writes profiles/hbm_traffic.json stamped with the commit and the hash of the kernel sources.
set -o pipefail
round=${1:-r06}
shift
only=" ${*:-headline 100k union 6bit 2bit uniform} "
out=gpurun_out/${round}_profiles
mkdir -p $out
export MEMB_SYNTH_DEVICE=0
for spec in "headline decode_trained<false,+2,+true> " "100k decode_records_persistent --workload+glove840b-300d-4bit-100k" \
            "union decode_union_split --workload+union-concat-500k" "6bit decode_trained<false,+2,+false> --workload+fasttext2m-300d-6bit-fullvocab" \
            "2bit decode_trained<false,+2,+true> --workload+glove840b-300d-2bit-fullvocab" "uniform dequant_uniform_tile --workload+uniform-8bit-500k"; do
    set -- $spec
    case "$only" in *" $1 "*) ;; *) continue ;; esac
    tag=${round}_$1; kernel=${2//+/ }; shift 2   # ('+' stands for a space inside a word of the list above)
    args=${*//+/ }
    echo "== prof $tag ($kernel) $args"
    timeout -k 10 700 bash tools/perf/prof.sh $tag "$kernel" $args > $out/prof_$tag.txt 2>&1 || { tail -20 $out/prof_$tag.txt; exit 1; }
    grep "AverageNs\|traffic_over_algorithmic\|lds_conflict_share\|hbm_traffic_bytes\|'frac'" $out/prof_$tag.txt | head -8
done
