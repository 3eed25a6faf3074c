#!/bin/bash
# Round 5, batch 9: what decode_union_split's time is made of (measurement build, phases switched off in turn), and the
# rocprofv3 passes (kernel trace + counters) of every workload for profiles/r05_*.
set -o pipefail
out=gpurun_out/r5_batch9
mkdir -p $out
export MEMB_SYNTH_DEVICE=0 AB3_ROUNDS=3
for words in 500000; do
    MEMB_PACKAGE_ROOT=build/measure AB3='nodecode:debug=1,nocopy:debug=0x4000,memonly:debug=0x4001,storesonly:debug=0x4005,loadsonly:debug=0x4003,noloads:debug=4,nooutput:debug=2' \
        AB3_UNION_WORDS=$words AB3_CASES=union timeout -k 10 400 python tools/perf/ab3.py > $out/union_phases_$words.txt 2>&1 || { tail -20 $out/union_phases_$words.txt; exit 1; }
    echo "union of $words words, phases"; sed -n '/--- median/,$p' $out/union_phases_$words.txt | grep -v "^---\|case"
done
python - <<'PY'
import ctypes, os, sys
sys.path.insert(0, os.getcwd())
import torch, bench
timer = bench.Timer(torch)
merged = torch.empty((500000, 600), dtype=torch.float32, device='cuda')
out = torch.empty((2196017, 300), dtype=torch.float32, device='cuda')
c = bench.box_ceilings(torch, timer, out, 2196017, union=(merged, 500000), patterns=(7,))
print('union_tile_fill_random_records on this box: %.4f ms' % c['union_tile_fill_random_records']['ms'])
PY
for spec in "r05_headline decode_trained<false,_2,_true>" "r05_100k decode_records_persistent --workload_glove840b-300d-4bit-100k" \
            "r05_union decode_union_split --workload_union-concat-500k" "r05_6bit decode_trained<false,_2,_false> --workload_fasttext2m-300d-6bit-fullvocab" \
            "r05_2bit decode_trained<false,_2,_true> --workload_glove840b-300d-2bit-fullvocab" "r05_uniform dequant_uniform_tile --workload_uniform-8bit-500k"; do
    set -- $spec
    tag=$1; kernel=${2//_/ }; shift 2
    args=${*//_/ }
    echo "== prof $tag ($kernel) $args"
    timeout -k 10 700 bash tools/perf/prof.sh $tag "$kernel" $args > $out/prof_$tag.txt 2>&1 || { tail -20 $out/prof_$tag.txt; exit 1; }
    grep "AverageNs\|traffic_over_algorithmic\|lds_conflict_share\|hbm_traffic_bytes\|'frac'" $out/prof_$tag.txt | head -8
done
