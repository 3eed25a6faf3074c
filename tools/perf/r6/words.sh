set -o pipefail; mkdir -p gpurun_out/r6_words; export MEMB_SYNTH_DEVICE=0
timeout -k 10 400 python bench.py --extras --no-configs --no-cpu-baseline --no-live-traffic > gpurun_out/r6_words/line.json 2> gpurun_out/r6_words/err.txt || { tail gpurun_out/r6_words/err.txt; exit 1; }
python - <<'P'
import json
e=json.load(open('gpurun_out/bench_detail.json'))['extras']['word_search']
for b in e['batches']:
    print(b['batch'], 'host %.3f device %.3f (fill %.3f kernel %.3f) packed %.3f (fill %.3f)' % (b['host_ms'], b['device_ms'], b['device_breakdown_ms']['strings -> pinned memory alone (no lookup)'], b['device_breakdown_ms']['resolve_words over the whole batch alone (reads the words over PCIe)'], b['device_ms_from_packed_words'], b['packed_words_to_pinned_memory_alone_ms']))
P
