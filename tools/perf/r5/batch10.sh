#!/bin/bash
# Round 5, batch 10: the loader / storer pattern on a second box, the C and C++ clients through the word search, and the
# full bench line with live HBM traffic (rocprofv3 --pmc child passes).
set -o pipefail
out=gpurun_out/r5_batch10
mkdir -p $out
export MEMB_SYNTH_DEVICE=0
timeout -k 10 300 python tools/perf/r5/specialised.py > $out/specialised.txt 2>&1 || { tail -30 $out/specialised.txt; exit 1; }
grep -A8 "pass 1" $out/specialised.txt | head -9; grep " 8       4 " $out/specialised.txt
timeout -k 10 600 python -m pytest tests/test_cabi.py tests/test_cpp_interface.py tests/test_gpu_words.py -x -q > $out/tests.log 2>&1 || { tail -40 $out/tests.log; exit 1; }
tail -2 $out/tests.log
start=$(date +%s)
timeout -k 10 900 python bench.py > $out/bench.json 2> $out/bench.err || { tail -30 $out/bench.err; exit 1; }
echo "bench.py wall: $(( $(date +%s) - start )) s"
python - <<'PY'
import json
line = json.loads(open('gpurun_out/r5_batch10/bench.json').read().strip().splitlines()[-1])
r = line['roofline']
print('value', line['value'], 'frac', r['frac'], 'kernel_avg_ms', r['kernel_avg_ms'])
print('traffic', r['traffic'], r['traffic_over_algorithmic'], r['traffic_recorded_in_profiles'])
print(r['traffic_source'])
for entry in line['configs']:
    print('%-90s ms %.4f frac %.3f' % (entry['workload'][:90], entry['kernel_ms'], entry['frac']))
for b in line['word_search']['batches']:
    print(b['batch'], b['host_ms'], b['device_ms'], b['speedup'])
print(line['host_api']['batch_seconds'], line['host_api']['sample_seconds'])
PY
