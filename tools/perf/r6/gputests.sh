#!/bin/bash
# The whole -m gpu suite in one process, then the default bench.py (the record on stdout, the detail on stderr).
set -o pipefail
out=gpurun_out/r6_gputests
mkdir -p $out
export MEMB_SYNTH_DEVICE=0
timeout -k 10 900 python -m pytest tests -x -q -m gpu --capture=sys > $out/pytest.txt 2>&1 || { tail -40 $out/pytest.txt; exit 1; }
tail -3 $out/pytest.txt
timeout -k 10 400 python bench.py > $out/bench.json 2> $out/bench.err || { tail -20 $out/bench.err; exit 1; }
wc -c $out/bench.json
python - <<'P'
import json
line = json.load(open('gpurun_out/r6_gputests/bench.json'))
print(line['value'], line['roofline']['frac'], line['roofline']['traffic_over_algorithmic'], line['cpu_baseline']['value'], line['parity_vs_cpu_checker'])
for c in line['configs']:
    print('%-70s %-45s %.4f %.3f %s %s' % (c['workload'][:70], c['kernel'], c['kernel_ms'], c['frac'], c.get('repeated_buffer_frac'), c['parity']))
P
