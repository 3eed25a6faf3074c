#!/bin/bash
set -o pipefail
out=gpurun_out/r5_batch12
mkdir -p $out
export MEMB_SYNTH_DEVICE=0
timeout -k 10 900 python -m pytest tests/test_bench_contract.py -x -q -m gpu > $out/tests.log 2>&1 || { tail -40 $out/tests.log; exit 1; }
tail -2 $out/tests.log
timeout -k 10 900 python bench.py --no-live-traffic > $out/bench.json 2> $out/bench.err || { tail -30 $out/bench.err; exit 1; }
python - <<'PY'
import json
line = json.loads(open('gpurun_out/r5_batch12/bench.json').read().strip().splitlines()[-1])
u = [e for e in line['configs'] if 'configs[4]' in e['workload']][0]
print(u['kernel_ms'], u['frac'], json.dumps(u['from_words']))
PY
