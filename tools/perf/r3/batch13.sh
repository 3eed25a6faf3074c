#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3
MEMB_HIP_VERBOSE=1 timeout -k 10 900 python3 bench.py --no-cpu-baseline > gpurun_out/r3/b13_bench.json 2> gpurun_out/r3/b13_bench.err; grep "large batches" gpurun_out/r3/b13_bench.err
python3 - <<'PY'
import json
line=[l for l in open('gpurun_out/r3/b13_bench.json') if l.startswith('{')][-1]
d=json.loads(line)
print('headline', d['ms_per_step'], d['roofline']['frac'], d['roofline']['kernel'], d['roofline']['large_batch_timing'])
for c in d['configs']: print('%-50s %-45s %.4f ms frac %.3f %s' % (c['workload'][:50], c['kernel'][:45], c['kernel_ms'], c['frac'], c.get('large_batch_timing')))
PY
