#!/bin/bash
# round 3, batch 33: with decode_union_split as the default for qualifying pairs: GPU suite, bench line, and the union
# workload under rocprofv3 (kernel trace + counter passes) -> profiles/r03_union_split_*
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3
timeout -k 10 900 python3 -m pytest tests -x -q -m gpu > gpurun_out/r3/b33_pytest.log 2>&1 || { tail -30 gpurun_out/r3/b33_pytest.log; exit 1; }
tail -2 gpurun_out/r3/b33_pytest.log
timeout -k 10 300 python3 bench.py > gpurun_out/r3/b33_bench.json 2> gpurun_out/r3/b33_bench.err || { tail -20 gpurun_out/r3/b33_bench.err; exit 1; }
python3 - <<'PY'
import json
d=json.loads(open('gpurun_out/r3/b33_bench.json').read().strip().split('\n')[-1])
print('value %.4g ms %.4f frac %.4f kernel %s' % (d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['kernel']))
for c in d['configs']:
    print('  %-55s %-40s ms %.4f frac %.3f' % (c.get('name', c.get('workload','?'))[:55], str(c.get('kernel'))[:40], c.get('kernel_ms', 0), c.get('frac', 0)))
PY
timeout -k 10 600 bash tools/perf/prof.sh r3_union_split decode_union_split --workload union-concat-500k > gpurun_out/r3/b33_prof.log 2>&1; tail -3 gpurun_out/r3/b33_prof.log
