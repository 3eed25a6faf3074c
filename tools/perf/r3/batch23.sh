#!/bin/bash
# round 3, batch 23: what the driver runs at round end, in its order: GPU suite with -x, smoke, the default bench line
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3
if [ -z "$B23_BENCH_ONLY" ]; then
timeout -k 10 900 python3 -m pytest tests -x -q -m gpu > gpurun_out/r3/b23_pytest.log 2>&1; tail -4 gpurun_out/r3/b23_pytest.log
timeout -k 10 300 python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
fi
start=$(date +%s); python3 bench.py > gpurun_out/r3/b23_bench.json 2> gpurun_out/r3/b23_bench.err; echo "bench.py wall $(( $(date +%s) - start )) s"; tail -c 300 gpurun_out/r3/b23_bench.err
python3 - <<'PY'
import json
line=[l for l in open('gpurun_out/r3/b23_bench.json') if l.startswith('{')][-1]
d=json.loads(line)
print('headline', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['kernel'], 'model_build_s', d['model_build_s'], d['parity_vs_cpu_checker'], d['roofline']['large_batch_timing']['chosen'])
for c in d['configs']: print('%-55s %-45s %.4f ms frac %.3f %s' % (c['workload'][:55], c['kernel'][:45], c['kernel_ms'], c['frac'], c['parity']))
print(d['cpu_baseline']['value'], d['cpu_baseline']['kind'], d['host_api']['batch_seconds'])
PY
