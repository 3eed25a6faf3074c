#!/bin/bash
# round 3, batch 37: final tree -- GPU suite, smoke, one complete bench line, soak (4 min)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3
timeout -k 10 900 python3 -m pytest tests -x -q -m gpu > gpurun_out/r3/b37_pytest.log 2>&1 || { tail -30 gpurun_out/r3/b37_pytest.log; exit 1; }
tail -2 gpurun_out/r3/b37_pytest.log
timeout -k 10 300 python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r3/b37_smoke.log 2>&1 || { tail -20 gpurun_out/r3/b37_smoke.log; exit 1; }
tail -1 gpurun_out/r3/b37_smoke.log
timeout -k 10 300 python3 bench.py > gpurun_out/r3/b37_bench.json 2> gpurun_out/r3/b37_bench.err || { tail -20 gpurun_out/r3/b37_bench.err; exit 1; }
python3 - <<'PY'
import json
d=json.loads(open('gpurun_out/r3/b37_bench.json').read().strip().split('\n')[-1])
print('value %.4g ms %.4f frac %.4f kernel %s build %.1f s' % (d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['kernel'], d['model_build_s']))
for c in d['configs']:
    print('  %-55s %-40s ms %.4f frac %.3f' % (c.get('name', c.get('workload','?'))[:55], str(c.get('kernel'))[:40], c.get('kernel_ms', 0), c.get('frac', 0)))
print('host_api', d['host_api']['batch_seconds'], d['host_api']['batch_embeddings_per_s'])
PY
SOAK_SECONDS=240 timeout -k 10 420 python3 tools/perf/soak.py > gpurun_out/r3/b37_soak.log 2>&1; tail -2 gpurun_out/r3/b37_soak.log
