#!/bin/bash
# round 3, batch 30: after the size classes of batch 29 went into planTrained: GPU suite, bench line, and the choice by
# size (base) against each kernel forced at the sizes where the class changed
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3
timeout -k 10 900 python3 -m pytest tests -x -q -m gpu > gpurun_out/r3/b30_pytest.log 2>&1 || { tail -30 gpurun_out/r3/b30_pytest.log; exit 1; }
tail -2 gpurun_out/r3/b30_pytest.log
timeout -k 10 300 python3 bench.py > gpurun_out/r3/b30_bench.json 2> gpurun_out/r3/b30_bench.err || { tail -20 gpurun_out/r3/b30_bench.err; exit 1; }
python3 - <<'PY'
import json
d=json.loads(open('gpurun_out/r3/b30_bench.json').read().strip().split('\n')[-1])
print('value %.4g ms %.4f frac %.4f kernel %s' % (d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['kernel']))
for c in d['configs']:
    print('  %-55s %-40s ms %.4f frac %.3f' % (c.get('name', c.get('workload','?'))[:55], str(c.get('kernel'))[:40], c.get('kernel_ms', 0), c.get('frac', 0)))
PY
export AB3_ROUNDS=4 AB3_REPS=40 AB3_BURST=1 MEMB_HIP_AUTOTUNE=0
AB3='o:persistent=0,r:persistent=2;pipeline=1,g:persistent=2;pipeline=0' AB3_CASES=50k,100k,250k,500k timeout -k 10 300 python3 tools/perf/ab3.py > gpurun_out/r3/b30_sizes.log 2>&1; sed -n '/^---/,$p' gpurun_out/r3/b30_sizes.log | grep -v "A/A"
