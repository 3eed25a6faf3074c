#!/bin/bash
# round 3, batch 44: final tree again (byte-key unions changed since batch 42): GPU suite + bench line; then, for the
# 65 k - 131 k window where decode_records_persistent runs, grids of tiles / K wavefronts (tiles_per_wave = K) instead of
# one wavefront per resident slot
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3
timeout -k 10 900 python3 -m pytest tests -x -q -m gpu > gpurun_out/r3/b44_pytest.log 2>&1 || { tail -30 gpurun_out/r3/b44_pytest.log; exit 1; }
tail -2 gpurun_out/r3/b44_pytest.log
timeout -k 10 300 python3 bench.py > gpurun_out/r3/b44_bench.json 2> gpurun_out/r3/b44_bench.err || { tail -20 gpurun_out/r3/b44_bench.err; exit 1; }
python3 - <<'PY'
import json
d=json.loads(open('gpurun_out/r3/b44_bench.json').read().strip().split('\n')[-1])
print('value %.4g ms %.4f frac %.4f kernel %s' % (d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['kernel']))
for c in d['configs']:
    print('  %-55s %-40s ms %.4f frac %.3f' % (c.get('name', c.get('workload','?'))[:55], str(c.get('kernel'))[:40], c.get('kernel_ms', 0), c.get('frac', 0)))
PY
export AB3_ROUNDS=4 AB3_REPS=40 AB3_BURST=1 MEMB_HIP_AUTOTUNE=0
AB3='t1:tiles_per_wave=1,t2:tiles_per_wave=2,t3:tiles_per_wave=3,o:persistent=0' AB3_CASES=80k,100k,130k timeout -k 10 300 python3 tools/perf/ab3.py > gpurun_out/r3/b44_tpw.log 2>&1; sed -n '/^variant/p;/^---/,$p' gpurun_out/r3/b44_tpw.log | grep -v "A/A"
