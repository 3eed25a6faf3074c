#!/bin/bash
# round 3, batch 35: host results written with non-temporal stores (host_api of the bench line: words in, numpy out):
# parity test, then the bench line with MEMB_HIP_HOST_STREAMING=0 and =1 on the same box
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py tests/test_host_path.py -x -q -m gpu -k "non_temporal or host" > gpurun_out/r3/b35_pytest.log 2>&1 || { tail -30 gpurun_out/r3/b35_pytest.log; exit 1; }
tail -2 gpurun_out/r3/b35_pytest.log
for mode in 0 1 0 1; do
MEMB_HIP_HOST_STREAMING=$mode timeout -k 10 300 python3 bench.py --no-configs > gpurun_out/r3/b35_bench_$mode.json 2> gpurun_out/r3/b35_bench.err || { tail -20 gpurun_out/r3/b35_bench.err; exit 1; }
python3 - $mode <<'PY'
import json, sys
d=json.loads(open('gpurun_out/r3/b35_bench_%s.json' % sys.argv[1]).read().strip().split('\n')[-1])
h=d['host_api']
print('streaming', sys.argv[1], 'batch_seconds %.4f' % h['batch_seconds'], {k[:28]: round(v,4) for k,v in h['batch_breakdown_seconds'].items()}, 'sample %.4f' % h['sample_seconds'])
PY
done
