#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
echo "== N=1 default"; timeout -k 10 900 python3 bench.py > gpurun_out/r2_bench_n1.json 2> gpurun_out/r2_bench_n1.err || { tail -30 gpurun_out/r2_bench_n1.err; exit 1; }
python3 -c "
import json; d=json.load(open('gpurun_out/r2_bench_n1.json'))
print('value %.4g frac %.3f ms %.4f parity %s build %.1fs' % (d['value'], d['roofline']['frac'], d['roofline']['kernel_avg_ms'], d['parity_vs_cpu_checker'], d['model_build_s']))
for c in d['configs']: print('  %-60s %.4f ms frac %.3f %s' % (c['workload'][:60], c['kernel_ms'], c['frac'], c['parity']))
print(json.dumps(d['host_api'], indent=1))
"
echo "== N=2 rehearsal (both ranks on cuda:0, gloo), self-launched"
MEMB_BENCH_REHEARSAL=1 timeout -k 10 600 python3 bench.py --gpus 2 --small --steps 3 --warmup 1 > gpurun_out/r2_bench_n2_rehearsal.json 2> gpurun_out/r2_bench_n2_rehearsal.err || { tail -30 gpurun_out/r2_bench_n2_rehearsal.err; exit 1; }
python3 -c "
import json; d=json.loads([l for l in open('gpurun_out/r2_bench_n2_rehearsal.json') if l.startswith('{')][-1])
print('n_gpus', d['n_gpus'], 'ranks_seen', d['ranks_seen'], 'value %.4g' % d['value'], d['parity_vs_cpu_checker'])
print(json.dumps(d['strong_scaling'], indent=1)[:1500])
"
