#!/bin/bash
# round 3, batch 12: 14 / 18 wavefronts per CU (2-wave blocks); the 2-bit model on row records against the compact layout;
# launcher facts on the box; smoke; GPU suite; final bench line on a fresh model cache
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3
export AB3_ROUNDS=4 AB3_REPS=20
AB3='w2b7:waves_per_block=2;blocks_per_cu=7;persistent=2;pipeline=0,w2b8:waves_per_block=2;blocks_per_cu=8;persistent=2;pipeline=0,w2b9:waves_per_block=2;blocks_per_cu=9;persistent=2;pipeline=0' AB3_CASES=sorted,random timeout -k 10 300 python3 tools/perf/ab3.py > gpurun_out/r3/b12_fine.log 2>&1; sed -n '/^---/,$p' gpurun_out/r3/b12_fine.log
AB3_BITS=2 AB3='compact:!MEMB_HIP_ROW_RECORDS=0,compactp:!MEMB_HIP_ROW_RECORDS=0;persistent=2,compacto:!MEMB_HIP_ROW_RECORDS=0;persistent=0,recordsp:persistent=2;pipeline=0,recordso:persistent=0' AB3_CASES=sorted,random,100k timeout -k 10 300 python3 tools/perf/ab3.py > gpurun_out/r3/b12_layout_2bit.log 2>&1; sed -n '/^variant/p;/^---/,$p' gpurun_out/r3/b12_layout_2bit.log
python3 -c "
import bench
print('kfd gpus', bench.kfd_gpu_count(), 'hip mapped', bench.hip_runtime_mapped())"
python3 bench.py --gpus 2 --dry-launch --small | tail -1 | cut -c1-300
timeout -k 10 300 python3 __graft_entry__.py --smoke 2>&1 | tail -2
timeout -k 10 900 python3 bench.py --cache-dir /tmp/memb_fresh_cache2 > gpurun_out/r3/b12_bench.json 2> gpurun_out/r3/b12_bench.err; tail -c 300 gpurun_out/r3/b12_bench.err
python3 - <<'PY'
import json
line=[l for l in open('gpurun_out/r3/b12_bench.json') if l.startswith('{')][-1]
d=json.loads(line)
print('headline', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['kernel'], 'model_build_s', d['model_build_s'], d['parity_vs_cpu_checker'])
for c in d['configs']: print('%-70s %-45s %.4f ms frac %.3f %s' % (c['workload'][:70], c['kernel'][:45], c['kernel_ms'], c['frac'], c['parity']))
print(d['cpu_baseline']['value'], d['host_api']['batch_seconds'], d['host_api']['sample_seconds'], d['roofline']['box_fill']['ms'])
PY
timeout -k 10 900 python3 -m pytest tests -m gpu -q > gpurun_out/r3/b12_pytest.log 2>&1; tail -4 gpurun_out/r3/b12_pytest.log
