#!/bin/bash
# round 3, batch 3: device writer tests; counters for the three weakest kernels (union, 100 k random rows, uniform);
# 8- vs 4-wave blocks on the 6- and 2-bit models
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3
timeout -k 10 1100 python3 -m pytest tests -m gpu -x -q -s > gpurun_out/r3/b3_pytest.log 2>&1 || { tail -40 gpurun_out/r3/b3_pytest.log; exit 1; }
tail -3 gpurun_out/r3/b3_pytest.log; grep "model build" gpurun_out/r3/b3_pytest.log
for bits in 6 2; do
AB3_BITS=$bits AB3='w8:waves_per_block=8,oneshot:persistent=0' AB3_ROUNDS=4 AB3_CASES=sorted,random,100k timeout -k 10 400 python3 tools/perf/ab3.py > gpurun_out/r3/b3_w8_bits$bits.log 2>&1; echo "bits $bits"; sed -n '/^---/,$p' gpurun_out/r3/b3_w8_bits$bits.log
done
timeout -k 10 500 bash tools/perf/prof.sh r3_union decode_trained_union --workload union-concat-500k > gpurun_out/r3/b3_prof_union.log 2>&1; tail -45 gpurun_out/r3/b3_prof_union.log
timeout -k 10 500 bash tools/perf/prof.sh r3_100k decode_trained_persistent --workload glove840b-300d-4bit-100k > gpurun_out/r3/b3_prof_100k.log 2>&1; tail -45 gpurun_out/r3/b3_prof_100k.log
timeout -k 10 500 bash tools/perf/prof.sh r3_uniform dequant_uniform --workload uniform-8bit-500k > gpurun_out/r3/b3_prof_uniform.log 2>&1; tail -45 gpurun_out/r3/b3_prof_uniform.log
