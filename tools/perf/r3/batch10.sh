#!/bin/bash
# round 3, batch 10: rocprofv3 kernel trace + counter passes of the final kernels: headline, union, 100 k rows, uniform;
# the autotune decision on three models
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3
export AB3_ROUNDS=3 AB3_REPS=20
for bits in 2 6 4; do
MEMB_HIP_VERBOSE=1 AB3_BITS=$bits AB3='general:persistent=2;pipeline=0,onetile:persistent=0' AB3_CASES=sorted,random timeout -k 10 300 python3 tools/perf/ab3.py > gpurun_out/r3/b10_autotune_bits$bits.log 2>&1
echo "bits $bits"; grep "large batches" gpurun_out/r3/b10_autotune_bits$bits.log | head -2; sed -n '/^---/,$p' gpurun_out/r3/b10_autotune_bits$bits.log | grep -v "A/A"
done
timeout -k 10 500 bash tools/perf/prof.sh r3_headline decode_trained_persistent > gpurun_out/r3/b10_prof_headline.log 2>&1; grep -E "^(trace|bench|traffic_over|hbm_|lds_conflict|read_latency|write_latency)" gpurun_out/r3/b10_prof_headline.log
timeout -k 10 500 bash tools/perf/prof.sh r3_union_after decode_records_union --workload union-concat-500k > gpurun_out/r3/b10_prof_union.log 2>&1; grep -E "^(trace|bench|traffic_over|hbm_|lds_conflict|read_latency|write_latency|SQ_INSTS_VALU)" gpurun_out/r3/b10_prof_union.log
timeout -k 10 500 bash tools/perf/prof.sh r3_100k_after decode_records_persistent --workload glove840b-300d-4bit-100k > gpurun_out/r3/b10_prof_100k.log 2>&1; grep -E "^(trace|bench|traffic_over|hbm_|lds_conflict|read_latency|write_latency)" gpurun_out/r3/b10_prof_100k.log
timeout -k 10 500 bash tools/perf/prof.sh r3_uniform_after dequant_uniform_persistent --workload uniform-8bit-500k > gpurun_out/r3/b10_prof_uniform.log 2>&1; grep -E "^(trace|bench|traffic_over|hbm_|lds_conflict|read_latency|write_latency)" gpurun_out/r3/b10_prof_uniform.log
