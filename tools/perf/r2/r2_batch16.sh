#!/bin/bash
# round 2, batch 16: row records (fixed-size row regions, record in front of the stream) vs the compact layout + rowMeta
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py -x -q > gpurun_out/r2_batch16_parity.log 2>&1 || { tail -30 gpurun_out/r2_batch16_parity.log; exit 1; }
tail -3 gpurun_out/r2_batch16_parity.log
export AB2_ROUNDS=3 AB2_REPS=15 AB2_CASES=sorted,coldsorted,random,100k
for bits in 4 2 8; do
  AB2_BITS=$bits AB2='records:0,rowmeta:0:MEMB_HIP_ROW_RECORDS=0,arrays:0:MEMB_HIP_ROW_META=0' timeout -k 10 500 python3 tools/perf/ab2.py > gpurun_out/r2_batch16_bits$bits.log 2>&1 || { tail gpurun_out/r2_batch16_bits$bits.log; exit 1; }
  echo "bits $bits"; grep DIFFERS gpurun_out/r2_batch16_bits$bits.log; tail -4 gpurun_out/r2_batch16_bits$bits.log
done
