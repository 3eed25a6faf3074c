#!/bin/bash
# round 3, batch 25: tiles drawn from a counter (schedule 1) in the row-record pipeline against static dealing and the
# one-tile kernel (whose dispatcher balances the load by itself), 16 and 20 wavefronts per CU
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3
timeout -k 10 300 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "record_pipelines" > gpurun_out/r3/b25_pytest.log 2>&1 || { tail -30 gpurun_out/r3/b25_pytest.log; exit 1; }
tail -2 gpurun_out/r3/b25_pytest.log
export AB3_ROUNDS=4 AB3_REPS=20 MEMB_HIP_AUTOTUNE=0
for bits in 4 2; do
AB3_BITS=$bits AB3='g:persistent=2;pipeline=0,o:persistent=0,r:persistent=2;pipeline=1,d:persistent=2;pipeline=1;schedule=1,r4:persistent=2;pipeline=1;blocks_per_cu=4,d4:persistent=2;pipeline=1;schedule=1;blocks_per_cu=4' AB3_CASES=sorted,random,500k timeout -k 10 400 python3 tools/perf/ab3.py > gpurun_out/r3/b25_schedule_bits$bits.log 2>&1; echo "bits $bits"; sed -n '/^variant/p;/^---/,$p' gpurun_out/r3/b25_schedule_bits$bits.log | grep -v "A/A"
done
