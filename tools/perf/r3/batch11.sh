#!/bin/bash
# round 3, batch 11: union kernel after the cheaper output indexing: occupancy (blocks of 4 wavefronts per CU), one-tile form
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3
export AB3_ROUNDS=4 AB3_REPS=20
AB3='b4:blocks_per_cu=4,b5:blocks_per_cu=5,w8:waves_per_block=8,onetile:persistent=0,dma:pipeline=2' AB3_CASES=union,500k timeout -k 10 400 python3 tools/perf/ab3.py > gpurun_out/r3/b11_union.log 2>&1 || { tail -30 gpurun_out/r3/b11_union.log; exit 1; }
sed -n '/^---/,$p' gpurun_out/r3/b11_union.log
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_full_size.py -m gpu -q -k "union" > gpurun_out/r3/b11_pytest.log 2>&1; tail -3 gpurun_out/r3/b11_pytest.log
