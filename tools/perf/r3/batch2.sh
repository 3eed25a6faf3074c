#!/bin/bash
# round 3, batch 2: persistent union kernel, even grid + early row loads (small batches), placement by alignment,
# nt loads on the 6- and 2-bit models, builder breakdown
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > gpurun_out/r3/b2_pytest.log 2>&1 || { tail -30 gpurun_out/r3/b2_pytest.log; exit 1; }
tail -3 gpurun_out/r3/b2_pytest.log
export AB3_ROUNDS=6 AB3_REPS=20
AB3='g0:grid_policy=0,oneshot:persistent=0,w4:waves_per_block=4,w4g0:waves_per_block=4;grid_policy=0,w2:waves_per_block=2,w1:waves_per_block=1,b3:blocks_per_cu=3' AB3_CASES=sorted,random,100k,10k,1k,union timeout -k 10 600 python3 tools/perf/ab3.py > gpurun_out/r3/b2_grid.log 2>&1 || { tail -30 gpurun_out/r3/b2_grid.log; exit 1; }
sed -n '/^---/,$p' gpurun_out/r3/b2_grid.log
AB3='a1g:!MEMB_HIP_STREAM_ALIGN=1073741824,a1g64m:!MEMB_HIP_STREAM_ALIGN=1073741824;!MEMB_HIP_STREAM_OFFSET=67108864,a2m256:!MEMB_HIP_STREAM_ALIGN=2097152;!MEMB_HIP_STREAM_OFFSET=256,a2m4k:!MEMB_HIP_STREAM_ALIGN=2097152;!MEMB_HIP_STREAM_OFFSET=4096' AB3_ROUNDS=4 AB3_CASES=sorted,random timeout -k 10 400 python3 tools/perf/ab3.py > gpurun_out/r3/b2_place.log 2>&1 || { tail -30 gpurun_out/r3/b2_place.log; exit 1; }
sed -n '/^---/,$p' gpurun_out/r3/b2_place.log
for bits in 6 2; do
AB3_BITS=$bits AB3='nt:nt_loads=1,g0:grid_policy=0' AB3_ROUNDS=4 AB3_CASES=sorted,random,100k timeout -k 10 400 python3 tools/perf/ab3.py > gpurun_out/r3/b2_aa_bits$bits.log 2>&1; echo "bits $bits"; sed -n '/^---/,$p' gpurun_out/r3/b2_aa_bits$bits.log
done
MEMB_BUILDER_VERBOSE=1 timeout -k 10 300 python3 tools/perf/buildtime.py > gpurun_out/r3/b2_buildtime.log 2>&1; cat gpurun_out/r3/b2_buildtime.log
nproc; free -g | head -2
