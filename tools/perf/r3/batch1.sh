#!/bin/bash
# round 3, batch 1: GPU suite after the refactor; A/A control + real nt loads on one allocation; placement; builder breakdown
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3
timeout -k 10 600 python3 -m pytest tests -m gpu -x -q > gpurun_out/r3/b1_pytest.log 2>&1 || { tail -30 gpurun_out/r3/b1_pytest.log; exit 1; }
tail -3 gpurun_out/r3/b1_pytest.log
export AB3_ROUNDS=6 AB3_REPS=20
AB3='nt:nt_loads=1' AB3_PLACEMENT=4 AB3_OUT_BUFFERS=2 AB3_CASES=sorted,random,100k timeout -k 10 500 python3 tools/perf/ab3.py > gpurun_out/r3/b1_aa.log 2>&1 || { tail -30 gpurun_out/r3/b1_aa.log; exit 1; }
sed -n '/^---/,$p' gpurun_out/r3/b1_aa.log
AB3='a1g:!MEMB_HIP_STREAM_ALIGN=1073741824,a1g64m:!MEMB_HIP_STREAM_ALIGN=1073741824;MEMB_HIP_STREAM_OFFSET=67108864,a2m256:!MEMB_HIP_STREAM_ALIGN=2097152;MEMB_HIP_STREAM_OFFSET=256,a2m4k:!MEMB_HIP_STREAM_ALIGN=2097152;MEMB_HIP_STREAM_OFFSET=4096' AB3_CASES=sorted,random timeout -k 10 400 python3 tools/perf/ab3.py > gpurun_out/r3/b1_place.log 2>&1 || { tail -30 gpurun_out/r3/b1_place.log; exit 1; }
sed -n '/^---/,$p' gpurun_out/r3/b1_place.log
for bits in 6 2; do
AB3_BITS=$bits AB3='nt:nt_loads=1' AB3_ROUNDS=4 AB3_CASES=sorted,random,100k timeout -k 10 400 python3 tools/perf/ab3.py > gpurun_out/r3/b1_aa_bits$bits.log 2>&1; echo "bits $bits"; sed -n '/^---/,$p' gpurun_out/r3/b1_aa_bits$bits.log
done
MEMB_BUILDER_VERBOSE=1 timeout -k 10 300 python3 tools/perf/buildtime.py > gpurun_out/r3/b1_buildtime.log 2>&1; cat gpurun_out/r3/b1_buildtime.log
nproc; free -g | head -2
