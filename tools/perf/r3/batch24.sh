#!/bin/bash
# round 3, batch 24: lanes per word (words per tile = 64 / lanes) with both large-batch kernels, on the A/A-controlled harness
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3
export AB3_ROUNDS=4 AB3_REPS=20 MEMB_HIP_AUTOTUNE=0
for bits in 4 2; do
AB3_BITS=$bits AB3='p:persistent=2,o:persistent=0,l4p:!MEMB_HIP_LANES=4;persistent=2,l4o:!MEMB_HIP_LANES=4;persistent=0,l16p:!MEMB_HIP_LANES=16;persistent=2,l16o:!MEMB_HIP_LANES=16;persistent=0' AB3_CASES=sorted,random,100k timeout -k 10 400 python3 tools/perf/ab3.py > gpurun_out/r3/b24_lanes_bits$bits.log 2>&1; echo "bits $bits"; sed -n '/^variant/p;/^---/,$p' gpurun_out/r3/b24_lanes_bits$bits.log | grep -v "A/A"
done
