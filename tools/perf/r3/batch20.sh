#!/bin/bash
# round 3, batch 20: one tile per wavefront: waves per block (each block copies table + codebook into LDS: 4-6 KB per 4 tiles)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3
export AB3_ROUNDS=4 AB3_REPS=20 MEMB_HIP_PERSISTENT=0
for bits in 2 4 6; do
AB3_BITS=$bits AB3='w8:waves_per_block=8,w2:waves_per_block=2' AB3_CASES=sorted,random,250k,10k timeout -k 10 300 python3 tools/perf/ab3.py > gpurun_out/r3/b20_onetile_blocks_bits$bits.log 2>&1; echo "bits $bits"; sed -n '/^---/,$p' gpurun_out/r3/b20_onetile_blocks_bits$bits.log | grep -v "A/A"
done
