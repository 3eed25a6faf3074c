#!/bin/bash
# round 3, batch 5: occupancy DOWNWARD (batch 4: every step above 16 wavefronts per CU cost the dumps 1-15 %), for the
# persistent kernel (b1..b4 = 4..16 wavefronts per CU), the one-tile kernel (o = 32, o6 / o4 / o3 / o2 = 24 / 16 / 12 / 8)
# and the records kernel, over batch sizes 1 k .. 2.2 M
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3
export AB3_ROUNDS=4 AB3_REPS=20
V='b3:blocks_per_cu=3,b2:blocks_per_cu=2,b1:blocks_per_cu=1,o:persistent=0,o6:persistent=0;blocks_per_cu=6,o4:persistent=0;blocks_per_cu=4,o3:persistent=0;blocks_per_cu=3,o2:persistent=0;blocks_per_cu=2,p1:pipeline=1,p1b4:pipeline=1;blocks_per_cu=4,p1b3:pipeline=1;blocks_per_cu=3'
for bits in 4 2 6; do
AB3_BITS=$bits AB3=$V AB3_CASES=sorted,random,500k,250k,100k,50k,10k,1k timeout -k 10 500 python3 tools/perf/ab3.py > gpurun_out/r3/b5_sweep_bits$bits.log 2>&1 || { tail -30 gpurun_out/r3/b5_sweep_bits$bits.log; exit 1; }
echo "bits $bits"; sed -n '/^---/,$p' gpurun_out/r3/b5_sweep_bits$bits.log | grep -v "A/A"
done
