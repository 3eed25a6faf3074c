#!/bin/bash
# round 3, batch 19: wavefront lifetime: persistent kernels launched as tiles / K wavefronts of ~K tiles each (K = 2 .. 16),
# general kernel (g..) and records kernel (r..), against persistent and one tile per wavefront
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3
export AB3_ROUNDS=4 AB3_REPS=20 MEMB_HIP_AUTOTUNE=0
V='onetile:persistent=0,g2:persistent=2;pipeline=0;tiles_per_wave=2,g4:persistent=2;pipeline=0;tiles_per_wave=4,g8:persistent=2;pipeline=0;tiles_per_wave=8,g16:persistent=2;pipeline=0;tiles_per_wave=16,r2:persistent=2;pipeline=1;tiles_per_wave=2,r4:persistent=2;pipeline=1;tiles_per_wave=4,r8:persistent=2;pipeline=1;tiles_per_wave=8,r16:persistent=2;pipeline=1;tiles_per_wave=16'
for bits in 2 4; do
AB3_BITS=$bits AB3=$V AB3_CASES=sorted,random,250k timeout -k 10 400 python3 tools/perf/ab3.py > gpurun_out/r3/b19_lifetime_bits$bits.log 2>&1; echo "bits $bits"; sed -n '/^---/,$p' gpurun_out/r3/b19_lifetime_bits$bits.log | grep -v "A/A"
done
