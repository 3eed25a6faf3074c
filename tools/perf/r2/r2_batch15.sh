#!/bin/bash
# round 2, batch 15: non-temporal stream / rowMeta loads, block barrier, combinations (steady state)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export AB2_ROUNDS=3 AB2_REPS=15 AB2_CASES=sorted,coldsorted,random,100k
AB2='base:0,ldnt:256,metant:512,ldmetant:768,sync:8,ldntsync:264,allsync:776,l4ldntsync:264:MEMB_HIP_LANES=4,l4ldnt:256:MEMB_HIP_LANES=4' timeout -k 10 700 python3 tools/perf/ab2.py > gpurun_out/r2_batch15_nt.log 2>&1 || { tail gpurun_out/r2_batch15_nt.log; exit 1; }
tail -10 gpurun_out/r2_batch15_nt.log
for bits in 6 2; do
AB2_BITS=$bits AB2='base:0,ldnt:256,ldntsync:264' timeout -k 10 400 python3 tools/perf/ab2.py > gpurun_out/r2_batch15_nt_bits$bits.log 2>&1; echo "bits $bits"; tail -3 gpurun_out/r2_batch15_nt_bits$bits.log
done
