#!/bin/bash
# round 3, batch 43: byte keys through the 4-byte PACKED tables in EVERY union kernel (was: decode_union_split only; the
# persistent and one-tile unions decoded byte keys through the 8-byte table). GPU union tests, then two 6-bit models:
# s = split, n = persistent union (union_split=0), no = one-tile union -- compare n / no with batch 38 (0.323 / 0.517 ms at 500 k)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3
timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_full_size.py -x -q -m gpu -k "union or Union" > gpurun_out/r3/b43_pytest.log 2>&1 || { tail -30 gpurun_out/r3/b43_pytest.log; exit 1; }
tail -2 gpurun_out/r3/b43_pytest.log
export AB3_ROUNDS=4 AB3_REPS=30 AB3_BURST=1 MEMB_HIP_AUTOTUNE=0 AB3_BITS=6 AB3_UNION_BITS=6 MEMB_SYNTH_DEVICE=0
for words in 100000 500000; do
echo "union of $words words, two 6-bit models"
AB3_UNION_WORDS=$words AB3='s:union_split=1,n:union_split=0,no:union_split=0;persistent=0' AB3_CASES=union timeout -k 10 400 python3 tools/perf/ab3.py > gpurun_out/r3/b43_tmp.log 2>&1; sed -n '/^case/,$p' gpurun_out/r3/b43_tmp.log | grep -v "A/A"; { echo "# union of $words words, two 6-bit models"; cat gpurun_out/r3/b43_tmp.log; } >> gpurun_out/r3/b43_union_packed.log
done
