#!/bin/bash
# Round 4, batch 3: (a) T tiles per wavefront in the one-tile kernels (the block's table / codebook copy paid once for T
# tiles: batch 2 found the copy at 2.8 % of the dump and 8.6 % of the union), blocks of eight wavefronts;
# (b) before the general persistent kernel goes: models WITHOUT row records (8-bit, MEMB_HIP_ROW_RECORDS=0).
set -o pipefail
out=gpurun_out/r4_batch3
mkdir -p $out
export MEMB_SYNTH_DEVICE=0
# correctness of the loop first: the parity tests with three tiles per wavefront
MEMB_HIP_TILES_PER_WAVE=3 timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -k "ragged or full_dump or union or dimensions or wide or golden or randomized or strided or epilogue" > $out/parity_t3.txt 2>&1 || { tail -30 $out/parity_t3.txt; exit 1; }
tail -3 $out/parity_t3.txt
export AB3_ROUNDS=4
AB3='onetile:persistent=0,t2:persistent=0;tiles_per_wave=2,t3:persistent=0;tiles_per_wave=3,t4:persistent=0;tiles_per_wave=4,t8:persistent=0;tiles_per_wave=8,w8:persistent=0;waves_per_block=8,w8t2:persistent=0;waves_per_block=8;tiles_per_wave=2' \
    AB3_CASES=sorted,random,100k,500k,union timeout -k 10 300 python tools/perf/ab3.py > $out/tiles_4bit.txt 2>&1 || exit 1
sed -n '/--- median/,$p' $out/tiles_4bit.txt
AB3='onetile:persistent=0,t2:persistent=0;tiles_per_wave=2,t4:persistent=0;tiles_per_wave=4,w8:persistent=0;waves_per_block=8' \
    AB3_BITS=6 AB3_WORDS=1999995 AB3_CASES=sorted,random,100k timeout -k 10 300 python tools/perf/ab3.py > $out/tiles_6bit.txt 2>&1 || exit 1
sed -n '/--- median/,$p' $out/tiles_6bit.txt
AB3='onetile:persistent=0,general:persistent=2;pipeline=0,t2:persistent=0;tiles_per_wave=2,t4:persistent=0;tiles_per_wave=4' \
    AB3_BITS=8 AB3_CASES=sorted,random,100k,500k timeout -k 10 300 python tools/perf/ab3.py > $out/kernels_8bit.txt 2>&1 || exit 1
sed -n '/--- median/,$p' $out/kernels_8bit.txt
MEMB_HIP_ROW_RECORDS=0 AB3='onetile:persistent=0,general:persistent=2;pipeline=0,t4:persistent=0;tiles_per_wave=4' \
    AB3_CASES=sorted,random,100k,500k timeout -k 10 300 python tools/perf/ab3.py > $out/kernels_4bit_norecords.txt 2>&1 || exit 1
sed -n '/--- median/,$p' $out/kernels_4bit_norecords.txt
grep -l "output differs" $out/*.txt
true
