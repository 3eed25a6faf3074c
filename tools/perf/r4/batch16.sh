#!/bin/bash
# Round 4, batch 16: dequant_uniform_tile (one tile per wavefront, no pipeline) against the LDS-DMA pipeline and the block
# kernel, one Reader; uniform parity tests first.
set -o pipefail
out=gpurun_out/r4_batch16
mkdir -p $out
export MEMB_SYNTH_DEVICE=0
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -q -x -k "uniform or round_trip or small_batches or dimensions or strided or epilogue or merged or degenerate or empty" > $out/tests.log 2>&1 || { tail -30 $out/tests.log; exit 1; }
tail -2 $out/tests.log
timeout -k 10 300 python tools/perf/r4/uniform_ab.py 2>/dev/null | tee $out/uniform_kernels.txt
