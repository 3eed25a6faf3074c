set -o pipefail; mkdir -p gpurun_out/r6_expand; export MEMB_SYNTH_DEVICE=0; timeout -k 10 300 python tools/perf/r6/expand.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r6_expand/expand.txt
