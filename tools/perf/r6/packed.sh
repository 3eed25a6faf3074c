set -o pipefail; mkdir -p gpurun_out/r6_packed; export MEMB_SYNTH_DEVICE=0
for threads in 64 32 16; do for chunks in -; do
MEMB_PACK_THREADS=$threads MEMB_PACK_CHUNKS=$chunks timeout -k 10 200 python tools/perf/r6/packed.py 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r6_packed/packed.txt || exit 1
done; done
