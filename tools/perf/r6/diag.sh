set -o pipefail; mkdir -p gpurun_out/r6_diag; export MEMB_SYNTH_DEVICE=0
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -s > gpurun_out/r6_diag/parity.txt 2>&1; code=$?
tail -30 gpurun_out/r6_diag/parity.txt | cut -c1-300
exit $code
