set -o pipefail; mkdir -p gpurun_out/r6_gputests2; export MEMB_SYNTH_DEVICE=0
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/r6_gputests2/pytest.txt 2>&1 || { tail -40 gpurun_out/r6_gputests2/pytest.txt; exit 1; }
tail -3 gpurun_out/r6_gputests2/pytest.txt
SOAK_SECONDS=300 SOAK_SEED=606 bash tools/perf/soak.sh
