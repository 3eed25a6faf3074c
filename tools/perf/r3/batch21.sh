#!/bin/bash
# round 3, batch 21: GPU suite after the mixed-format union
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3
timeout -k 10 900 python3 -m pytest tests -m gpu -q > gpurun_out/r3/b21_pytest.log 2>&1; tail -15 gpurun_out/r3/b21_pytest.log
