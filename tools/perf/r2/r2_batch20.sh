#!/bin/bash
# round 2, batch 20: non-temporal value loads in dequant_uniform / gather_full (two library builds, alternated)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
cp memb_amd/libmemb_hip.so /tmp/main.so
for round in 1 2; do
  cp /tmp/main.so memb_amd/libmemb_hip.so; echo "nt loads (round $round)"; python3 tools/perf/rowwise_ab.py 2>&1 | grep median
  cp tools/perf/variants/libmemb_hip_plainrowwise.so memb_amd/libmemb_hip.so; echo "plain loads (round $round)"; python3 tools/perf/rowwise_ab.py 2>&1 | grep median
done
cp /tmp/main.so memb_amd/libmemb_hip.so
