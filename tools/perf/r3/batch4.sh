#!/bin/bash
# round 3, batch 4: decode_records_persistent (stream registers / LDS-DMA) and its union form: parity, then the
# occupancy sweep (blocks of 4 wavefronts: blocks_per_cu x 4 wavefronts per CU) against the A/A floor; builder on the device
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3
export AB3_ROUNDS=4 AB3_REPS=20
AB3='p1:pipeline=1,p2:pipeline=2,oneshot:persistent=0,p1b4:pipeline=1;blocks_per_cu=4,p2b4:pipeline=2;blocks_per_cu=4,p2b5:pipeline=2;blocks_per_cu=5,p2b6:pipeline=2;blocks_per_cu=6' AB3_CASES=sorted,random,100k,10k,union timeout -k 10 600 python3 tools/perf/ab3.py > gpurun_out/r3/b4_occupancy.log 2>&1 || { tail -30 gpurun_out/r3/b4_occupancy.log; exit 1; }
sed -n '/^variant/p;/^---/,$p' gpurun_out/r3/b4_occupancy.log
for bits in 6 2; do
AB3_BITS=$bits AB3='p1:pipeline=1,p2:pipeline=2,oneshot:persistent=0,p2b4:pipeline=2;blocks_per_cu=4' AB3_CASES=sorted,random,100k timeout -k 10 400 python3 tools/perf/ab3.py > gpurun_out/r3/b4_occupancy_bits$bits.log 2>&1; echo "bits $bits"; sed -n '/^variant/p;/^---/,$p' gpurun_out/r3/b4_occupancy_bits$bits.log
done
# the same with output bursts of three pieces per lane (58 VGPRs with LDS-DMA: 8 wavefronts per SIMD)
MEMB_PACKAGE_ROOT=build/measure AB3='p1:pipeline=1,p2:pipeline=2,oneshot:persistent=0,p2b4:pipeline=2;blocks_per_cu=4,p2b6:pipeline=2;blocks_per_cu=6,p2b7:pipeline=2;blocks_per_cu=7' AB3_CASES=sorted,random,100k,union timeout -k 10 500 python3 tools/perf/ab3.py > gpurun_out/r3/b4_occupancy_burst3.log 2>&1; sed -n '/^variant/p;/^---/,$p' gpurun_out/r3/b4_occupancy_burst3.log
BT_DEVICE=0 MEMB_BUILDER_VERBOSE=1 timeout -k 10 300 python3 tools/perf/buildtime.py > gpurun_out/r3/b4_buildtime_device.log 2>&1; cat gpurun_out/r3/b4_buildtime_device.log
timeout -k 10 900 python3 -m pytest tests -m gpu -q > gpurun_out/r3/b4_pytest.log 2>&1; tail -5 gpurun_out/r3/b4_pytest.log
