#!/bin/bash
# usage: tools/perf/pmc.sh <tag>   (run from repo root on the GPU box)
tag=$1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/prof_$tag
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$tag/trace -o trace -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-configs --no-live-traffic > gpurun_out/prof_$tag/bench.json 2> gpurun_out/prof_$tag/trace.err
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAIT_INST_LDS --output-format csv -d gpurun_out/prof_$tag/pmc1 -o pmc -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-configs --no-live-traffic > /dev/null 2> gpurun_out/prof_$tag/pmc1.err
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_WR --output-format csv -d gpurun_out/prof_$tag/pmc2 -o pmc -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-configs --no-live-traffic > /dev/null 2> gpurun_out/prof_$tag/pmc2.err
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL --output-format csv -d gpurun_out/prof_$tag/pmc3 -o pmc -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-configs --no-live-traffic > /dev/null 2> gpurun_out/prof_$tag/pmc3.err
rocprofv3 --pmc TCP_TCC_WRITE_REQ_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_LATENCY_sum TCP_TCC_READ_REQ_LATENCY_sum --output-format csv -d gpurun_out/prof_$tag/pmc4 -o pmc -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-configs --no-live-traffic > /dev/null 2> gpurun_out/prof_$tag/pmc4.err
rocprofv3 --pmc TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_GATE_EN1_sum TCP_TOTAL_CACHE_ACCESSES_sum --output-format csv -d gpurun_out/prof_$tag/pmc5 -o pmc -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-configs --no-live-traffic > /dev/null 2> gpurun_out/prof_$tag/pmc5.err
python3 - <<PY
import csv, collections, glob, json
out={}
for f in sorted(glob.glob('gpurun_out/prof_$tag/pmc*/*counter_collection.csv')):
    agg=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if 'decode_trained' in r['Kernel_Name'] and ', 3' not in r['Kernel_Name']:
            agg[r['Counter_Name']].append(float(r['Counter_Value']))
    for k,v in agg.items(): out[k]=sum(v)/len(v)
json.dump(out, open('gpurun_out/prof_$tag/pmc_summary.json','w'), indent=1)
for k in sorted(out): print('%-26s %.4g'%(k,out[k]))
PY
cut -c1-160 gpurun_out/prof_$tag/trace/trace_kernel_stats.csv | head -4
