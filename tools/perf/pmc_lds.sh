#!/bin/bash
# usage: tools/perf/pmc_lds.sh <tag> <bench workload>   -- LDS / wave-state counters of the decode kernel of one workload
tag=$1; workload=$2
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/prof_$tag; mkdir -p $out
args="bench.py --workload $workload --no-configs --no-cpu-baseline --no-live-traffic --steps 3 --warmup 1"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o trace -- python3 bench.py --workload $workload --no-configs --no-cpu-baseline --no-live-traffic --steps 10 --warmup 3 > $out/bench.json 2> $out/trace.err
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES --output-format csv -d $out/pmc1 -o pmc -- python3 $args > /dev/null 2> $out/pmc1.err
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_WAIT_INST_LDS SQ_WAVES SQ_BUSY_CYCLES SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL --output-format csv -d $out/pmc2 -o pmc -- python3 $args > /dev/null 2> $out/pmc2.err
python3 - <<PY
import csv, collections, glob, json
out={}
for f in sorted(glob.glob('$out/pmc*/*counter_collection.csv')):
    agg=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if 'decode_trained' in r['Kernel_Name'] and ', 3' not in r['Kernel_Name']:
            agg[r['Counter_Name']].append(float(r['Counter_Value']))
            out['kernel']=r['Kernel_Name']; out['VGPR_Count']=r.get('VGPR_Count'); out['LDS_Block_Size']=r.get('LDS_Block_Size')
    for k,v in agg.items(): out[k]=sum(v)/len(v)
if out.get('SQ_LDS_IDX_ACTIVE'): out['lds_conflict_share']=out['SQ_LDS_BANK_CONFLICT']/out['SQ_LDS_IDX_ACTIVE']
if out.get('GRBM_GUI_ACTIVE') and out.get('SQ_LDS_IDX_ACTIVE'): out['lds_busy_share_of_kernel_cycles']=out['SQ_LDS_IDX_ACTIVE']/256/(out['GRBM_GUI_ACTIVE']/8)
json.dump(out, open('$out/pmc_summary.json','w'), indent=1)
print(json.dumps(out, indent=1))
PY
grep decode_trained $out/trace/trace_kernel_stats.csv | cut -c1-200
