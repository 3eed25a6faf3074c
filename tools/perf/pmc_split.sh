#!/bin/bash
# usage: tools/perf/pmc_split.sh <tag> <workload>  -- LDS counters of the kernel with the decode / the output switched off
tag=$1; workload=$2
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for flags in 0 1 2; do
  out=gpurun_out/prof_${tag}_flags$flags; mkdir -p $out
  MEMB_HIP_DEBUG=$flags rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_INSTS_VALU GRBM_GUI_ACTIVE --output-format csv -d $out/pmc1 -o pmc -- python3 bench.py --workload $workload --no-configs --no-cpu-baseline --no-live-traffic --steps 3 --warmup 1 > /dev/null 2> $out/pmc1.err
  python3 - <<PY
import csv, collections, glob
agg=collections.defaultdict(list)
for f in glob.glob('$out/pmc1/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        if 'decode_trained' in r['Kernel_Name'] and ', 3' not in r['Kernel_Name']:
            agg[r['Counter_Name']].append(float(r['Counter_Value']))
m={k:sum(v)/len(v) for k,v in agg.items()}
print('flags $flags: LDS instrs %.3g  LDS cycles %.3g  conflict cycles %.3g (%.0f%%)  cycles/instr %.1f  kernel cycles/XCD %.3g  LDS busy %.0f%%' % (
    m['SQ_INSTS_LDS'], m['SQ_LDS_IDX_ACTIVE'], m['SQ_LDS_BANK_CONFLICT'], 100*m['SQ_LDS_BANK_CONFLICT']/m['SQ_LDS_IDX_ACTIVE'],
    m['SQ_LDS_IDX_ACTIVE']/m['SQ_INSTS_LDS'], m['GRBM_GUI_ACTIVE']/8, 100*m['SQ_LDS_IDX_ACTIVE']/256/(m['GRBM_GUI_ACTIVE']/8)))
PY
done
