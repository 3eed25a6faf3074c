#!/bin/bash
# counters of the one-shot fill with 1 and 3 stores per thread
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/wcounters; mkdir -p $out
rocprofv3 -L > $out/avail.txt 2>&1
sets=("TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_STALL_sum TCC_EA0_WR_UNCACHED_32B_sum"
      "TCC_REQ_sum TCC_WRITE_sum TCC_WRITEBACK_sum TCC_NORMAL_WRITEBACK_sum"
      "TCC_EA0_WRREQ_IO_CREDIT_STALL_sum TCC_EA0_WRREQ_GMI_CREDIT_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum"
      "TCC_EA0_WRREQ_DRAM_sum TCC_EA0_WRREQ_LEVEL_sum TCC_TAG_STALL_sum TCC_BUSY_sum"
      "TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TCC_WRITE_REQ_LATENCY_sum"
      "WRITE_SIZE GRBM_GUI_ACTIVE")
i=0
for set in "${sets[@]}"; do
  for steps in 1 3; do
    rocprofv3 --pmc $set --output-format csv -d $out/s${i}_steps$steps -o pmc -- tools/perf/wsingle $steps 256 > $out/s${i}_steps$steps.log 2>&1
  done
  i=$((i+1))
done
python3 - <<PY
import csv, collections, glob
for f in sorted(glob.glob('$out/*/*counter_collection.csv')):
    agg=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        agg[r['Counter_Name']].append(float(r['Counter_Value']))
    for k,v in sorted(agg.items()): print('%-28s %-44s n=%d mean=%.6g' % (f.split('/')[2], k, len(v), sum(v)/len(v)))
PY
