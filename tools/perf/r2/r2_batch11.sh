#!/bin/bash
# round 2, batch 11: block-synchronous output (bit 3), with 8- and 4-wave blocks; torch fill for box calibration
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python3 - <<'PY' > gpurun_out/r2_batch11_fill.log 2>&1
import torch
n=2196017
out=torch.empty((n,300),dtype=torch.float32,device='cuda')
for _ in range(3): out.fill_(1.0)
torch.cuda.synchronize()
ev=[(torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)) for _ in range(20)]
for a,b in ev:
    a.record(); out.fill_(1.0); b.record()
torch.cuda.synchronize()
ms=sorted(a.elapsed_time(b) for a,b in ev); print('torch fill_ of the output: median %.4f ms = %.2f TB/s' % (ms[10], n*1200/ms[10]/1e9))
PY
cat gpurun_out/r2_batch11_fill.log
export AB2_ROUNDS=3 AB2_REPS=12 AB2_CASES=sorted,random,100k
AB2='base:0,sync:8,w4:0:MEMB_HIP_WAVES=4,w4sync:8:MEMB_HIP_WAVES=4,nodecode:1,nodecsync:9,noloadsync:13' timeout -k 10 500 python3 tools/perf/ab2.py > gpurun_out/r2_batch11_sync.log 2>&1 || { tail gpurun_out/r2_batch11_sync.log; exit 1; }
grep DIFFERS gpurun_out/r2_batch11_sync.log; tail -8 gpurun_out/r2_batch11_sync.log
