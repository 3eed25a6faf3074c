import os, sys, numpy as np
sys.path.insert(0, os.getcwd())
import memb_amd, oracle
print('runtime', memb_amd.HIP_RUNTIME_PRELOADED)
G='tests/golden/'
f='six_words_trained.bin'
r = memb_amd.Reader(G+f); o = oracle.OracleReader(G+f)
print(r.info())
rows=np.arange(6,dtype=np.uint32)
print('batch6\n', r.rows_embedding(rows))
for i in range(6):
    print('single', i, r.rows_embedding(np.array([i],dtype=np.uint32)), o.rows_embedding(np.array([i],dtype=np.uint32)), 'streambytes', o.stream_bytes(i))
print('rev', r.rows_embedding(rows[::-1].copy()))
import torch
print(torch.cuda.is_available(), torch.zeros(3).cuda())
