import os, sys, numpy as np, torch
sys.path.insert(0, os.getcwd())
import memb_amd, oracle
from memb_amd import synthetic
G='tests/golden/'
for f in ['six_words_trained.bin','synthetic_4bit.bin']:
    r = memb_amd.Reader(G+f); o = oracle.OracleReader(G+f)
    n=len(r); rows=np.arange(n,dtype=np.uint32)
    exp=o.rows_embedding(rows)
    host=r.rows_embedding(rows)
    dev=r.rows_embedding_device(torch.from_numpy(rows.view(np.int32)).cuda()); torch.cuda.synchronize(); dev=dev.cpu().numpy()
    print(f, 'host==exp', np.array_equal(host,exp), 'dev==exp', np.array_equal(dev,exp), r.info())
    if not np.array_equal(dev,exp):
        bad=np.argwhere(dev!=exp); print('dev mismatches', len(bad), bad[:10])
    if not np.array_equal(host,exp):
        bad=np.argwhere(host!=exp); print('host mismatches', len(bad), bad[:10])
