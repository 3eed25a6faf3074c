import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
import memb_amd, oracle
from memb_amd import synthetic
for dim, bits, count in ((4096, 8, 300), (20000, 8, 60), (20000, 4, 60), (60000, 8, 20), (100000, 2, 12)):
    path='/tmp/big_%d_%d.bin'%(dim,bits)
    words=synthetic.build_file(path, count, dim, 'trained', bits, seed=dim)
    try:
        r=memb_amd.Reader(path); info=r.info()
        got=r.batch_embedding(sorted(words)+['zz'])
        want=oracle.OracleReader(path).batch_embedding(sorted(words)+['zz'])
        print(dim,bits,'ok' if np.array_equal(got.view(np.uint32),want.view(np.uint32)) else 'MISMATCH', {k:info[k] for k in ('lanes_per_word','segment_symbols','waves_per_block','max_stream_bytes','lds_bytes_per_block')}, flush=True)
    except Exception as e:
        print(dim,bits,'FAILED:',e, flush=True)
