"""Where the time of opening + staging a model goes (MEMB_HIP_VERBOSE=1 prints the library's breakdown)."""
import os, sys, time
sys.path.insert(0, os.getcwd())
os.environ['MEMB_HIP_VERBOSE'] = '1'
import memb_amd
from memb_amd import synthetic
path,_=synthetic.cached_model(2196017,300,'trained',4)
import torch; torch.zeros(1).cuda()   # runtime initialised, as in a running application
for i in range(3):
    t=time.time(); r=memb_amd.Reader(path); t1=time.time(); r.info(); t2=time.time()
    print('open (mmap + parse) %.3fs   stage to HBM %.3fs'%(t1-t, t2-t1), flush=True)
    del r
