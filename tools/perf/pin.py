import time, torch, numpy as np
n=2196017
g=torch.randn((n,300),device='cuda'); torch.cuda.synchronize()
for size in (100000, n):
    t=time.time(); p=torch.empty((size,300),pin_memory=True); t1=time.time()-t
    t=time.time(); p.copy_(g[:size]); torch.cuda.synchronize(); t2=time.time()-t
    t=time.time(); p.copy_(g[:size]); torch.cuda.synchronize(); t3=time.time()-t
    a=np.empty((size,300),dtype=np.float32)
    t=time.time(); torch.from_numpy(a).copy_(g[:size]); torch.cuda.synchronize(); t4=time.time()-t
    t=time.time(); torch.from_numpy(a).copy_(g[:size]); torch.cuda.synchronize(); t5=time.time()-t
    t=time.time(); torch.cuda.cudart().cudaHostRegister(a.ctypes.data, a.nbytes, 0); t6=time.time()-t
    t=time.time(); torch.from_numpy(a).copy_(g[:size]); torch.cuda.synchronize(); t7=time.time()-t
    torch.cuda.cudart().cudaHostUnregister(a.ctypes.data)
    gb=size*1200/1e9
    print('rows %8d (%.2f GB): pinned alloc %.3fs | D2H pinned %.4fs (%.1f GB/s) again %.4fs (%.1f GB/s) | pageable %.4fs / %.4fs (%.1f GB/s) | hostRegister %.3fs then D2H %.4fs (%.1f GB/s)'%(size,gb,t1,t2,gb/t2,t3,gb/t3,t4,t5,gb/t5,t6,t7,gb/t7))
