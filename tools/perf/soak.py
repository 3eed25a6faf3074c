"""Soak: Readers created and dropped in a loop, host and device lookups of random sizes, unions, several threads --
every result checked against the CPU checker. Exits non-zero on the first mismatch; prints progress every 50 rounds."""
import os, sys, time, threading
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import memb_amd, oracle
from memb_amd import synthetic
seconds=float(os.environ.get('SOAK_SECONDS','240'))
rng=np.random.default_rng(int(os.environ.get('SOAK_SEED','1')))
models=[]
for i,(count,dim,storage,bits) in enumerate([(30000,300,'trained',4),(20000,300,'trained',6),(5000,64,'trained',2),(8000,300,'uniform',8),(3000,100,'full',8),(12000,300,'trained',8)]):
    path='/tmp/soak_%d.bin'%i
    words=synthetic.build_file(path,count,dim,storage,bits,seed=100+i)
    models.append((path,words,oracle.OracleReader(path)))
start=time.time(); rounds=0; lookups=0
def check(got, want, what):
    if not np.array_equal(np.asarray(got).view(np.uint32), want.view(np.uint32)):
        print('MISMATCH', what, flush=True); os._exit(1)
while time.time()-start < seconds:
    path,words,checker=models[int(rng.integers(0,len(models)))]
    reader=memb_amd.Reader(path)
    if rng.random()<0.5 and 'trained' in reader._impl.storage_name():
        # round 3: the kernels are chosen by batch size; force other choices now and then (results never depend on them)
        reader.set_option('tiles_per_wave', int(rng.choice([0,0,1,2,3,7]))); reader.set_option('persistent', int(rng.integers(0,3)))
        reader.set_option('waves_per_block', int(rng.choice([0,0,1,2,4,7,8])))
        reader.set_option('fine_lanes', int(rng.integers(0,3)))   # round 5: the finer segment index by rule / never / always
    for _ in range(int(rng.integers(1,6))):
        n=int(rng.choice([1,2,17,64,500,513,3000,20000,28672,28673,57345,60000,65537,140000,300000,600000]))   # (600 000: past 16 R tiles, where the block size follows the order of the batch before)
        batch=[words[i] for i in rng.integers(0,len(words),size=n)]
        if rng.random()<0.5: batch[::7]=['?']*len(batch[::7])
        want=checker.batch_embedding(batch)
        kind=int(rng.integers(0,8))
        if kind==4:
            # round 5: word -> row on the device against the checker's binary search
            got=reader.resolve_rows_device(batch).cpu().numpy().view(np.uint32)
            if not np.array_equal(got, checker.resolve_rows(batch)): print('MISMATCH resolve', n, flush=True); os._exit(1)
        elif kind==5:
            # several batches in one launch: the batch cut into ragged pieces
            rows=reader.resolve_rows_device(batch)
            cuts=sorted(set([0,n]+[int(c) for c in rng.integers(0,n+1,size=int(rng.integers(0,6)))]))
            pieces=[(rows[a:b].contiguous(), torch.empty((b-a,reader.dim),dtype=torch.float32,device='cuda')) for a,b in zip(cuts[:-1],cuts[1:])]
            outs=reader.rows_embedding_device_many(pieces)
            check(torch.cat(outs).cpu().numpy() if outs else np.zeros((0,reader.dim),dtype=np.float32), want, 'many')
        elif kind==7:
            # round 6: the same words packed already (UTF-8 bytes + offsets), from host memory or from device tensors
            encoded=[w.encode('utf-8') for w in batch]; blob=b''.join(encoded)
            starts=np.zeros(n+1,dtype=np.uint32); np.cumsum([len(e) for e in encoded],out=starts[1:])
            if rng.random()<0.5 or not blob:
                got=reader.resolve_packed_device(blob, starts)
            else:
                got=reader.resolve_packed_device(torch.from_numpy(np.frombuffer(blob,dtype=np.uint8).copy()).cuda(), torch.from_numpy(starts.view(np.int32).copy()).cuda())
            if not np.array_equal(got.cpu().numpy().view(np.uint32), checker.resolve_rows(batch)): print('MISMATCH packed', n, flush=True); os._exit(1)
            if rng.random()<0.5: rows_sorted,_=torch.sort(got.to(torch.int64)); check(reader.rows_embedding_device(rows_sorted.to(torch.int32)).cpu().numpy(), checker.rows_embedding(rows_sorted.cpu().numpy().astype(np.uint32)), 'sorted rows')
        elif kind==6:
            rows=reader.resolve_rows_device(batch)
            check(reader.rows_embedding_device(rows, order='random').cpu().numpy(), want, 'order hint')
        elif kind==0: check(reader.batch_embedding(batch), want, 'host')
        elif kind==1: check(reader.batch_embedding_device(batch).cpu().numpy(), want, 'device')
        elif kind==2:
            wide=np.zeros((n,reader.dim+9),dtype=np.float32); reader.batch_embedding_into(batch,wide,5); check(np.ascontiguousarray(wide[:,5:5+reader.dim]), want, 'strided')
        else:
            results=[None,None]
            def work(i):
                results[i]=reader.batch_embedding(batch)
            ts=[threading.Thread(target=work,args=(i,)) for i in range(2)]
            [t.start() for t in ts]; [t.join() for t in ts]
            check(results[0], want, 'thread0'); check(results[1], want, 'thread1')
        lookups+=1
    if rng.random()<0.3:
        a=memb_amd.Reader(models[0][0]); b=memb_amd.Reader(models[1][0])
        pool=models[0][1][:2000]+models[1][1][:2000]
        if rng.random()<0.5: a.set_option('tiles_per_wave', int(rng.choice([0,1,2,3])))
        a.set_option('union_split', int(rng.integers(0,2))); a.set_option('persistent', int(rng.integers(0,3)))
        batch=[pool[i] for i in rng.integers(0,len(pool),size=int(rng.choice([5,700,9000,90000])))]
        for mode in ('concatenate','average'):
            u=memb_amd.ReadersUnion([a,b],mode)
            rows=[models[0][2].batch_embedding(batch), models[1][2].batch_embedding(batch)]
            want=np.concatenate(rows,axis=1) if mode=='concatenate' else np.mean(rows,axis=0)
            check(u.batch_embedding_device(batch).cpu().numpy(), want, 'union '+mode)
            check(u.batch_embedding(batch), want, 'union host '+mode)
        del a,b
    del reader
    if rng.random()<0.05:
        # the device writer: same bytes as the host writer
        count=int(rng.choice([300,9999,10000,10001,25000])); dim=int(rng.choice([3,7,64,300]))
        w=synthetic.make_words(count); v=synthetic.make_vectors(count,dim,seed=int(rng.integers(0,1000)))
        bits_used=int(rng.choice([2,4,6,8]))
        for device,name in ((None,'/tmp/soak_host.bin'),(0,'/tmp/soak_device.bin')):
            b=memb_amd.Builder(dim,'trained',bits_used,device=device)
            edges=np.linspace(0,count,int(rng.integers(1,5))+1).astype(int)
            for lo,hi in zip(edges[:-1],edges[1:]): b.add_words(w[lo:hi],v[lo:hi])
            b.save(name)
        if open('/tmp/soak_host.bin','rb').read()!=open('/tmp/soak_device.bin','rb').read():
            print('MISMATCH device writer',count,dim,bits_used,flush=True); os._exit(1)
    rounds+=1
    if rounds%50==0: print('round %d, %d lookups, %.0f s'%(rounds,lookups,time.time()-start), flush=True)
print('soak ok: %d rounds, %d lookups in %.0f s'%(rounds,lookups,time.time()-start))
