"""Odd but legal models through the HIP path against the checker (a scratch probe; the cases that failed became tests)."""
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
import memb_amd, oracle
def check(name, words, vectors, storage, bits, extra=()):
    path='/tmp/odd_%s.bin'%name
    b=memb_amd.Builder(vectors.shape[1], storage, bits); b.add_words(words, vectors); b.save(path)
    try:
        r=memb_amd.Reader(path); c=oracle.OracleReader(path)
        batch=list(words)+['nope']+list(extra)
        g=r.batch_embedding(batch); w=c.batch_embedding(batch)
        same=np.array_equal(g.view(np.uint32), w.view(np.uint32))
        big=(batch*(700//len(batch)+1))[:700]
        same2=np.array_equal(r.batch_embedding(big).view(np.uint32), c.batch_embedding(big).view(np.uint32))
        print('%-28s %-8s %d bits: %s / %s  %s'%(name,storage,bits,'ok' if same else 'MISMATCH','ok' if same2 else 'MISMATCH', {k:v for k,v in r.info().items() if k in ('max_code_bits','root_bits','lanes_per_word','max_stream_bytes')}), flush=True)
    except Exception as e:
        print('%-28s %-8s %d bits: FAILED %s'%(name,storage,bits,e), flush=True)
rng=np.random.default_rng(0)
words=['w%04d'%i for i in range(300)]
for bits in (1,4,8):
    check('constant_%d'%bits, words, np.full((300,40),0.25,dtype=np.float32), 'trained', bits)
    check('zeros_%d'%bits, words, np.zeros((300,40),dtype=np.float32), 'trained', bits)
    check('two_values_%d'%bits, words, rng.choice(np.array([-1.0,2.0],dtype=np.float32),size=(300,40)), 'trained', bits)
    check('one_word_%d'%bits, words[:1], rng.standard_normal((1,40)).astype(np.float32), 'trained', bits)
    check('heavy_tail_%d'%bits, words, (rng.standard_t(1.5,size=(300,64))).astype(np.float32), 'trained', bits)
check('uniform_constant', words, np.full((300,40),3.0,dtype=np.float32), 'uniform', 8)
check('uniform_tiny', words, (rng.standard_normal((300,40))*1e-40).astype(np.float32), 'uniform', 8)
check('uniform_huge', words, (rng.standard_normal((300,40))*1e38).astype(np.float32), 'uniform', 8)
check('uniform_wide', words[:20], rng.standard_normal((20,50000)).astype(np.float32), 'uniform', 8)
check('full_wide', words[:20], rng.standard_normal((20,50000)).astype(np.float32), 'full', 8)
special=rng.standard_normal((300,40)).astype(np.float32); special[0,0]=np.inf; special[1,1]=-np.inf; special[2,2]=np.nan; special[3,3]=-0.0
check('full_special', words, special, 'full', 8)
check('dim1', words, rng.standard_normal((300,1)).astype(np.float32), 'trained', 4)
check('dim2_uniform', words, rng.standard_normal((300,2)).astype(np.float32), 'uniform', 4)
check('unicode_words', ['été','日本','a b','','\U0001F600','z'*300], rng.standard_normal((6,8)).astype(np.float32), 'trained', 4, extra=['é','日'])
