"""Where the time of writing a synthetic 2.2 M-word model goes (host side; run on the GPU box)."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
from memb_amd import synthetic
from memb_amd.builder import Builder
n = int(os.environ.get('BT_WORDS', '2196017')); bits = int(os.environ.get('BT_BITS', '4'))
t = time.time(); words = synthetic.make_words(n); print('make_words %.2fs' % (time.time() - t), flush=True)
device = os.environ.get('BT_DEVICE')
builder = Builder(300, 'trained', bits, device=None if device in (None, '') else int(device))
print('writer:', 'host' if device in (None, '') else 'device ' + device, flush=True)
rng = np.random.default_rng(1234)
t_rng = t_add = 0.0
for start in range(0, n, 200000):
    stop = min(n, start + 200000)
    t = time.time(); block = rng.standard_normal((stop - start, 300), dtype=np.float32) * np.float32(0.4); t_rng += time.time() - t
    t = time.time(); builder.add_words(words[start:stop], block); t_add += time.time() - t
print('rng %.2fs add_words %.2fs' % (t_rng, t_add), flush=True)
os.environ['MEMB_HIP_VERBOSE'] = '1'
t = time.time(); builder.save('/tmp/bt.bin'); print('save %.2fs' % (time.time() - t), flush=True)
