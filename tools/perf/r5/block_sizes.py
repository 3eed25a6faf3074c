"""What block size does the rule choose for a full-size batch of each model kind, and what does LDS allow?"""
import os
import sys

sys.path.insert(0, os.getcwd())
import memb_amd
from memb_amd import synthetic

for bits, words, seed, dist in ((4, 2196017, 1234, 'normal'), (2, 2196017, 1234, 'normal'), (6, 1999995, 1234, 'normal'),
                                (4, 2196017, 99, 'normal'), (4, 2196017, 1234, 'student'), (8, 2196017, 1234, 'normal')):
    path, _ = synthetic.cached_model(words, 300, 'trained', bits, seed=seed, distribution=dist, device=0)
    reader = memb_amd.Reader(path)
    line = '%d-bit seed %d %-7s' % (bits, seed, dist)
    for waves in (0, 4, 8):
        reader.set_option('waves_per_block', waves)
        info = reader.info()
        line += ' | forced %d: waves %d, LDS %6d B -> %d blocks = %2d wavefronts per CU' % (
            waves, info['waves_per_block'], info['lds_bytes_per_block'], min(163840 // ((info['lds_bytes_per_block'] + 1023) // 1024 * 1024), 32 // info['waves_per_block']),
            min(163840 // ((info['lds_bytes_per_block'] + 1023) // 1024 * 1024), 32 // info['waves_per_block']) * info['waves_per_block'])
    print(line, '| max code bits', info['max_code_bits'], 'row bytes', info['row_bytes'], 'T', info['tiles_per_wavefront'], flush=True)
