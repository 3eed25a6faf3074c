"""tools/converter/memb_converter (reference tools/converter/memb_converter:8-87): median dimension, discarded
lines, unparsable weights, duplicates, --max-words, the .vec header line; read back through the host path."""
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import REPO, bits_equal

CONVERTER = os.path.join(REPO, 'tools', 'converter', 'memb_converter')


def run(arguments):
    result = subprocess.run([sys.executable, CONVERTER] + arguments, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                            timeout=300)
    return result.returncode, result.stdout, result.stderr


def test_converter_follows_the_reference_script(native, tmp_path):
    rng = np.random.default_rng(4)
    vectors = {'w{}'.format(i): rng.standard_normal(5).astype(np.float32) for i in range(40)}
    lines = ['40 5']                                                  # .vec header: two fields, skipped
    for word, vector in vectors.items():
        lines.append(word + ' ' + ' '.join(repr(float(x)) for x in vector))
    lines.insert(7, 'short 1.0 2.0 3.0')                              # another length: discarded
    lines.insert(9, 'longer ' + ' '.join(['0.5'] * 6))
    lines.insert(11, 'broken 1.0 2.0 x 4.0 5.0')                      # not a number: reported, skipped
    lines.insert(13, 'w3 ' + ' '.join(['9.0'] * 5))                   # seen before: reported, first one stays
    lines.insert(15, 'lonely')                                        # no weights
    source = tmp_path / 'vectors.txt'
    source.write_text('\n'.join(lines) + '\n')
    target = str(tmp_path / 'full.bin')
    code, out, err = run(['--from', str(source), '--to', target, '--quantization', 'full'])
    assert code == 0, err
    assert 'Expetion while parsing' in out                            # (the reference's spelling)
    assert '2 items are discarded due to inconsistent vector length' in out
    assert 'Exception (Attempt to add duplicate word w3 to index) while trying to add word' in out
    reader = native.Reader(target, device='cpu')
    assert reader.dim == 5 and reader.keys() == sorted(vectors)
    for word, vector in vectors.items():
        assert bits_equal(reader[word], vector), word
    # --max-words counts LINES of the source, header included (islice in the reference)
    code, out, err = run(['--from', str(source), '--to', target, '--quantization', 'trained', '--bits-per-weight', '6',
                          '--max-words', '6'])
    assert code == 0, err
    assert native.Reader(target, device='cpu').keys() == sorted('w{}'.format(i) for i in range(5))
    # refusals of the command line
    assert run(['--from', str(source), '--to', target, '--quantization', 'zip'])[0] != 0
    assert run(['--to', target, '--quantization', 'full'])[0] != 0


@pytest.mark.gpu
def test_converter_on_the_device_writes_the_same_file(native, tmp_path):
    rng = np.random.default_rng(5)
    count, dim = 12000, 24
    matrix = (rng.standard_normal((count, dim)) * 0.4).astype(np.float32)
    source = tmp_path / 'vectors.txt'
    with open(source, 'w') as f:
        for i in range(count):
            f.write('word{} '.format(i) + ' '.join('%.6f' % x for x in matrix[i]) + '\n')
    paths = []
    for device in (None, 0):
        target = str(tmp_path / 'trained_{}.bin'.format(device))
        arguments = ['--from', str(source), '--to', target, '--quantization', 'trained']
        code, out, err = run(arguments + (['--device', str(device)] if device is not None else []))
        assert code == 0, err
        paths.append(target)
    assert open(paths[0], 'rb').read() == open(paths[1], 'rb').read()
