#!/usr/bin/env python3
"""Golden vectors for the Kaldi table reader: a small ark file with float, double and 8-bit compressed
matrix records (written by robust_e2e_gan_amd.data.kaldi_io) and the matrices the REFERENCE's reader
(/root/reference/data/kaldi_io.py, imported here) decodes from it.  Runs only in the build container."""
import importlib.util
import io
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from robust_e2e_gan_amd.data import kaldi_io as kio   # noqa: E402


def main():
    spec = importlib.util.spec_from_file_location('ref_kaldi_io', '/root/reference/data/kaldi_io.py')
    ref = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref)
    rng = np.random.default_rng(3)
    buf = io.BytesIO()
    shapes = [(37, 13), (20, 13), (64, 13), (5, 13), (129, 13)]
    for i, (r, c) in enumerate(shapes):
        spect = (np.abs(rng.standard_normal((r, c)) + 1j * rng.standard_normal((r, c))) * 300).astype(np.float32)
        spect[rng.random((r, c)) < 0.05] = 0.0                       # exercises the 1e-7 clamp
        if i % 3 == 0:
            kio.write_mat_compressed(buf, spect, 'cm_%d' % i)
        elif i % 3 == 1:
            kio.write_mat(buf, spect, 'fm_%d' % i)
        else:
            kio.write_mat(buf, spect.astype(np.float64), 'dm_%d' % i)
    ark = buf.getvalue()
    path = os.path.join('/tmp', 'kaldi_tiny.ark')
    with open(path, 'wb') as f:
        f.write(ark)
    fx = dict(ark=np.frombuffer(ark, np.uint8))
    keys = []
    for key, mat in ref.read_mat_ark(path):
        keys.append(key)
        fx['mat.' + key] = np.asarray(mat, np.float32)
    fx['keys'] = np.array(keys)
    np.savez_compressed(os.path.join(HERE, 'kaldi_tiny.npz'), **fx)
    print('written kaldi_tiny.npz', keys)


if __name__ == '__main__':
    main()
