#!/usr/bin/env python3
"""F1 pinned to the reference: collate_tiny.npz holds the OUTPUT of the reference's own
``data.mix_data_loader._collate_fn`` (mix_data_loader.py:264-302) and ``data.data_loader._collate_fn``
(data_loader.py:236-265) on hand-built ragged samples.  Build container only (needs /root/reference).

The loader modules import librosa / torchaudio / python_speech_features / soundfile / scipy.io.wavfile front-ends and a
py3.5-ABI ``extract_fbanks_module.so`` that do not exist here; none of them is touched by ``_collate_fn`` (40 lines of
pure torch), so they are replaced by permissive stub modules for the import only.

    python tests/golden/make_fixtures_collate.py      # rewrites tests/golden/collate_tiny.npz

Cases: 5 samples with a TIE in lengths (python's sort is stable: ties keep their input order), one sample with an EMPTY
target list, one single-frame sample; the ASR twin on 3 samples.
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_fixtures as mf  # noqa: E402


class _Anything(types.ModuleType):
    """a module whose every attribute exists (import-time names of front-ends that _collate_fn never calls)"""

    def __getattr__(self, name):
        if name.startswith('__'):
            raise AttributeError(name)
        return _Anything(self.__name__ + '.' + name)

    def __call__(self, *a, **k):
        raise RuntimeError('stubbed front-end %s was called' % self.__name__)


def main():
    mf.install_shims()
    for name in ('librosa', 'torchaudio', 'python_speech_features', 'soundfile', 'data.extract_fbanks_module', 'tqdm'):
        if name not in sys.modules:
            try:
                __import__(name)
            except Exception:
                sys.modules[name] = _Anything(name)
    from data.mix_data_loader import _collate_fn as mix_collate
    from data.data_loader import _collate_fn as asr_collate

    g = torch.Generator().manual_seed(9)
    F_ = 6
    slens = [4, 7, 5, 7, 1]                       # samples 1 and 3 tie
    tgts = [[3, 1], [2, 5, 4], [], [9], [7, 7, 8, 1]]
    samples = []
    for i, l in enumerate(slens):
        s = [torch.rand(l, F_, generator=g) for _ in range(5)]
        samples.append(('utt%d' % i, 'spk%d' % (i % 2), s[0], s[1], s[2], s[3], s[4], tgts[i]))
    out = mix_collate(list(samples))
    fx = {'n': np.int64(len(samples))}
    for i, smp in enumerate(samples):
        for k in range(5):
            fx['s%d_%d' % (i, k)] = smp[2 + k].numpy()
        fx['t%d' % i] = np.array(smp[7], np.int64)
    fx['order'] = np.array([int(u[3:]) for u in out[0]])
    fx['spk'] = np.array([int(s[3:]) for s in out[1]])
    fx['expected'] = np.stack([out[2 + k].numpy() for k in range(5)])
    fx['targets'] = out[7].numpy()
    fx['input_sizes'] = out[8].numpy()
    fx['target_sizes'] = out[9].numpy()
    assert out[7].dtype == torch.int64 and out[8].dtype == torch.int32 and out[9].dtype == torch.int32

    alens = [3, 5, 5]
    atg = [[1, 2], [3], [4, 4, 4]]
    asamples = [('a%d' % i, 's', torch.rand(l, 4, generator=g), torch.rand(l, 4, generator=g), atg[i]) for i, l in enumerate(alens)]
    ao = asr_collate(list(asamples))
    fx['asr.n'] = np.int64(len(asamples))
    for i, smp in enumerate(asamples):
        fx['asr.s%d_0' % i], fx['asr.s%d_1' % i] = smp[2].numpy(), smp[3].numpy()
        fx['asr.t%d' % i] = np.array(smp[4], np.int64)
    fx['asr.order'] = np.array([int(u[1:]) for u in ao[0]])
    fx['asr.expected'] = np.stack([ao[2].numpy(), ao[3].numpy()])
    fx['asr.targets'], fx['asr.input_sizes'], fx['asr.target_sizes'] = ao[4].numpy(), ao[5].numpy(), ao[6].numpy()
    np.savez_compressed(os.path.join(HERE, 'collate_tiny.npz'), **fx)
    print('collate_tiny.npz written from the reference _collate_fn; order', fx['order'], 'asr order', fx['asr.order'])


if __name__ == '__main__':
    main()
