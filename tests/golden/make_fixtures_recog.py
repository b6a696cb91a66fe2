#!/usr/bin/env python3
"""Golden vectors for decoding: n-best lists of the reference's own beam search
(E2E.recognize, model/e2e_model.py:204-236 -> Decoder.recognize_beam, model/e2e_decoder.py:171-369, with
CTCPrefixScore, model/e2e_ctc.py:78-155) on a tiny random-initialised model, for several search configurations.
Runs only in the build container (imports /root/reference); writes recog_tiny.npz."""
import argparse
import os
import random
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_fixtures as mf   # noqa: E402

CONFIGS = [  # name, beam, penalty, ctc_weight, maxlenratio, minlenratio, nbest
    ('att_b3', 3, 0.0, 0.0, 0.0, 0.0, 3),
    ('joint_b4', 4, 0.1, 0.3, 0.0, 0.0, 4),
    ('joint_ratio', 3, 0.0, 0.5, 0.6, 0.2, 2),
    ('ctc_heavy', 2, 0.2, 0.9, 0.0, 0.0, 2),
]


def main():
    mf.install_shims()
    from model.e2e_model import E2E
    from model.feat_model import FbankModel
    opt = mf.tiny_opt()
    torch.manual_seed(606)
    random.seed(0)
    asr = E2E(opt)
    fb = FbankModel(opt)
    clean, mix, mix_log, cos = mf.synth_batch(3, [37, 29, 20], seed=11)
    cm = torch.stack([torch.linspace(10, 14, 80), torch.linspace(0.3, 0.6, 80)])
    feats = fb(clean, cm).detach()
    fx = dict(feats=feats.numpy(), lens=np.array([37, 29, 20], np.int32))
    fx.update(mf.sd_np('p.', asr))
    char_list = [str(i) for i in range(opt.odim)]
    for name, beam, penalty, ctcw, maxr, minr, nbest in CONFIGS:
        args = argparse.Namespace(beam_size=beam, penalty=penalty, ctc_weight=ctcw, maxlenratio=maxr, minlenratio=minr, nbest=nbest,
                                  lm_weight=0.0)
        for u, T in enumerate([37, 29, 20]):
            with torch.no_grad():
                hyps = asr.recognize(feats[u:u + 1, :T], args, char_list)
            L = max(len(h['yseq']) for h in hyps)
            seqs = np.full((len(hyps), L), -1, np.int64)
            for i, h in enumerate(hyps):
                seqs[i, :len(h['yseq'])] = h['yseq']
            fx['%s.u%d.yseq' % (name, u)] = seqs
            fx['%s.u%d.score' % (name, u)] = np.array([float(h['score']) for h in hyps], np.float64)
    np.savez_compressed(os.path.join(HERE, 'recog_tiny.npz'), **fx)
    for k in sorted(fx):
        if k.endswith('yseq'):
            print(k, fx[k].tolist(), fx[k.replace('yseq', 'score')].round(4).tolist())


if __name__ == '__main__':
    main()
