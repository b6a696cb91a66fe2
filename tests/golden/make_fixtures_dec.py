#!/usr/bin/env python3
"""Golden vectors for the decoder loop at widths the PERSISTENT form (csrc/decloop.hip) accepts, from the reference import.

Every other reference-vector fixture uses dunits=14 (make_fixtures.tiny_opt), which the resident loop declines (D % 4 != 0): those
tests exercise the launch-per-token kernels.  Here the reference's own Decoder + AttLoc (model/e2e_decoder.py:27-168,
model/e2e_attention.py:199-299) run teacher-forced on synthetic encoder states with dunits=16, eprojs=32 (the resident backward wants eprojs % 16 == 0), adim=20:
  a.*   B=3, T'=40, ragged lengths (40, 31, 22), label lengths (5, 4, 3)
  b.*   B=2, T'=300 (> 256 frames: two frame chunks per utterance in the resident form), lengths (300, 270), labels (4, 3)
recorded per case: parameters, hpad, hlens, ys, loss, accuracy, the attention weights of every step (forward hook on the
attention module), d(loss)/d(hpad) and every parameter gradient.  Build container only (needs /root/reference); writes
dec_persist_tiny.npz."""
import os
import random
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_fixtures as mf   # noqa: E402

EPROJS, DUNITS, ADIM, CHANS, FILTS, ODIM = 32, 16, 20, 4, 5, 12


def run_case(tag, lens, tl, seed):
    from model.e2e_attention import AttLoc
    from model.e2e_decoder import Decoder
    torch.manual_seed(seed)
    random.seed(0)
    att = AttLoc(EPROJS, DUNITS, ADIM, CHANS, FILTS, 'softmax')
    dec = Decoder(EPROJS, ODIM, 1, DUNITS, ODIM - 1, ODIM - 1, att, 0, [str(i) for i in range(ODIM)], None, 0.0, None, None, 'char', 0.0)
    g = torch.Generator().manual_seed(seed + 1)
    with torch.no_grad():                       # biases away from zero so that every term counts
        for k, v in dec.named_parameters():
            if v.dim() == 1:
                v.copy_(0.2 * torch.randn(v.shape, generator=g))
    dec.train()
    B, T = len(lens), max(lens)
    hpad = (0.7 * torch.randn(B, T, EPROJS, generator=g)).requires_grad_(True)
    ys = [torch.randint(1, ODIM - 1, (n,), generator=g) for n in tl]
    atts = []
    hook = att.register_forward_hook(lambda m, i, o: atts.append(o[1].detach().numpy().copy()))
    loss, acc = dec(hpad, lens, ys, 0.0)
    hook.remove()
    dec.zero_grad()
    loss.backward()
    fx = {tag + 'hpad': hpad.detach().numpy(), tag + 'hlens': np.array(lens, np.int32), tag + 'tlens': np.array(tl, np.int32),
          tag + 'ys': torch.cat(ys).numpy(), tag + 'loss_att': loss.detach().numpy().reshape(-1), tag + 'acc': np.float64(acc),
          tag + 'att_w': np.stack(atts, 1), tag + 'd_hpad': hpad.grad.numpy()}
    fx.update(mf.sd_np(tag + 'p.dec.', dec))
    fx.update(mf.grads_np(tag + 'g.dec.', dec))
    return fx


def main():
    mf.install_shims()
    fx = {}
    fx.update(run_case('a.', [40, 31, 22], [5, 4, 3], 1201))
    fx.update(run_case('b.', [300, 270], [4, 3], 1301))
    np.savez_compressed(os.path.join(HERE, 'dec_persist_tiny.npz'), **fx)
    print('written dec_persist_tiny.npz; losses', fx['a.loss_att'], fx['b.loss_att'], 'att_w', fx['a.att_w'].shape, fx['b.att_w'].shape)
    print(sorted(k for k in fx if k.startswith('a.p.')))


if __name__ == '__main__':
    main()
