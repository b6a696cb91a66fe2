#!/usr/bin/env python3
"""float64 arbitration vectors: the REFERENCE modules run in double precision on the inputs and
initial parameters of joint_tiny.npz (build container only; needs /root/reference, see
make_fixtures.py for the shims).  The fp32 reference and the fp32 HIP path both differ from
these by rounding; gradient tolerances in tests/ are stated against THIS run so that neither
fp32 side is treated as exact.

    python tests/golden/make_fixtures_fp64.py      # writes tests/golden/joint_tiny_fp64.npz

Stored (all float64): losses, grad norms, every parameter gradient of the three nets for the composed
joint_train.py:156-212 step (S1-S3 semantics, same composition as make_fixtures.py), enhance_out.
"""
import os
import random
import sys

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_fixtures as mf  # noqa: E402


def main():
    mf.install_shims()
    torch.set_default_dtype(torch.float64)
    from model.enhance_model import EnhanceModel
    from model.feat_model import FbankModel
    from model.e2e_model import E2E
    from model.gan_model import GANModel, GANLoss
    from model.e2e_common import set_requires_grad

    opt = mf.tiny_opt()
    fx = dict(np.load(os.path.join(HERE, 'joint_tiny.npz')))
    t = lambda k: torch.from_numpy(fx[k]).double()
    lens, tl = fx['lens'].tolist(), fx['tlens'].tolist()
    B = len(lens)
    input_sizes, target_sizes = torch.IntTensor(lens), torch.IntTensor(tl)
    targets = torch.from_numpy(fx['targets'])
    random.seed(0)
    enh, asr, gan, fb = EnhanceModel(opt), E2E(opt), GANModel(opt), FbankModel(opt)

    def load(m, pre):
        sd = {k[len(pre):]: torch.from_numpy(v) for k, v in fx.items() if k.startswith(pre)}
        m.load_state_dict(sd)
        return m.double().train()

    enh, asr, gan = load(enh, 'enh.'), load(asr, 'asr.'), load(gan, 'gan.')
    W32 = dict(np.load(os.path.join(HERE, 'fbank_tiny.npz')))['W']
    fb.fc.data = torch.from_numpy(W32).double()
    fb = fb.double().train()
    crit = GANLoss(use_lsgan=True)
    cmvn = t('cmvn')
    mix, mix_log, clean = t('mix'), t('mix_log'), t('clean')

    enhance_out = enh(mix, mix_log, input_sizes)
    enhance_feat = fb(enhance_out)
    clean_feat = fb(clean)
    enhance_loss = opt.enhance_loss_lambda * F.mse_loss(enhance_feat, clean_feat.detach())
    nf = lambda z: (z + cmvn[0, :]) * cmvn[1, :]
    loss_ctc, loss_att, acc = asr(nf(enhance_feat), targets, input_sizes, target_sizes, 0.0)
    h_mix, hl = asr.enc(nf(enhance_feat), input_sizes)
    h_cln, _ = asr.enc(nf(clean_feat), input_sizes)
    ctx_mix = torch.cat([h_mix[i, :hl[i]] for i in range(B)], 0)
    ctx_cln = torch.cat([h_cln[i, :hl[i]] for i in range(B)], 0)
    coral_loss = opt.coral_loss_lambda * mf.coral(ctx_cln, ctx_mix)
    asr_loss = opt.mtlalpha * loss_ctc + (1 - opt.mtlalpha) * loss_att
    loss = asr_loss + enhance_loss + coral_loss
    set_requires_grad([gan], False)
    gan_loss = opt.gan_loss_lambda * crit(gan(nf(enhance_feat)), True)
    loss = loss + gan_loss
    enh.zero_grad()
    asr.zero_grad()
    loss.backward()
    out = {}
    out.update(mf.grads_np('genh.', enh))
    out.update(mf.grads_np('gasr.', asr))
    gn = float(torch.sqrt(sum((p.grad ** 2).sum() for p in asr.parameters() if p.grad is not None)))
    set_requires_grad([gan], True)
    gan.zero_grad()
    l_real = crit(gan(nf(clean_feat.detach())), True)
    l_fake = crit(gan(nf(enhance_feat.detach())), False)
    loss_D = (l_real + l_fake) * 0.5
    loss_D.backward()
    out.update(mf.grads_np('ggan.', gan))
    gnD = float(torch.sqrt(sum((p.grad ** 2).sum() for p in gan.parameters() if p.grad is not None)))
    for k, v in out.items():
        assert v.dtype == np.float64, (k, v.dtype)
    out.update(enhance_out=enhance_out.detach().numpy(), enhance_feat=enhance_feat.detach().numpy(),
               loss=np.float64(loss.item()), loss_ctc=np.float64(loss_ctc.item()), loss_att=np.float64(loss_att.item()),
               enhance_loss=np.float64(enhance_loss.item()), coral_loss=np.float64(coral_loss.item()),
               gan_loss=np.float64(gan_loss.item()), loss_D=np.float64(loss_D.item()), acc=np.float64(acc),
               grad_norm_asr=np.float64(gn), grad_norm_gan=np.float64(gnD))
    # sanity: the fp32 reference run (joint_tiny.npz) agrees with this run to fp32 rounding
    for k in ('loss', 'loss_ctc', 'loss_att', 'enhance_loss', 'coral_loss', 'gan_loss', 'loss_D'):
        a, b = float(out[k]), float(fx[k][0])
        assert abs(a - b) <= 2e-5 * max(1.0, abs(a)), (k, a, b)
    np.savez_compressed(os.path.join(HERE, 'joint_tiny_fp64.npz'), **out)
    worst = max(np.abs(out[k] - fx[k]).max() / np.abs(out[k]).max() for k in out if k.startswith('g') and k in fx and out[k].ndim)
    print('written; worst fp32-reference gradient deviation from fp64 (rel. to tensor max): %.3e' % worst)


if __name__ == '__main__':
    main()
