#!/usr/bin/env python3
"""Golden vectors for two non-default options, from the reference import:
  sub.*  E2E with etype 'blstmp', 3 layers, subsample '1_2_2_1_1' ('skip' frame subsampling, model/e2e_encoder.py:133-139)
  lsm.*  E2E (tiny vggblstmp) with label smoothing (lsm_type 'unigram', lsm_weight 0.1, model/e2e_decoder.py:162-166)
Runs only in the build container; writes n4_tiny.npz."""
import argparse
import os
import random
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_fixtures as mf   # noqa: E402


def run(asr, feats, targets, input_sizes, target_sizes, fx, pre, names):
    loss_ctc, loss_att, acc = asr(feats, targets, input_sizes, target_sizes, 0.0)
    hpad, hlens = asr.enc(feats, input_sizes)
    asr.zero_grad()
    (0.5 * loss_ctc + 0.5 * loss_att).backward()
    fx.update({pre + 'loss_ctc': loss_ctc.detach().numpy().reshape(-1), pre + 'loss_att': loss_att.detach().numpy().reshape(-1),
               pre + 'acc': np.float64(acc), pre + 'hpad': hpad.detach().numpy(), pre + 'hlens': np.array(list(map(int, hlens)), np.int32)})
    named = dict(asr.named_parameters())
    for n in names:
        fx[pre + 'g.' + n] = named[n].grad.numpy().copy()


def main():
    mf.install_shims()
    from model.e2e_model import E2E
    from model.feat_model import FbankModel
    opt = mf.tiny_opt()
    lens, tl = [37, 29, 20], [5, 4, 3]
    clean, mix, mix_log, cos = mf.synth_batch(3, lens, seed=11)
    input_sizes, target_sizes = torch.IntTensor(lens), torch.IntTensor(tl)
    g = torch.Generator().manual_seed(5)
    targets = torch.randint(1, opt.odim - 1, (sum(tl),), generator=g)
    cm = torch.stack([torch.linspace(10, 14, 80), torch.linspace(0.3, 0.6, 80)])
    feats = FbankModel(opt)(clean, cm).detach()
    fx = dict(feats=feats.numpy(), targets=targets.numpy(), lens=np.array(lens, np.int32), tlens=np.array(tl, np.int32))

    sub_opt = argparse.Namespace(**{**vars(opt), 'etype': 'blstmp', 'elayers': 3, 'subsample': '1_2_2_1_1'})
    torch.manual_seed(707)
    random.seed(0)
    asr = E2E(sub_opt)
    asr.train()
    fx.update(mf.sd_np('sub.p.', asr))
    run(asr, feats, targets, input_sizes, target_sizes, fx, 'sub.',
        ['enc.enc1.bilstm0.weight_ih_l0', 'enc.enc1.bt1.weight', 'enc.enc1.bilstm2.weight_hh_l0_reverse', 'dec.output.weight', 'ctc.ctc_lo.weight'])

    mp_opt = argparse.Namespace(**{**vars(opt), 'etype': 'blstmp', 'elayers': 3, 'subsample': '1_2_2_1_1', 'subsample_type': 'maxpooling'})
    torch.manual_seed(707)
    random.seed(0)
    asr = E2E(mp_opt)            # same initial parameters as sub.p.
    asr.train()
    run(asr, feats, targets, input_sizes, target_sizes, fx, 'mp.',
        ['enc.enc1.bilstm0.weight_ih_l0', 'enc.enc1.bt1.weight', 'enc.enc1.bilstm2.weight_hh_l0_reverse', 'dec.output.weight', 'ctc.ctc_lo.weight'])

    dist = np.random.default_rng(4).dirichlet(np.ones(opt.odim)).astype(np.float32)
    lsm_opt = argparse.Namespace(**{**vars(opt), 'lsm_type': 'unigram', 'lsm_weight': 0.1, 'labeldist': dist})
    torch.manual_seed(708)
    random.seed(0)
    asr = E2E(lsm_opt)
    asr.train()
    fx['lsm.labeldist'] = dist
    fx.update(mf.sd_np('lsm.p.', asr))
    run(asr, feats, targets, input_sizes, target_sizes, fx, 'lsm.', ['dec.output.weight', 'dec.output.bias', 'dec.embed.weight', 'enc.enc2.bt0.weight'])
    # pixel discriminator (gan_model.py:98-116) on a (B, T, 160) "fake_AB" input, LSGAN real + fake, one backward
    from model.gan_model import GANModel, GANLoss
    pix_opt = argparse.Namespace(**{**vars(opt), 'netD_type': 'pixel'})
    torch.manual_seed(709)
    gan = GANModel(pix_opt)
    gan.train()
    crit = GANLoss(use_lsgan=True)
    fx.update(mf.sd_np('pix.p.', gan))
    xin = torch.cat([feats, feats * 0.5 + 0.1], 2).clone().requires_grad_(True)
    d = gan(xin)
    loss = (crit(d, True) + crit(gan(xin * 0.9), False)) * 0.5
    gan.zero_grad()
    loss.backward()
    fx.update({'pix.x': xin.detach().numpy(), 'pix.d_out': d.detach().numpy(), 'pix.loss': loss.detach().numpy().reshape(-1),
               'pix.dx': xin.grad.numpy()})
    fx.update(mf.grads_np('pix.g.', gan))
    fx.update(mf.sd_np('pix.after.', gan))
    np.savez_compressed(os.path.join(HERE, 'n4_tiny.npz'), **fx)
    print('written n4_tiny.npz; sub hlens', fx['sub.hlens'], 'lsm loss_att', fx['lsm.loss_att'])


if __name__ == '__main__':
    main()
