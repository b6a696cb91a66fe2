#!/usr/bin/env python3
"""Generate golden input/output vectors by importing the REFERENCE python modules.

Runs ONLY in the build container (needs /root/reference, read-only).  Nothing from the
reference is copied: the script imports its modules with the five shims of SURVEY.md
section 8c (numpy.int alias, stub modules, warp-ctc stand-in built on F.ctc_loss, uint8 mask
cast) and records tensors.  The .npz files written next to this script are DATA (inputs,
initial parameters, expected outputs / gradients) and travel to the GPU box; this script and
/root/reference do not need to exist there.

    python tests/golden/make_fixtures.py        # rewrites tests/golden/*.npz

Fixtures
  enhance_tiny.npz  EnhanceModel (blstm) forward, mask-L1 loss, grads        (F2)
  fbank_tiny.npz    FbankModel forward (+cmvn), input grad, compute_cmvn     (F3, F4)
  e2e_tiny.npz      E2E (vggblstmp + CTC + AttLoc + decoder) fwd/bwd         (F5-F10)
  gan_tiny.npz      GANModel 'basic' + GANLoss fwd/bwd, BN running stats     (F11, F12)
  joint_tiny.npz    one composed joint_train step with S1-S3 semantics       (F13)
  (collate_tiny.npz, F1, is written by make_fixtures_collate.py from the reference's own _collate_fn)
"""
import argparse
import os
import random
import sys
import types

import numpy as np
import torch
import torch.nn.functional as F

REF = '/root/reference'
HERE = os.path.dirname(os.path.abspath(__file__))


def install_shims():
    sys.path.insert(0, REF)
    np.int = int  # shim 1

    def stub(name, **kw):
        m = types.ModuleType(name)
        m.__dict__.update(kw)
        sys.modules[name] = m

    class PB:  # shim 2
        def start(self):
            return self

        def update(self, *a):
            pass

        def finish(self):
            pass

    stub('progressbar', ProgressBar=PB)
    stub('jiwer', wer=lambda a, b: 0.0)
    stub('kenlm')

    class WarpCTC(torch.nn.Module):  # shim 3 (acts: T x B x V raw logits)
        def __init__(self, size_average=False, length_average=False):
            super().__init__()
            self.sa = size_average

        def forward(self, acts, labels, act_lens, label_lens):
            nll = F.ctc_loss(acts.log_softmax(2), labels.long(), act_lens.long(),
                             label_lens.long(), blank=0, reduction='sum')
            return (nll / acts.size(1) if self.sa else nll).view(1)

    stub('warpctc_pytorch', CTCLoss=WarpCTC)
    _mf = torch.Tensor.masked_fill  # shim 4
    torch.Tensor.masked_fill = lambda self, m, v: _mf(self, m.bool(), v)


def tiny_opt():
    V = 12
    return argparse.Namespace(
        idim=257, odim=V, fbank_dim=80, char_list=[str(i) for i in range(V)], gpu_ids=[],
        verbose=0, enhance_type='blstm', enhance_layers=2, enhance_units=16, enhance_projs=16,
        dropout_rate=0.0, subsample_type='skip', subsample='1_1_1_1_1',
        fbank_opti_type='frozen', train_dataset_len=6, num_utt_cmvn=20000,
        etype='vggblstmp', elayers=2, eunits=24, eprojs=20, atype='location', adim=18,
        aconv_chans=4, aconv_filts=5, awin=5, aheads=4, dlayers=1, dunits=14, mtlalpha=0.5,
        lsm_type='', lsm_weight=0.0, labeldist=None, fusion='', lmtype=None, rnnlm=None,
        ndf=8, norm_D='batch', input_nc=1, n_layers_D=3, no_lsgan=False, netD_type='basic',
        enhance_loss_type='L2', enhance_loss_lambda=1.0, coral_loss_lambda=0.5,
        gan_loss_lambda=1.0, grad_clip=5.0, eps=1e-8, isGAN=True)


def synth_batch(B, lens, F_=257, seed=0):
    g = torch.Generator().manual_seed(seed)
    T = max(lens)
    clean = torch.zeros(B, T, F_)
    mix = torch.zeros(B, T, F_)
    cos = torch.zeros(B, T, F_)
    for b, l in enumerate(lens):
        c = torch.complex(torch.randn(l, F_, generator=g), torch.randn(l, F_, generator=g)) * 30.0
        n = torch.complex(torch.randn(l, F_, generator=g), torch.randn(l, F_, generator=g)) * 10.0
        clean[b, :l] = c.abs()
        mix[b, :l] = (c + n).abs()
        cos[b, :l] = torch.cos(torch.angle(c) - torch.angle(c + n))
    mix_log = torch.zeros(B, T, F_)
    for b, l in enumerate(lens):
        x = 10.0 * torch.log10(torch.clamp(mix[b, :l], min=1e-7))
        mix_log[b, :l] = (x - x.mean(0, keepdim=True)) / x.std(0, keepdim=True)
    return clean, mix, mix_log, cos


def sd_np(prefix, module):
    return {prefix + k: v.detach().cpu().numpy().copy() for k, v in module.state_dict().items()}


def grads_np(prefix, module):
    out = {}
    for k, p in module.named_parameters():
        if p.grad is not None:
            out[prefix + k] = p.grad.detach().numpy().copy()
    return out


def coral(src, tgt):
    """S2 (build-defined): Deep-CORAL on (n,d) rows."""
    d = src.size(1)

    def cov(x):
        n = x.size(0)
        xm = x - x.mean(0, keepdim=True)
        return xm.t().mm(xm) / (n - 1)

    diff = cov(src) - cov(tgt)
    return (diff * diff).sum() / (4.0 * d * d)


def main():
    install_shims()
    from model.enhance_model import EnhanceModel
    from model.feat_model import FbankModel
    from model.e2e_model import E2E
    from model.gan_model import GANModel, GANLoss
    from model.e2e_common import set_requires_grad

    opt = tiny_opt()
    lens = [37, 29, 20]
    tl = [5, 4, 3]
    B = 3
    clean, mix, mix_log, cos = synth_batch(B, lens, seed=11)
    input_sizes = torch.IntTensor(lens)
    target_sizes = torch.IntTensor(tl)
    g = torch.Generator().manual_seed(5)
    targets = torch.randint(1, opt.odim - 1, (sum(tl),), generator=g)

    # ---------------- EnhanceModel -----------------------------------------------------
    torch.manual_seed(101)
    enh = EnhanceModel(opt)
    enh.train()
    out = enh(mix, mix_log, input_sizes)
    loss, out2 = enh(mix, mix_log, input_sizes, clean, cos)
    enh.zero_grad()
    (loss + (out2 * torch.linspace(0.5, 1.5, 257)).mean()).backward()
    fx = dict(mix=mix.numpy(), mix_log=mix_log.numpy(), clean=clean.numpy(), cos=cos.numpy(),
              lens=np.array(lens, np.int32), enhance_out=out.detach().numpy(),
              l1_loss=loss.detach().numpy())
    fx.update(sd_np('p.', enh))
    fx.update(grads_np('g.', enh))
    np.savez_compressed(os.path.join(HERE, 'enhance_tiny.npz'), **fx)

    # ---------------- FbankModel -------------------------------------------------------
    fb = FbankModel(opt)
    x = out.detach().clone().requires_grad_(True)
    cm = torch.stack([torch.linspace(10, 14, 80), torch.linspace(0.3, 0.6, 80)])
    y0 = fb(x)
    y1 = fb(x, cm)
    (y1 * torch.linspace(-1, 1, 80)).sum().backward()
    fbc = FbankModel(argparse.Namespace(**{**vars(opt), 'train_dataset_len': 3}))
    r1 = fbc.compute_cmvn(out.detach(), input_sizes)   # accumulates 3 utts -> None
    r2 = fbc.compute_cmvn(out.detach(), input_sizes)   # now returns the estimate
    assert r1 is None and r2 is not None
    np.savez_compressed(os.path.join(HERE, 'fbank_tiny.npz'), x=x.detach().numpy(),
                        W=fb.fc.detach().numpy(), cmvn=cm.numpy(), y_nocmvn=y0.detach().numpy(),
                        y_cmvn=y1.detach().numpy(), dx=x.grad.numpy(),
                        lens=np.array(lens, np.int32), cmvn_est=r2.copy())

    # ---------------- E2E --------------------------------------------------------------
    torch.manual_seed(202)
    random.seed(0)
    asr = E2E(opt)
    asr.train()
    feat = fb(out.detach(), cm).detach().clone().requires_grad_(True)
    loss_ctc, loss_att, acc = asr(feat, targets, input_sizes, target_sizes, 0.0)
    hpad, hlens = asr.enc(feat, input_sizes)
    asr.zero_grad()
    (0.5 * loss_ctc + 0.5 * loss_att).backward()
    fx = dict(feat=feat.detach().numpy(), targets=targets.numpy(), lens=np.array(lens, np.int32),
              tlens=np.array(tl, np.int32), loss_ctc=loss_ctc.detach().numpy(),
              loss_att=loss_att.detach().numpy(), acc=np.float64(acc),
              hpad=hpad.detach().numpy(), hlens=np.array(list(hlens), np.int32),
              dfeat=feat.grad.numpy())
    fx.update(sd_np('p.', asr))
    fx.update(grads_np('g.', asr))
    np.savez_compressed(os.path.join(HERE, 'e2e_tiny.npz'), **fx)

    # ---------------- GANModel ---------------------------------------------------------
    torch.manual_seed(303)
    gan = GANModel(opt)
    gan.train()
    crit = GANLoss(use_lsgan=True)
    xin = fb(out.detach(), cm).detach().clone().requires_grad_(True)
    p0 = sd_np('p.', gan)
    d = gan(xin)
    l_real = crit(d, True)
    l_fake = crit(gan(xin * 0.9 + 0.1), False)
    gan.zero_grad()
    ((l_real + l_fake) * 0.5).backward()
    fx = dict(x=xin.detach().numpy(), d_out=d.detach().numpy(), l_real=l_real.detach().numpy(),
              l_fake=l_fake.detach().numpy(), dx=xin.grad.numpy())
    fx.update(p0)
    fx.update(grads_np('g.', gan))
    fx.update(sd_np('after.', gan))
    np.savez_compressed(os.path.join(HERE, 'gan_tiny.npz'), **fx)

    # ---------------- composed joint step (joint_train.py:156-212, S1-S3) ---------------
    torch.manual_seed(404)
    random.seed(0)
    enh = EnhanceModel(opt)
    asr = E2E(opt)
    gan = GANModel(opt)
    fb = FbankModel(opt)
    for m in (enh, asr, gan, fb):
        m.train()
    crit = GANLoss(use_lsgan=True)
    p_init = {}
    p_init.update(sd_np('enh.', enh))
    p_init.update(sd_np('asr.', asr))
    p_init.update(sd_np('gan.', gan))
    cmvn = cm
    eopt = torch.optim.Adadelta(enh.parameters(), rho=0.95, eps=opt.eps)
    aopt = torch.optim.Adadelta(asr.parameters(), rho=0.95, eps=opt.eps)
    gopt = torch.optim.Adadelta(gan.parameters(), rho=0.95, eps=opt.eps)

    enhance_out = enh(mix, mix_log, input_sizes)
    enhance_feat = fb(enhance_out)
    clean_feat = fb(clean)
    enhance_loss = opt.enhance_loss_lambda * F.mse_loss(enhance_feat, clean_feat.detach())
    # S1: ShareE2E = E2E on CMVN-normalised features; encoder shared by both branches
    nf = lambda z: (z + cmvn[0, :]) * cmvn[1, :]
    loss_ctc, loss_att, acc = asr(nf(enhance_feat), targets, input_sizes, target_sizes, 0.0)
    h_mix, hl = asr.enc(nf(enhance_feat), input_sizes)
    h_cln, _ = asr.enc(nf(clean_feat), input_sizes)
    ctx_mix = torch.cat([h_mix[i, :hl[i]] for i in range(B)], 0)
    ctx_cln = torch.cat([h_cln[i, :hl[i]] for i in range(B)], 0)
    coral_loss = opt.coral_loss_lambda * coral(ctx_cln, ctx_mix)
    asr_loss = opt.mtlalpha * loss_ctc + (1 - opt.mtlalpha) * loss_att
    loss = asr_loss + enhance_loss + coral_loss
    set_requires_grad([gan], False)
    gan_loss = opt.gan_loss_lambda * crit(gan(nf(enhance_feat)), True)   # S3
    loss = loss + gan_loss
    eopt.zero_grad()
    aopt.zero_grad()
    loss.backward()
    g_enh = grads_np('genh.', enh)
    g_asr = grads_np('gasr.', asr)
    gn = torch.nn.utils.clip_grad_norm_(asr.parameters(), opt.grad_clip)
    eopt.step()
    aopt.step()
    set_requires_grad([gan], True)
    gopt.zero_grad()
    l_real = crit(gan(nf(clean_feat.detach())), True)
    l_fake = crit(gan(nf(enhance_feat.detach())), False)
    loss_D = (l_real + l_fake) * 0.5
    loss_D.backward()
    g_gan = grads_np('ggan.', gan)
    gnD = torch.nn.utils.clip_grad_norm_(gan.parameters(), opt.grad_clip)
    gopt.step()

    fx = dict(mix=mix.numpy(), mix_log=mix_log.numpy(), clean=clean.numpy(),
              lens=np.array(lens, np.int32), tlens=np.array(tl, np.int32),
              targets=targets.numpy(), cmvn=cmvn.numpy(),
              enhance_out=enhance_out.detach().numpy(), enhance_feat=enhance_feat.detach().numpy(),
              loss=loss.detach().numpy().reshape(-1), loss_ctc=loss_ctc.detach().numpy().reshape(-1),
              loss_att=loss_att.detach().numpy().reshape(-1), acc=np.float64(acc),
              enhance_loss=enhance_loss.detach().numpy().reshape(-1),
              coral_loss=coral_loss.detach().numpy().reshape(-1),
              gan_loss=gan_loss.detach().numpy().reshape(-1),
              loss_D=loss_D.detach().numpy().reshape(-1),
              grad_norm_asr=np.float64(gn), grad_norm_gan=np.float64(gnD))
    fx.update(p_init)
    fx.update(g_enh)
    fx.update(g_asr)
    fx.update(g_gan)
    fx.update(sd_np('enh_after.', enh))
    fx.update(sd_np('asr_after.', asr))
    fx.update(sd_np('gan_after.', gan))
    np.savez_compressed(os.path.join(HERE, 'joint_tiny.npz'), **fx)

    # collate_tiny.npz (F1) comes from the reference's own _collate_fn: make_fixtures_collate.py
    print('fixtures written to', HERE)


if __name__ == '__main__':
    main()
