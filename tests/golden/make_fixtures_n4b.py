#!/usr/bin/env python3
"""Golden vectors for the remaining non-default options, from the reference import:
  bce.*   GANModel 'basic' with --no_lsgan (Sigmoid head, gan_model.py:90-91,126) + GANLoss(use_lsgan=False) = nn.BCELoss
          (gan_model.py:157-160): real + fake, one backward, BatchNorm running statistics
  enhb.*  EnhanceModel with enhance_type 'blstmp' (BLSTMP with per-layer projection, enhance_model.py:90-93), mask-L1 loss, grads
  fbt.*   FbankModel with fbank_opti_type 'train' (the (257,80) mel matrix is a trainable dense parameter,
          feat_model.py:105-109): forward (+cmvn), dx and dW
  drop.*  E2E (tiny vggblstmp) with dropout_rate 0.3: the only site it reaches is the CTC head's F.dropout(hs_pad, p)
          (e2e_ctc.py:51; BLSTMP's per-layer nn.LSTM(num_layers=1, dropout=p) never drops).  torch's own generator cannot be
          reproduced on the GPU, so F.dropout is replaced FOR THIS RUN by the product's counter-based mask (oracle/philox.py,
          seed 20261003, mask 0, drawn over the time-major tensor): everything else is the reference's arithmetic.
  unet.*  EnhanceModel with enhance_type 'unet_128' (pix2pix U-Net, enhance_model.py:224-303; ngf 4, a (64, 32) image): mask
          product (upstream's double sigmoid included), mask-L1 loss, all gradients, BatchNorm running statistics.
          lecun_normal_init_parameters zeroes every 1-d parameter -- BatchNorm gains included (enhance_model.py:183) --, so
          the gains / shifts are set to non-trivial values before the run.
Build container only (needs /root/reference); writes n4b_tiny.npz."""
import argparse
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_fixtures as mf   # noqa: E402


def main():
    mf.install_shims()
    from model.enhance_model import EnhanceModel
    from model.feat_model import FbankModel
    from model.gan_model import GANModel, GANLoss
    opt = mf.tiny_opt()
    lens = [37, 29, 20]
    clean, mix, mix_log, cos = mf.synth_batch(3, lens, seed=11)
    input_sizes = torch.IntTensor(lens)
    cm = torch.stack([torch.linspace(10, 14, 80), torch.linspace(0.3, 0.6, 80)])
    feats = FbankModel(opt)(clean, cm).detach()
    fx = dict(lens=np.array(lens, np.int32), clean=clean.numpy(), mix=mix.numpy(), mix_log=mix_log.numpy(), cos=cos.numpy(), cmvn=cm.numpy(),
              feats=feats.numpy())

    # ---- --no_lsgan: sigmoid discriminator + BCE
    bce_opt = argparse.Namespace(**{**vars(opt), 'no_lsgan': True})
    torch.manual_seed(811)
    gan = GANModel(bce_opt)
    gan.train()
    crit = GANLoss(use_lsgan=False)
    fx.update(mf.sd_np('bce.p.', gan))
    xin = feats.clone().requires_grad_(True)
    d = gan(xin)
    l_real = crit(d, True)
    l_fake = crit(gan(xin * 0.9 + 0.1), False)
    gan.zero_grad()
    ((l_real + l_fake) * 0.5).backward()
    fx.update({'bce.d_out': d.detach().numpy(), 'bce.l_real': l_real.detach().numpy().reshape(-1),
               'bce.l_fake': l_fake.detach().numpy().reshape(-1), 'bce.dx': xin.grad.numpy()})
    fx.update(mf.grads_np('bce.g.', gan))
    fx.update(mf.sd_np('bce.after.', gan))
    assert float(d.min()) > 0.0 and float(d.max()) < 1.0

    # ---- blstmp enhancer
    eb_opt = argparse.Namespace(**{**vars(opt), 'enhance_type': 'blstmp', 'enhance_layers': 2, 'subsample': '1_1_1'})
    torch.manual_seed(812)
    enh = EnhanceModel(eb_opt)
    enh.train()
    fx.update(mf.sd_np('enhb.p.', enh))
    out = enh(mix, mix_log, input_sizes)
    loss, out2 = enh(mix, mix_log, input_sizes, clean, cos)
    enh.zero_grad()
    (loss + (out2 * torch.linspace(0.5, 1.5, 257)).mean()).backward()
    fx.update({'enhb.enhance_out': out.detach().numpy(), 'enhb.l1_loss': loss.detach().numpy().reshape(-1)})
    fx.update(mf.grads_np('enhb.g.', enh))

    # ---- trainable fbank matrix
    ft_opt = argparse.Namespace(**{**vars(opt), 'fbank_opti_type': 'train'})
    fb = FbankModel(ft_opt)
    assert fb.fc.requires_grad
    with torch.no_grad():                       # move the matrix off its banded initial value: every entry takes part
        g = torch.Generator().manual_seed(813)
        fb.fc.add_(torch.rand(fb.fc.shape, generator=g) * 1e-3)
    x = out.detach().clone().requires_grad_(True)
    y0 = fb(x)
    y1 = fb(x, cm)
    (y1 * torch.linspace(-1, 1, 80)).sum().backward()
    fx.update({'fbt.W': fb.fc.detach().numpy().copy(), 'fbt.x': x.detach().numpy(), 'fbt.y_nocmvn': y0.detach().numpy(),
               'fbt.y_cmvn': y1.detach().numpy(), 'fbt.dx': x.grad.numpy(), 'fbt.dW': fb.fc.grad.numpy().copy()})
    # ---- dropout through the CTC head
    import random
    import model.e2e_ctc as ref_ctc
    from model.e2e_model import E2E
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    from oracle.philox import dropout_mask
    SEED, calls = 20261003, [0]

    def recorded_dropout(x, p=0.5, training=True, inplace=False):
        B, T, E = x.shape
        m = torch.from_numpy(dropout_mask(B * T * E, p, SEED, calls[0])).view(T, B, E).transpose(0, 1)
        calls[0] += 1
        return x * m
    d_opt = argparse.Namespace(**{**vars(opt), 'dropout_rate': 0.3})
    tl = [5, 4, 3]
    g = torch.Generator().manual_seed(5)
    targets = torch.randint(1, opt.odim - 1, (sum(tl),), generator=g)
    torch.manual_seed(814)
    random.seed(0)
    asr = E2E(d_opt)
    asr.train()
    fx.update(mf.sd_np('drop.p.', asr))
    real = ref_ctc.F.dropout
    ref_ctc.F.dropout = recorded_dropout
    try:
        loss_ctc, loss_att, acc = asr(feats, targets, input_sizes, torch.IntTensor(tl), 0.0)
    finally:
        ref_ctc.F.dropout = real
    assert calls[0] == 1
    asr.zero_grad()
    (0.5 * loss_ctc + 0.5 * loss_att).backward()
    fx.update({'drop.targets': targets.numpy(), 'drop.tlens': np.array(tl, np.int32), 'drop.seed': np.int64(SEED),
               'drop.loss_ctc': loss_ctc.detach().numpy().reshape(-1), 'drop.loss_att': loss_att.detach().numpy().reshape(-1)})
    named = dict(asr.named_parameters())
    for n in ('ctc.ctc_lo.weight', 'ctc.ctc_lo.bias', 'enc.enc2.bt1.weight', 'enc.enc1.conv1_1.weight'):
        fx['drop.g.' + n] = named[n].grad.numpy().copy()
    # ---- U-Net enhancer
    u_opt = argparse.Namespace(**{**vars(opt), 'enhance_type': 'unet_128', 'idim': 32, 'enhance_input_nc': 1, 'enhance_output_nc': 1,
                                  'enhance_ngf': 4, 'enhance_norm': 'batch'})
    torch.manual_seed(815)
    unet = EnhanceModel(u_opt)
    unet.train()
    gq = torch.Generator().manual_seed(816)
    with torch.no_grad():
        for k, v in unet.named_parameters():
            if v.dim() == 1:
                v.copy_((1.0 if k.endswith('weight') else 0.0) + 0.2 * torch.randn(v.shape, generator=gq))
    fx.update(mf.sd_np('unet.p.', unet))
    ulens = [64, 40]
    uc, um, uml, ucos = mf.synth_batch(2, ulens, F_=32, seed=17)
    ul = torch.IntTensor(ulens)
    uout = unet(um, uml.unsqueeze(1), ul)
    uloss, uout2 = unet(um, uml.unsqueeze(1), ul, uc, ucos)
    unet.zero_grad()
    (uloss + (uout2 * torch.linspace(0.5, 1.5, 32)).mean()).backward()
    fx.update({'unet.lens': np.array(ulens, np.int32), 'unet.clean': uc.numpy(), 'unet.mix': um.numpy(), 'unet.mix_log': uml.numpy(),
               'unet.cos': ucos.numpy(), 'unet.enhance_out': uout.detach().numpy(), 'unet.l1_loss': uloss.detach().numpy().reshape(-1)})
    fx.update(mf.grads_np('unet.g.', unet))
    fx.update(mf.sd_np('unet.after.', unet))
    np.savez_compressed(os.path.join(HERE, 'n4b_tiny.npz'), **fx)
    print('written n4b_tiny.npz; bce losses', fx['bce.l_real'], fx['bce.l_fake'], 'enhb l1', fx['enhb.l1_loss'])


if __name__ == '__main__':
    main()
