#!/usr/bin/env python3
"""Golden vectors for the InstanceNorm variants, from the reference import -- round 3 checked them only
against networks assembled inside the tests:
  ind.*    GANModel 'basic' with --norm_D instance (gan_model.py:42-46,57: InstanceNorm2d(affine=False), convolutions with biases),
           LSGAN real + fake, one backward: output, both losses, input gradient, every parameter gradient
  inu.*    EnhanceModel 'unet_128' with --enhance_norm instance (enhance_model.py:258-261), ngf 4, a (64, 32) image: mask product,
           mask-L1 loss, every gradient
Build container only (needs /root/reference); writes n4c_tiny.npz."""
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
    clean, mix, mix_log, cos = mf.synth_batch(3, lens, seed=21)
    cm = torch.stack([torch.linspace(10, 14, 80), torch.linspace(0.3, 0.6, 80)])
    feats = FbankModel(opt)(clean, cm).detach()
    fx = dict(lens=np.array(lens, np.int32), cmvn=cm.numpy(), feats=feats.numpy())

    # ---- --norm_D instance
    d_opt = argparse.Namespace(**{**vars(opt), 'norm_D': 'instance'})
    torch.manual_seed(911)
    gan = GANModel(d_opt)
    gan.train()
    assert not any('running' in k for k in gan.state_dict())
    crit = GANLoss(use_lsgan=True)
    fx.update(mf.sd_np('ind.p.', gan))
    xin = feats.clone().requires_grad_(True)
    d = gan(xin)
    l_real = crit(d, True)
    l_fake = crit(gan(xin * 0.9 + 0.1), False)
    gan.zero_grad()
    ((l_real + l_fake) * 0.5).backward()
    fx.update({'ind.d_out': d.detach().numpy(), 'ind.l_real': l_real.detach().numpy().reshape(-1),
               'ind.l_fake': l_fake.detach().numpy().reshape(-1), 'ind.dx': xin.grad.numpy()})
    fx.update(mf.grads_np('ind.g.', gan))

    # ---- --enhance_norm instance (U-Net)
    u_opt = argparse.Namespace(**{**vars(opt), 'enhance_type': 'unet_128', 'idim': 32, 'enhance_input_nc': 1, 'enhance_output_nc': 1,
                                  'enhance_ngf': 4, 'enhance_norm': 'instance'})
    torch.manual_seed(915)
    unet = EnhanceModel(u_opt)
    unet.train()
    gq = torch.Generator().manual_seed(916)
    with torch.no_grad():                       # lecun_normal_init_parameters zeroes every 1-d parameter (the convolutions' biases): make them count
        for k, v in unet.named_parameters():
            if v.dim() == 1:
                v.copy_(0.2 * torch.randn(v.shape, generator=gq))
    assert not any('running' in k for k in unet.state_dict())
    fx.update(mf.sd_np('inu.p.', unet))
    ulens = [64, 40]
    uc, um, uml, ucos = mf.synth_batch(2, ulens, F_=32, seed=27)
    ul = torch.IntTensor(ulens)
    uout = unet(um, uml.unsqueeze(1), ul)
    uloss, uout2 = unet(um, uml.unsqueeze(1), ul, uc, ucos)
    unet.zero_grad()
    (uloss + (uout2 * torch.linspace(0.5, 1.5, 32)).mean()).backward()
    fx.update({'inu.lens': np.array(ulens, np.int32), 'inu.clean': uc.numpy(), 'inu.mix': um.numpy(), 'inu.mix_log': uml.numpy(),
               'inu.cos': ucos.numpy(), 'inu.enhance_out': uout.detach().numpy(), 'inu.l1_loss': uloss.detach().numpy().reshape(-1)})
    fx.update(mf.grads_np('inu.g.', unet))
    np.savez_compressed(os.path.join(HERE, 'n4c_tiny.npz'), **fx)
    print('written n4c_tiny.npz; ind losses', fx['ind.l_real'], fx['ind.l_fake'], 'inu l1', fx['inu.l1_loss'])


if __name__ == '__main__':
    main()
