"""The other trainers' step functions and the validation pass on CPU (TEST INFRASTRUCTURE, see oracle/__init__.py).

Restates /root/reference enhance_base_train.py:85-95, enhance_fbank_train.py:108-134, enhance_gan_train.py:123-150,
asr_train.py:118-131 and the no-grad validation pass joint_train.py:237-275 with the functional nets of
oracle/nets.py.  Pinned by tests/golden/trainers_tiny.npz (tests/test_oracle_golden.py)."""
import math

import torch
import torch.nn.functional as F

from . import nets
from .joint import _grads, _zero, clip_grad_norm


def leaf(d):
    return {k: v.clone().float().requires_grad_(True) for k, v in d.items() if v.dtype.is_floating_point and 'running_' not in k}


def buffers(d):
    return {k: v.clone() for k, v in d.items() if 'running_' in k or 'num_batches' in k}


def feature_loss(kind, a, b):
    return {'L2': F.mse_loss, 'L1': F.l1_loss, 'smooth_L1': F.smooth_l1_loss}[kind](a, b.detach())


def _update(params, opt, max_norm):
    g = _grads(params)
    gn = clip_grad_norm(list(g.values()), max_norm)
    if not math.isnan(gn):
        opt.step(g)
    return gn


def enhance_base_step(enh, opt_enh, batch, cfg):
    """enhance_base_train.py:85-95.  batch = (clean, mix, mix_log, cos, lens)."""
    clean, mix, mix_log, cos, lens = batch
    loss, enhance_out = nets.enhance_forward(enh, mix, mix_log, lens, cfg['enhance_layers'], clean, cos)
    _zero(enh)
    loss.backward()
    gn = _update(enh, opt_enh, cfg['grad_clip'])
    return dict(loss=loss.detach(), grad_norm=gn, enhance_out=enhance_out.detach())


def enhance_fbank_step(enh, opt_enh, W, batch, cfg):
    """enhance_fbank_train.py:108-134."""
    clean, mix, mix_log, cos, lens = batch
    enhance_out = nets.enhance_forward(enh, mix, mix_log, lens, cfg['enhance_layers'])
    loss = feature_loss(cfg.get('enhance_loss_type', 'L2'), nets.fbank_forward(enhance_out, W), nets.fbank_forward(clean, W))
    _zero(enh)
    loss.backward()
    gn = _update(enh, opt_enh, cfg['grad_clip'])
    return dict(loss=loss.detach(), grad_norm=gn, enhance_out=enhance_out.detach())


def enhance_gan_step(enh, gan, gan_buf, opt_enh, opt_gan, W, batch, cmvn, cfg):
    """enhance_gan_train.py:123-150 (D sees CMVN-normalised features, 1-argument GANModel.forward)."""
    clean, mix, mix_log, cos, lens = batch
    enhance_loss, enhance_out = nets.enhance_forward(enh, mix, mix_log, lens, cfg['enhance_layers'], clean, cos)
    enhance_feat = nets.fbank_forward(enhance_out, W, cmvn)
    clean_feat = nets.fbank_forward(clean, W, cmvn)
    for v in gan.values():
        v.requires_grad_(False)
    gan_loss = nets.gan_loss(nets.discriminator_forward(gan, gan_buf, enhance_feat), True)
    _zero(enh)
    loss = enhance_loss + cfg['gan_loss_lambda'] * gan_loss
    loss.backward()
    gn = _update(enh, opt_enh, cfg['grad_clip'])
    for v in gan.values():
        v.requires_grad_(True)
    _zero(gan)
    loss_D = (nets.gan_loss(nets.discriminator_forward(gan, gan_buf, clean_feat.detach()), True)
              + nets.gan_loss(nets.discriminator_forward(gan, gan_buf, enhance_feat.detach()), False)) * 0.5
    loss_D.backward()
    gnD = _update(gan, opt_gan, cfg['grad_clip'])
    return dict(loss=loss.detach(), gan_loss=gan_loss.detach(), enhance_loss=enhance_loss.detach(), loss_D=loss_D.detach(),
                grad_norm=gn, grad_norm_D=gnD)


def asr_step(asr, opt_asr, feats, targets, lens, tlens, cfg):
    """asr_train.py:118-131 (E2E on fbank features)."""
    loss_ctc, loss_att, acc, _, _ = nets.e2e_forward(asr, feats, targets, lens, tlens, cfg['elayers'], cfg['mtlalpha'])
    loss = cfg['mtlalpha'] * loss_ctc + (1 - cfg['mtlalpha']) * loss_att
    _zero(asr)
    loss.backward()
    gn = _update(asr, opt_asr, cfg['grad_clip'])
    return dict(loss=loss.detach(), loss_ctc=loss_ctc.detach(), loss_att=loss_att.detach(), acc=acc, grad_norm=gn)


def joint_validate(enh, asr, gan, gan_buf, W, batch, cmvn, cfg):
    """joint_train.py:237-275: no-grad pass; the discriminator is NOT switched to eval there, so its BatchNorm uses batch
    statistics and moves its running statistics.  batch = (clean, mix, mix_log, targets, lens, tlens)."""
    clean, mix, mix_log, targets, lens, tlens = batch
    with torch.no_grad():
        enhance_out = nets.enhance_forward(enh, mix, mix_log, lens, cfg['enhance_layers'])
        enhance_feat, clean_feat = nets.fbank_forward(enhance_out, W), nets.fbank_forward(clean, W)
        enhance_loss = feature_loss(cfg.get('enhance_loss_type', 'L2'), enhance_feat, clean_feat)
        out = {}
        if cfg.get('isGAN', True):
            gan_loss = nets.gan_loss(nets.discriminator_forward(gan, gan_buf, enhance_feat, cmvn), True)
            enhance_loss = enhance_loss + cfg['gan_loss_lambda'] * gan_loss
            out['gan_loss'] = cfg['gan_loss_lambda'] * gan_loss
        loss_ctc, loss_att, acc, _, _ = nets.share_e2e_forward(asr, clean_feat, enhance_feat, targets, lens, tlens, cfg['elayers'],
                                                               cmvn, cfg['mtlalpha'])
        enhance_loss = cfg['enhance_loss_lambda'] * enhance_loss
        loss = cfg['mtlalpha'] * loss_ctc + (1 - cfg['mtlalpha']) * loss_att + enhance_loss
        nf = (lambda z: z) if cmvn is None else (lambda z: (z + cmvn[0, :]) * cmvn[1, :])
        hpad, hlens = nets.encoder_forward(asr, nf(enhance_feat), lens, cfg['elayers'])
        V = asr['dec.output.weight'].size(0)
        ys = nets.split_targets(targets, tlens)
        greedy = [i > 0 for i in range(max(len(y) for y in ys) + 1)]          # e2e_decoder.py:408-412
        _, _, att = nets.decoder_forward(asr, hpad, hlens, ys, V - 1, return_att=True, sample_steps=greedy)
    out.update(loss=loss, loss_ctc=loss_ctc, loss_att=loss_att, acc=acc, enhance_loss=enhance_loss, att_ws=att)
    return out
