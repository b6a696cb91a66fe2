"""One joint_train step on CPU (TEST INFRASTRUCTURE, see oracle/__init__.py).

Restates /root/reference joint_train.py:156-212 with the S1-S3 semantics of SURVEY 8a.
Used (a) by tests as the checker for the HIP path and (b) by bench.py's ``cpu_baseline`` leg
as the timed host-CPU baseline ("port")."""
import math

import torch

from . import nets


def clip_grad_norm(grads, max_norm):
    """torch.nn.utils.clip_grad_norm_ restated (Appendix A.16)."""
    total = math.sqrt(sum(float((g.double() ** 2).sum()) for g in grads))
    c = max_norm / (total + 1e-6)
    if c < 1:
        for g in grads:
            g.mul_(c)
    return total


class Adadelta:
    """torch.optim.Adadelta(rho, eps, lr=1) restated (Appendix A.16)."""

    def __init__(self, params, rho=0.95, eps=1e-8, lr=1.0):
        self.params = params
        self.rho, self.eps, self.lr = rho, eps, lr
        self.sq = {k: torch.zeros_like(v) for k, v in params.items()}
        self.acc = {k: torch.zeros_like(v) for k, v in params.items()}

    def step(self, grads):
        with torch.no_grad():
            for k, p in self.params.items():
                g = grads.get(k)
                if g is None:
                    continue
                v, u = self.sq[k], self.acc[k]
                v.mul_(self.rho).addcmul_(g, g, value=1 - self.rho)
                delta = (u + self.eps).sqrt() / (v + self.eps).sqrt() * g
                u.mul_(self.rho).addcmul_(delta, delta, value=1 - self.rho)
                p.sub_(self.lr * delta)


class JointState:
    """Parameters of the four nets as leaf tensors + optimizer state + D's BN buffers."""

    def __init__(self, enh, asr, gan, fbank_W, cfg, dtype=torch.float32):
        """``dtype=torch.float64`` runs the same restatement in double precision: the arbiter for gradient tolerances
        (both fp32 sides -- this oracle and the HIP path -- are compared against it; inputs must be double as well)."""
        self.cfg = cfg
        leaf = lambda d: {k: v.clone().to(dtype).requires_grad_(True) for k, v in d.items()
                          if v.dtype.is_floating_point and 'running_' not in k}
        self.enh = leaf(enh)
        self.asr = leaf({k: v for k, v in asr.items() if not k.startswith('dec.att.')})
        self.gan = leaf(gan)
        self.gan_buf = {k: (v.clone().to(dtype) if v.dtype.is_floating_point else v.clone()) for k, v in gan.items()
                        if 'running_' in k or 'num_batches' in k}
        self.W = fbank_W.clone().to(dtype)
        self.opt_enh = Adadelta(self.enh, eps=cfg['eps'])
        self.opt_asr = Adadelta(self.asr, eps=cfg['eps'])
        self.opt_gan = Adadelta(self.gan, eps=cfg['eps'])


def _zero(d):
    for v in d.values():
        v.grad = None


def _grads(d):
    return {k: v.grad for k, v in d.items() if v.grad is not None}


def joint_step(st, batch, cmvn, update=True):
    """batch = (clean, mix, mix_log, targets, lens, tlens).  Returns dict of scalars/tensors."""
    cfg = st.cfg
    clean, mix, mix_log, targets, lens, tlens = batch
    enhance_out = nets.enhance_forward(st.enh, mix, mix_log, lens, cfg['enhance_layers'])
    enhance_feat = nets.fbank_forward(enhance_out, st.W)
    clean_feat = nets.fbank_forward(clean, st.W)
    lt = cfg.get('enhance_loss_type', 'L2')
    if lt == 'L2':
        el = torch.nn.functional.mse_loss(enhance_feat, clean_feat.detach())
    elif lt == 'L1':
        el = torch.nn.functional.l1_loss(enhance_feat, clean_feat.detach())
    else:
        el = torch.nn.functional.smooth_l1_loss(enhance_feat, clean_feat.detach())
    enhance_loss = cfg['enhance_loss_lambda'] * el
    loss_ctc, loss_att, acc, ctx_c, ctx_m = nets.share_e2e_forward(
        st.asr, clean_feat, enhance_feat, targets, lens, tlens, cfg['elayers'], cmvn, cfg['mtlalpha'])
    coral_loss = cfg['coral_loss_lambda'] * nets.coral(ctx_c, ctx_m)
    asr_loss = cfg['mtlalpha'] * loss_ctc + (1 - cfg['mtlalpha']) * loss_att
    loss = asr_loss + enhance_loss + coral_loss
    out = {}
    if cfg.get('isGAN', True):
        for v in st.gan.values():
            v.requires_grad_(False)
        gan_loss = cfg['gan_loss_lambda'] * nets.gan_loss(
            nets.discriminator_forward(st.gan, st.gan_buf, enhance_feat, cmvn), True)
        loss = loss + gan_loss
        out['gan_loss'] = gan_loss.detach()
    _zero(st.enh)
    _zero(st.asr)
    loss.backward()
    g_enh, g_asr = _grads(st.enh), _grads(st.asr)
    out['g_enh'] = {k: v.clone() for k, v in g_enh.items()}
    out['g_asr'] = {k: v.clone() for k, v in g_asr.items()}
    gn = clip_grad_norm(list(g_asr.values()), cfg['grad_clip'])
    if update and not math.isnan(gn):
        st.opt_enh.step(g_enh)
        st.opt_asr.step(g_asr)
    out.update(loss=loss.detach(), loss_ctc=loss_ctc.detach(), loss_att=loss_att.detach(), acc=acc,
               enhance_loss=enhance_loss.detach(), coral_loss=coral_loss.detach(),
               enhance_out=enhance_out.detach(), enhance_feat=enhance_feat.detach(), grad_norm_asr=gn)
    if cfg.get('isGAN', True):
        for v in st.gan.values():
            v.requires_grad_(True)
        _zero(st.gan)
        l_real = nets.gan_loss(nets.discriminator_forward(st.gan, st.gan_buf, clean_feat.detach(), cmvn), True)
        l_fake = nets.gan_loss(nets.discriminator_forward(st.gan, st.gan_buf, enhance_feat.detach(), cmvn), False)
        loss_D = (l_real + l_fake) * 0.5
        loss_D.backward()
        g_gan = _grads(st.gan)
        out['g_gan'] = {k: v.clone() for k, v in g_gan.items()}
        gnD = clip_grad_norm(list(g_gan.values()), cfg['grad_clip'])
        if update and not math.isnan(gnD):
            st.opt_gan.step(g_gan)
        out.update(loss_D=loss_D.detach(), grad_norm_gan=gnD)
    return out
