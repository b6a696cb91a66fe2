"""Step functions of the reference's other four trainers on the HIP path.

Each class mirrors one loop body of the reference (same order of operations, same error-meter keys) and, like
``joint_train.JointTrainer``, adds the flat fused optimizer with the device-side NaN gate and the RCCL gradient
average (one process per GPU).  ``step(data)`` takes the reference's 10-tuple
``(utt_ids, spk_ids, clean_inputs, clean_log_inputs, mix_inputs, mix_log_inputs, cos_angles, targets, input_sizes,
target_sizes)`` (AsrTrainer: the 6-tuple of data_loader.py) and returns a dict of DEVICE scalars
(``JointTrainer.to_floats`` is the only host synchronisation).

The reference passes EnhanceModel.forward's arguments in an order that does not match its signature
(enhance_base_train.py:87 / enhance_gan_train.py:125 vs enhance_model.py:125); these trainers call it with the
signature's meaning: ``enhance_model(mix_inputs, mix_log_inputs, input_sizes, clean_inputs, cos_angles)``."""
import torch

from . import ops
from .dist import GradSync
from .joint_train import _LOSS_KIND
from .model.e2e_common import set_requires_grad
from .model.gan_model import GANLoss
from .optim import FlatOptimizer


def make_optimizer(opt, module):
    """torch.optim.Adadelta(rho=0.95, eps=opt.eps) / Adam(lr, betas=(beta1, 0.999)) as in every reference trainer."""
    ps = [p for p in module.parameters() if p.requires_grad]
    if opt.opt_type == 'adadelta':
        return FlatOptimizer(ps, 'adadelta', rho=0.95, eps=opt.eps)
    return FlatOptimizer(ps, 'adam', lr=opt.lr, betas=(opt.beta1, 0.999), eps=1e-8)


def _update(optimizer, max_norm):
    """all-reduce (DP) -> clip_grad_norm_ -> NaN-gated step; returns the (device) total norm."""
    GradSync().finish([optimizer])
    gn = optimizer.clip_grad_norm(max_norm)
    optimizer.step()
    return gn


class EnhanceBaseTrainer(object):
    """enhance_base_train.py:85-95: mask-L1 loss of the enhancer against ``clean * cos``."""

    def __init__(self, opt, enhance_model):
        self.opt, self.enhance_model = opt, enhance_model
        self.enhance_optimizer = make_optimizer(opt, enhance_model)

    def step(self, data):
        clean_inputs, mix_inputs, mix_log_inputs, cos_angles, input_sizes = data[2], data[4], data[5], data[6], data[8]
        loss, enhance_out = self.enhance_model(mix_inputs, mix_log_inputs, input_sizes, clean_inputs, cos_angles)
        self.enhance_optimizer.zero_grad()
        loss.backward()
        gn = _update(self.enhance_optimizer, self.opt.grad_clip)
        self.last = dict(enhance_out=enhance_out)
        return {'train/loss': loss.detach(), 'grad_norm': gn}


class EnhanceFbankTrainer(object):
    """enhance_fbank_train.py:108-134: feature-domain loss between fbank(enhanced) and fbank(clean)."""

    def __init__(self, opt, enhance_model, feat_model):
        self.opt, self.enhance_model, self.feat_model = opt, enhance_model, feat_model
        self.enhance_optimizer = make_optimizer(opt, enhance_model)

    def step(self, data):
        clean_inputs, mix_inputs, mix_log_inputs, input_sizes = data[2], data[4], data[5], data[8]
        enhance_out = self.enhance_model(mix_inputs, mix_log_inputs, input_sizes)
        enhance_feat = self.feat_model(enhance_out)
        with torch.no_grad():
            clean_feat = self.feat_model(clean_inputs)
        loss = ops.mean_loss(enhance_feat, clean_feat, 0.0, _LOSS_KIND[self.opt.enhance_loss_type])
        self.enhance_optimizer.zero_grad()
        loss.backward()
        gn = _update(self.enhance_optimizer, self.opt.grad_clip)
        self.last = dict(enhance_out=enhance_out, enhance_feat=enhance_feat)
        return {'train/loss': loss.detach(), 'grad_norm': gn}


class EnhanceGanTrainer(object):
    """enhance_gan_train.py:123-150: mask-L1 + lambda * LSGAN on CMVN-normalised fbank features; then the D-step."""

    def __init__(self, opt, enhance_model, feat_model, gan_model):
        self.opt, self.enhance_model, self.feat_model, self.gan_model = opt, enhance_model, feat_model, gan_model
        self.enhance_optimizer = make_optimizer(opt, enhance_model)
        self.gan_optimizer = make_optimizer(opt, gan_model)
        self.criterionGAN = GANLoss(use_lsgan=not opt.no_lsgan)

    def step(self, data, enhance_cmvn):
        opt = self.opt
        clean_inputs, mix_inputs, mix_log_inputs, cos_angles, input_sizes = data[2], data[4], data[5], data[6], data[8]
        enhance_loss, enhance_out = self.enhance_model(mix_inputs, mix_log_inputs, input_sizes, clean_inputs, cos_angles)
        enhance_feat = self.feat_model(enhance_out, enhance_cmvn)
        with torch.no_grad():
            clean_feat = self.feat_model(clean_inputs, enhance_cmvn)
        set_requires_grad([self.gan_model], False)
        gan_loss = self.criterionGAN(self.gan_model(enhance_feat), True)
        self.enhance_optimizer.zero_grad()
        loss = enhance_loss + opt.gan_loss_lambda * gan_loss
        loss.backward()
        gn = _update(self.enhance_optimizer, opt.grad_clip)
        set_requires_grad([self.gan_model], True)
        self.gan_optimizer.zero_grad()
        loss_D = (self.criterionGAN(self.gan_model(clean_feat.detach()), True)
                  + self.criterionGAN(self.gan_model(enhance_feat.detach()), False)) * 0.5
        loss_D.backward()
        gnD = _update(self.gan_optimizer, opt.grad_clip)
        self.last = dict(enhance_out=enhance_out, enhance_feat=enhance_feat)
        return {'train/loss': loss.detach(), 'train/gan_loss': gan_loss.detach(), 'train/enhance_loss': enhance_loss.detach(),
                'train/loss_D': loss_D.detach(), 'grad_norm': gn, 'grad_norm_D': gnD}


class AsrTrainer(object):
    """asr_train.py:118-131: E2E (CTC + attention) on pre-computed fbank features."""

    def __init__(self, opt, asr_model):
        self.opt, self.asr_model = opt, asr_model
        self.optimizer = make_optimizer(opt, asr_model)
        self.asr_model.dec.return_acc_tensor = True

    def step(self, data, sche_samp_rate=0.0):
        fbank_features, targets, input_sizes, target_sizes = data[2], data[3], data[4], data[5]
        loss_ctc, loss_att, acc = self.asr_model(fbank_features, targets, input_sizes, target_sizes, sche_samp_rate)[:3]
        loss = self.opt.mtlalpha * loss_ctc.view(()) + (1 - self.opt.mtlalpha) * loss_att
        self.optimizer.zero_grad()
        loss.backward()
        gn = _update(self.optimizer, self.opt.grad_clip)
        return {'train/loss': loss.detach(), 'train/loss_ctc': loss_ctc.detach().view(()), 'train/acc': acc, 'train/loss_att': loss_att.detach(),
                'grad_norm': gn}
