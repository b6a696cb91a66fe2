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
import contextlib
import os

import torch

from . import lib, ops
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


class StepStreams(object):
    """The JointTrainer's stream schedule in small, for these trainers (round 6): the step on a HIGH-priority stream of its own, the weight-gradient
    kernels (``ops.param_grads``) on a filler stream -- nothing downstream in a backward pass waits for them, so they fill the CUs the latency-bound
    recurrent chains leave idle -- and, EnhanceGanTrainer, the D-step on a second filler stream under the enhancer's backward chain.  Same kernels,
    same arithmetic, bitwise the single-stream results (tests/test_fullsize_gpu.py: run-to-run equality, parity against the oracle with the schedule
    on; tests/test_trainers_gpu.py::test_n1_trainers_streams_equal_single_stream).  RE2E_NO_OVERLAP=1 (or no GPU): one stream, the reference's order."""

    def __init__(self):
        self.on = torch.cuda.is_available() and os.environ.get('RE2E_NO_OVERLAP', '0') != '1'
        self.main = self.side = self.wgrad = None
        if self.on:
            self.main, self.side, self.wgrad = lib.step_streams()      # the process's ONE set of step streams (shared with JointTrainer)

    @contextlib.contextmanager
    def step(self):
        """``with streams.step() as st:`` -- st is None on the single-stream path; otherwise the body runs on the step's own stream with the
        weight-gradient stream armed, and has to ``st.join()`` before its optimizer update."""
        if not self.on:
            yield None
            return
        caller = torch.cuda.current_stream()
        self.main.wait_stream(caller)
        ops.MULTI_STREAM, ops.WGRAD_STREAM, ops.AUX_STREAM = True, self.wgrad, self.side        # (aux: the CTC head beside the decoder, E2E.forward)
        try:
            with torch.cuda.stream(self.main):
                yield self
                torch.cuda.current_stream().wait_stream(self.wgrad)        # (whatever the body left there or on the side stream)
                torch.cuda.current_stream().wait_stream(self.side)
        finally:
            ops.MULTI_STREAM, ops.WGRAD_STREAM, ops.AUX_STREAM = False, None, None
            caller.wait_stream(self.main)

    def join(self):
        """The current stream waits for the weight-gradient kernels enqueued so far: in front of the clip / update that reads them."""
        torch.cuda.current_stream().wait_stream(self.wgrad)


class EnhanceBaseTrainer(object):
    """enhance_base_train.py:85-95: mask-L1 loss of the enhancer against ``clean * cos``."""

    def __init__(self, opt, enhance_model):
        self.opt, self.enhance_model = opt, enhance_model
        self.enhance_optimizer = make_optimizer(opt, enhance_model)
        self.streams = StepStreams()

    def step(self, data):
        clean_inputs, mix_inputs, mix_log_inputs, cos_angles, input_sizes = data[2], data[4], data[5], data[6], data[8]
        with self.streams.step() as st:
            loss, enhance_out = self.enhance_model(mix_inputs, mix_log_inputs, input_sizes, clean_inputs, cos_angles)
            self.enhance_optimizer.zero_grad()
            loss.backward()
            if st is not None:
                st.join()
            gn = _update(self.enhance_optimizer, self.opt.grad_clip)
        self.last = dict(enhance_out=enhance_out)
        return {'train/loss': loss.detach(), 'grad_norm': gn}


class EnhanceFbankTrainer(object):
    """enhance_fbank_train.py:108-134: feature-domain loss between fbank(enhanced) and fbank(clean)."""

    def __init__(self, opt, enhance_model, feat_model):
        self.opt, self.enhance_model, self.feat_model = opt, enhance_model, feat_model
        self.enhance_optimizer = make_optimizer(opt, enhance_model)
        self.streams = StepStreams()

    def step(self, data):
        clean_inputs, mix_inputs, mix_log_inputs, input_sizes = data[2], data[4], data[5], data[8]
        with self.streams.step() as st:
            enhance_out = self.enhance_model(mix_inputs, mix_log_inputs, input_sizes)
            enhance_feat = self.feat_model(enhance_out)
            with torch.no_grad():
                clean_feat = self.feat_model(clean_inputs)
            loss = ops.mean_loss(enhance_feat, clean_feat, 0.0, _LOSS_KIND[self.opt.enhance_loss_type])
            self.enhance_optimizer.zero_grad()
            loss.backward()
            if st is not None:
                st.join()
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
        self.streams = StepStreams()

    def step(self, data, enhance_cmvn):
        with self.streams.step() as st:
            return self._step_streams(data, enhance_cmvn, st) if st is not None else self._step_plain(data, enhance_cmvn)

    def _step_plain(self, data, enhance_cmvn):
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

    def _step_streams(self, data, enhance_cmvn, st):
        """The same step in two backward phases: (1) the G-step's gradient through the frozen discriminator and the fbank down to ``enhance_out``,
        on the step's stream; then the WHOLE D-step (enhance_gan_train.py:141-150: its two forwards, backward, clip, update -- D's weights are not
        touched by the G-step, and its three forwards keep the reference's order: fake, real, fake) is enqueued on the side stream; (2) the
        enhancer's backward -- 1600 dependent recurrence steps that leave most of the chip idle -- runs beside it."""
        opt = self.opt
        clean_inputs, mix_inputs, mix_log_inputs, cos_angles, input_sizes = data[2], data[4], data[5], data[6], data[8]
        main, side = torch.cuda.current_stream(), st.side
        enhance_loss, enhance_out = self.enhance_model(mix_inputs, mix_log_inputs, input_sizes, clean_inputs, cos_angles)
        enhance_feat = self.feat_model(enhance_out, enhance_cmvn)
        with torch.no_grad():
            clean_feat = self.feat_model(clean_inputs, enhance_cmvn)
        set_requires_grad([self.gan_model], False)
        gan_loss = self.criterionGAN(self.gan_model(enhance_feat), True)
        self.enhance_optimizer.zero_grad()
        loss = enhance_loss + opt.gan_loss_lambda * gan_loss
        (g_eo,) = torch.autograd.grad(loss, [enhance_out])                   # phase 1: D (input gradient only), fbank, the mask loss
        set_requires_grad([self.gan_model], True)
        ev = torch.cuda.Event()
        ev.record(main)
        side.wait_event(ev)
        with torch.cuda.stream(side):
            fake, real = enhance_feat.detach(), clean_feat.detach()
            fake.record_stream(side)
            real.record_stream(side)
            self.gan_optimizer.zero_grad()
            loss_D = (self.criterionGAN(self.gan_model(real), True) + self.criterionGAN(self.gan_model(fake), False)) * 0.5
            loss_D.backward()
            st.join()                                                        # D's weight gradients (the only ones enqueued there so far)
            gnD = _update(self.gan_optimizer, opt.grad_clip)
            for t_ in (loss_D, gnD):
                t_.record_stream(main)
        enhance_out.backward(g_eo)                                           # phase 2: the enhancer's backward chain
        st.join()
        gn = _update(self.enhance_optimizer, opt.grad_clip)
        main.wait_stream(side)
        self.last = dict(enhance_out=enhance_out, enhance_feat=enhance_feat)
        return {'train/loss': loss.detach(), 'train/gan_loss': gan_loss.detach(), 'train/enhance_loss': enhance_loss.detach(),
                'train/loss_D': loss_D.detach(), 'grad_norm': gn, 'grad_norm_D': gnD}


class AsrTrainer(object):
    """asr_train.py:118-131: E2E (CTC + attention) on pre-computed fbank features."""

    def __init__(self, opt, asr_model):
        self.opt, self.asr_model = opt, asr_model
        self.optimizer = make_optimizer(opt, asr_model)
        self.asr_model.dec.return_acc_tensor = True
        self.streams = StepStreams()

    def step(self, data, sche_samp_rate=0.0):
        fbank_features, targets, input_sizes, target_sizes = data[2], data[3], data[4], data[5]
        with self.streams.step() as st:
            loss_ctc, loss_att, acc = self.asr_model(fbank_features, targets, input_sizes, target_sizes, sche_samp_rate)[:3]
            loss = self.opt.mtlalpha * loss_ctc.view(()) + (1 - self.opt.mtlalpha) * loss_att
            self.optimizer.zero_grad()
            loss.backward()
            if st is not None:
                st.join()
            gn = _update(self.optimizer, self.opt.grad_clip)
        return {'train/loss': loss.detach(), 'train/loss_ctc': loss_ctc.detach().view(()), 'train/acc': acc, 'train/loss_att': loss_att.detach(),
                'grad_norm': gn}
