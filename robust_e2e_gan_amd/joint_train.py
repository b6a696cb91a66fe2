"""joint_train loop API (reference joint_train.py:32-50 compute_cmvn_epoch, :156-214 training step).

``JointTrainer.step(data)`` is the hot path BASELINE.json names: enhancer -> fbank -> shared E2E
(+CTC, +location-attention decoder) + discriminator, G-step then D-step, with the reference's
order of operations (Appendix A.12-14): freeze toggling, clip on the ASR net only, NaN guard
gating both G optimizers, D clipped separately.  Every rank runs it on its own utterance shard;
gradients are averaged by dist.GradSync (RCCL) before clipping."""
import copy
import logging
import math
import os
import time

import numpy as np
import torch

from . import lib, ops
from . import dist as rdist
from .dist import GradSync
from .model.e2e_common import set_requires_grad
from .model.gan_model import CORAL, GANLoss, replay_running_stats
from .optim import FlatOptimizer

_LOSS_KIND = {'L2': lib.LOSS_L2, 'L1': lib.LOSS_L1, 'smooth_L1': lib.LOSS_SMOOTH_L1}


class stepwise_kernels(object):
    """``with stepwise_kernels():`` the launch-per-step recurrences and the launch-per-token decoder loop instead of the persistent kernels --
    same arithmetic, no in-launch hand-off that can time out (what a step / pass is repeated with after a persistent kernel gave up)."""
    _KEYS = ('RE2E_LSTM_PERSIST', 'RE2E_LSTM_PERSIST_BWD')

    def __enter__(self):
        self.saved = {k: os.environ.get(k) for k in self._KEYS}
        for k in self._KEYS:
            os.environ[k] = '0'
        self.dec, ops.DECODER_PERSIST = ops.DECODER_PERSIST, False        # (csrc/decloop.hip counts its give-ups with the recurrences')
        return self

    def __exit__(self, *exc):
        ops.DECODER_PERSIST = self.dec
        for k, v in self.saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
        return False


def compute_cmvn_epoch(opt, train_loader, enhance_model, feat_model):
    """joint_train.py:32-50 (quirk kept: the accumulators are never reset between calls)."""
    enhance_model.eval()
    feat_model.eval()
    enhance_cmvn = None
    with torch.no_grad():
        for data in train_loader:
            mix_inputs, mix_log_inputs, input_sizes = data[4], data[5], data[8]
            enhance_out = enhance_model(mix_inputs, mix_log_inputs, input_sizes)
            enhance_cmvn = feat_model.compute_cmvn(enhance_out, input_sizes)
            if enhance_cmvn is not None:
                if getattr(opt, 'exp_path', None) and rdist.rank() == 0:       # one writer per experiment directory
                    os.makedirs(opt.exp_path, exist_ok=True)
                    np.save(os.path.join(opt.exp_path, 'enhance_cmvn.npy'), enhance_cmvn)
                break
    enhance_model.train()
    feat_model.train()
    if rdist.any_rank(1 if enhance_cmvn is None else 0):        # every replica raises together (the broadcast below is a collective)
        raise RuntimeError('train_loader exhausted before cmvn_num utterances were accumulated'
                           + ('' if enhance_cmvn is None else ' (on another rank)'))
    # data parallel: every rank estimated the statistics on its own shard; all replicas use rank 0's (= the saved file)
    return rdist.broadcast_(torch.FloatTensor(enhance_cmvn), 0)


def build_optimizers(opt, enhance_model, asr_model, gan_model=None):
    """joint_train.py:127-140 on flat buffers (fbank params are in no optimizer, Appendix A.16)."""
    def mk(m):
        ps = [p for p in m.parameters() if p.requires_grad]
        if opt.opt_type == 'adadelta':
            return FlatOptimizer(ps, 'adadelta', rho=0.95, eps=opt.eps)
        return FlatOptimizer(ps, 'adam', lr=opt.lr, betas=(opt.beta1, 0.999), eps=1e-8)
    return mk(enhance_model), mk(asr_model), (mk(gan_model) if gan_model is not None else None)


class JointTrainer(object):
    def __init__(self, opt, enhance_model, feat_model, asr_model, gan_model=None):
        self.opt = opt
        self.enhance_model, self.feat_model, self.asr_model, self.gan_model = enhance_model, feat_model, asr_model, gan_model
        self.isGAN = bool(getattr(opt, 'isGAN', False)) and gan_model is not None
        if self.isGAN and getattr(opt, 'netD_type', 'basic') == 'pixel':
            raise lib.Re2eError('the joint loop cannot drive netD_type pixel: upstream reads an undefined mix_feat for it '
                                '(joint_train.py:178) and applies the 80-wide CMVN to the 160-wide concatenation')
        self.enhance_optimizer, self.asr_optimizer, self.gan_optimizer = build_optimizers(opt, enhance_model, asr_model,
                                                                                          gan_model if self.isGAN else None)
        self.criterionGAN = GANLoss(use_lsgan=not opt.no_lsgan) if self.isGAN else None
        self.asr_model.dec.return_acc_tensor = True
        # Give-up protocol of the persistent kernels (csrc/lstm.hip re2e_step_gate; ``fit`` / ``recover_aborted_step``): device scalars,
        # allocated at the first step -- the count of give-ups this trainer has acknowledged, the factor (1.0 or NaN) the losses of a step
        # are multiplied by (two of them, alternating: step k reads one while its gate writes the other for step k + 1), the step's delta
        self._gate = None
        self._step_no = 0
        # Dropout masks (ops.dropout: counter-based Philox, nothing stored): one stream per process, keyed by (opt.seed, rank) so that
        # data-parallel replicas draw DIFFERENT masks, as upstream's per-process RNG would; the (seed, next mask index) pair is part
        # of ``state()`` and ``restore_dropout`` puts it back, so a resumed run continues the mask sequence instead of replaying
        # it from index 0.  NB validation consumes mask indices too: upstream's CTC head calls F.dropout with training=True in
        # eval mode as well (e2e_ctc.py:51), so the training masks after a validation pass depend on validate_freq -- upstream's
        # global RNG has the same property.
        if any(float(getattr(opt, k, 0.0) or 0.0) > 0.0 for k in ('dropout_rate', 'enhance_dropout_rate')):
            ops.dropout_seed(self._dropout_stream_seed(int(getattr(opt, 'seed', 1234))))
        # run the D passes on a side HIP stream under the latency-bound recurrent chains (RE2E_NO_OVERLAP=1: profiling)
        self.overlap_dstep = os.environ.get('RE2E_NO_OVERLAP', '0') != '1'
        self.marks = [] if os.environ.get('RE2E_TIMELINE') else None
        # the D-step's D(fake) forward equals the G-step's (same input, same weights): keep that graph and walk it twice
        self.early_dreal = lib.exp_env('RE2E_NO_EARLY_DREAL', '0') != '1'     # D-step real half under the enhancer's forward chain
        self.reuse_dfake = lib.exp_env('RE2E_NO_DFAKE_REUSE', '0') != '1' 
        self.side_stream = self.wgrad_stream = None
        if torch.cuda.is_available():
            # filler streams: optionally restricted to a subset of the CUs (RE2E_FILLER_CUS, default all) so that the
            # chains on the main stream always find idle CUs
            # MI355X sweeps (ms/step).  With launch-per-step recurrences: 128: 112.5, 160: 103.1, 192: 99.6, 224: 98.0, 256: 99.5.
            # With the persistent recurrences: 192: 90.6, 208: 91.2, 224: 89.0, 240: 90.1, 256 (no mask): 88.2 -- resident
            # chains no longer need CUs kept free for their launches, so the fillers get the whole chip.
            ncu = int(lib.exp_env('RE2E_FILLER_CUS', '256'))
            dev = next(enhance_model.parameters()).device
            if 0 < ncu < 256:
                self.side_stream = lib.cu_masked_stream(ncu, 256, dev)
                self.wgrad_stream = lib.cu_masked_stream(ncu, 256, dev)
            shared = lib.step_streams(dev, int(lib.exp_env('RE2E_MAIN_PRIORITY', '-1')))       # ONE set per process: see lib.step_streams
            if self.side_stream is None:
                self.side_stream, self.wgrad_stream = shared[1], shared[2]
            else:
                # both run beside the resident recurrences of the main stream: 4-wave engine tiles there (re2e_stream_role)
                lib.set_stream_role(self.side_stream, True)
                lib.set_stream_role(self.wgrad_stream, True)
            # NB: no further streams.  A process gets 4 hardware queues by default; a fifth stream (default + main + side
            # + wgrad + one more) is multiplexed onto an occupied queue and serialises against it (measured with a
            # dedicated D-step stream: 91 -> 155 ms/step).  Measured and rejected as well: running the D-step's
            # forward/backward early, under the ASR forward (+2.3 ms/step: it slows the encoder chain more than it
            # relieves the backward) -- and again with the persistent recurrences, only the FAKE half, started exactly when the
            # encoder's chain starts: ASR forward +3.9 ms, enhancer backward -2.7 ms, step 88.9 -> 91.0 ms; holding the G-step's
            # backward through D until the decoder's backward chain is done (decoder backward -1.9 ms, BLSTMP backward +1.6 ms:
            # within the noise of the step); the FAKE half of the D-step between two backward calls cut at the encoder output,
            # i.e. under the BLSTMP's backward recurrences (75.25 -> 75.6 ms); a dedicated stream for the recurrent sequences masked to the 32 CUs the fillers leave
            # alone (GPU_MAX_HW_QUEUES=8): the chains are no faster there (16.9 vs 18 ms for the enhancer forward under the
            # D(real) filler -- the slowdown under load is not CU sharing) and 256-workgroup sequences do not fit 32 CUs.
        self.main_stream = None
        if torch.cuda.is_available():
            # The step runs on its OWN stream, never on the legacy default stream (which synchronises implicitly with
            # the blocking CU-masked filler streams: 91 -> 140 ms).  It is a high-priority stream (RE2E_MAIN_PRIORITY,
            # default -1; the device's range is (0, -1): there is no lower priority to give the fillers): the launch-per-step chains gain from it (91.4 vs 92.4 ms).  A fifth, normal-priority stream for
            # the persistent sequences (GPU_MAX_HW_QUEUES=8) changed nothing: what slows a resident chain beside the
            # fillers is two of its workgroups sharing a CU (tools/bench_fill_under_chain.py), not the queue it came from.
            try:
                self.main_stream = lib.step_streams(next(enhance_model.parameters()).device, int(lib.exp_env('RE2E_MAIN_PRIORITY', '-1')))[0]
            except Exception:
                self.main_stream = None

    def step(self, data, sche_samp_rate, enhance_cmvn):
        """One training iteration (joint_train.py:157-213).  Returns a dict of DEVICE scalars (call
        ``to_floats`` to log them: that is the only host synchronisation).

        With overlap enabled the critical path (recurrent chains, decoder) runs on a HIGH-priority stream and the
        filler work (clean-branch convs, discriminator passes, weight gradients) on normal-priority streams, so
        that the short dependent launches of the chains are dispatched ahead of queued bulk kernels."""
        try:
            if not (self.overlap_dstep and self.main_stream is not None):
                return self._step(data, sche_samp_rate, enhance_cmvn)
            caller = torch.cuda.current_stream()
            self.main_stream.wait_stream(caller)
            with torch.cuda.stream(self.main_stream):
                out = self._step(data, sche_samp_rate, enhance_cmvn)
            caller.wait_stream(self.main_stream)
            return out
        finally:
            # the stream routing is this step's: ops used outside it (validation, other trainers, tests) stay on ONE stream
            ops.MULTI_STREAM, ops.WGRAD_STREAM, ops.AUX_STREAM, ops.MARKS = False, None, None, None
            ops.SYNC_BN = False

    def _mark(self, label):
        """RE2E_TIMELINE=1: remember (label, host time, event on the current stream) -- ``timeline()`` prints how far the
        GPU runs behind the host at each phase boundary (host-bound phases show a lag near zero)."""
        if self.marks is not None:
            ev = torch.cuda.Event(enable_timing=True)
            ev.record()
            self.marks.append((label, time.perf_counter(), ev))

    def timeline(self):
        torch.cuda.synchronize()
        (l0, h0, e0), rows = self.marks[0], []
        for label, h, e in self.marks:
            rows.append((label, (h - h0) * 1e3, e0.elapsed_time(e)))
        self.marks = []
        return rows

    def _step(self, data, sche_samp_rate, enhance_cmvn):
        opt = self.opt
        self._mark('start')
        ops.MARKS = self.marks
        clean_inputs, mix_inputs, mix_log_inputs, targets, input_sizes, target_sizes = data[2], data[4], data[5], data[7], data[8], data[9]
        overlap = self.overlap_dstep
        ops.MULTI_STREAM = bool(overlap)
        ops.SYNC_BN = bool(getattr(opt, 'sync_bn', False)) and rdist.world_size() > 1     # ONE global batch sharded over the ranks
        ops.WGRAD_STREAM = self.wgrad_stream if overlap else None
        ops.AUX_STREAM = self.side_stream if (overlap and lib.exp_env('RE2E_CTC_MAIN') != '1') else None
        main = torch.cuda.current_stream()
        hold = self._hold_factor()
        clean_branch, d_real_part = None, None
        if overlap and getattr(self.asr_model, 'etype', '').startswith('vgg'):
            # the clean branch (fbank -> CMVN -> VGG conv stack) does not depend on the enhancer: enqueue it on the side
            # stream first so that it fills the CUs the enhancer's 1600-launch recurrent chain leaves idle
            side = self.side_stream
            side.wait_stream(main)
            with torch.cuda.stream(side):
                with torch.no_grad():
                    clean_feat = self.feat_model(clean_inputs)
                ev_cf = torch.cuda.Event()
                ev_cf.record()
                clean_branch = self.asr_model.encode_clean(clean_feat, enhance_cmvn, input_sizes)
            self._mark('clean branch enqueued (side)')
            enhance_out = self.enhance_model(mix_inputs, mix_log_inputs, input_sizes)
            self._mark('enhancer fwd')
            if self.isGAN and self.reuse_dfake and self.early_dreal:
                # D-step, real half (joint_train.py:198-201): needs only clean_feat and D's current weights, so it goes
                # under the enhancer's forward chain instead of under the (already saturated) backward chain.  Enqueued
                # AFTER the enhancer: its ~2.5 ms of host launches must not delay the start of that chain.
                with torch.cuda.stream(side):
                    d_real_part = self._d_real(clean_feat, enhance_cmvn, hold)
            ops.mark_grad(enhance_out, 'enhance_out (fbank bwd done)')
            enhance_feat = self.feat_model(enhance_out)
            ops.mark_grad(enhance_feat, 'enhance_feat (VGG, D, L1 bwd done)')
            main.wait_event(ev_cf)
            clean_feat.record_stream(main)
        else:
            enhance_out = self.enhance_model(mix_inputs, mix_log_inputs, input_sizes)
            enhance_feat = self.feat_model(enhance_out)
            with torch.no_grad():
                clean_feat = self.feat_model(clean_inputs)
        enhance_loss = opt.enhance_loss_lambda * ops.mean_loss(enhance_feat, clean_feat, 0.0, _LOSS_KIND[opt.enhance_loss_type])
        out = {}
        gan_loss = None
        overlap = self.isGAN and self.overlap_dstep
        if self.isGAN:
            # G-step discriminator pass (joint_train.py:175-182).  It depends only on enhance_feat, so with
            # overlap it is enqueued on the side stream BEFORE the ASR forward and runs under it; autograd runs
            # its backward on the same side stream.
            reuse = self.reuse_dfake
            # upstream freezes D here (:176); with ``reuse`` the graph is built with trainable parameters instead and the
            # G-step backward is told not to produce their gradients (ops.FROZEN_PARAMS below) -- same arithmetic
            set_requires_grad([self.gan_model], reuse)
            fake_stats = [] if reuse else None

            def d_fake_forward():
                ops.BN_STATS_SINK = fake_stats
                try:
                    d = self.gan_model(enhance_feat, enhance_cmvn)
                finally:
                    ops.BN_STATS_SINK = None
                return d, opt.gan_loss_lambda * self.criterionGAN(d, True)
            if overlap:
                self.side_stream.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(self.side_stream):
                    enhance_feat.record_stream(self.side_stream)
                    d_fake, gan_loss = d_fake_forward()
                    ops.mark_grad(d_fake, 'd_fake (side: G-step D bwd starts)')
            else:
                d_fake, gan_loss = d_fake_forward()
            fake_bn_layers = list(self.gan_model._bn_layers_last)
        self._mark('fbank + G-step D fwd enqueued (side)')
        loss_ctc, loss_att, acc, clean_context, mix_context = self.asr_model(
            clean_feat, enhance_feat, targets, input_sizes, target_sizes, sche_samp_rate, enhance_cmvn, clean_branch=clean_branch,
            context_loss=lambda cc, mc: opt.coral_loss_lambda * CORAL(cc, mc))      # on the filler stream, beside the decoder
        self._mark('ASR fwd')
        coral_loss = self.asr_model.last_context_loss
        asr_loss = opt.mtlalpha * loss_ctc.view(()) + (1 - opt.mtlalpha) * loss_att
        loss = asr_loss + enhance_loss + coral_loss
        if self.isGAN:
            if overlap:
                torch.cuda.current_stream().wait_stream(self.side_stream)
            loss = loss + gan_loss
            out['train/gan_loss'] = opt.gan_loss_lambda * gan_loss.detach()
        out['train/loss'] = loss.detach()
        # x 1.0 (exact) -- or x NaN while a step that a persistent kernel gave up on has not been repeated yet: this step was enqueued before
        # the host could know, it must apply nothing (on any replica: the NaN travels through the gradient average), see ``fit``
        loss = loss * hold
        self.enhance_optimizer.zero_grad()
        self.asr_optimizer.zero_grad()
        sync = GradSync()
        armed = clean_branch is None       # with the clean branch on the side stream the ASR gradients are complete only
        if armed:                          # after that stream has been joined, so the early all-reduce hook is not used
            # the ASR weight-gradient kernels run on the wgrad stream (and the CTC branch on the aux stream) when the step is
            # multi-stream: the hook issues the collective from the wgrad stream, behind events of the other two
            sync.arm(enhance_feat, self.asr_optimizer, issue_stream=ops.WGRAD_STREAM, also_wait=[ops.AUX_STREAM])
        if self.isGAN and self.overlap_dstep:
            # Phase 1: backward of everything downstream of the enhancer (ASR, D, fbank) on the main stream.
            main = torch.cuda.current_stream()
            ev_fwd = torch.cuda.Event()
            ev_fwd.record(main)
            # The ASR parameters are listed as inputs so that autograd also runs the nodes that lead ONLY to them (the
            # clean branch's conv stack reaches the loss through CORAL but not enhance_out and would be pruned);
            # the fused ops accumulate parameter gradients themselves and hand None back for them.
            asr_params = [p for p in self.asr_model.parameters() if p.requires_grad]
            reuse = self.isGAN and self.reuse_dfake
            ops.FROZEN_PARAMS = frozenset(id(p) for p in self.gan_model.parameters()) if reuse else frozenset()
            cut = getattr(self.asr_model, 'clean_cut', None) if clean_branch is not None else None
            self.asr_model.clean_cut = None
            ev_cut, cut_fired = torch.cuda.Event(), []
            if cut is not None:
                def on_cut_grad(g):            # BLSTMP backward enqueued on main: d(loss)/d(leaf) is on its way
                    ev_cut.record(main)
                    cut_fired.append(True)
                cut[1].register_hook(on_cut_grad)
            try:
                gs = torch.autograd.grad(loss, [enhance_out] + ([cut[1]] if cut else []) + asr_params, allow_unused=True, retain_graph=reuse)
            finally:
                ops.FROZEN_PARAMS = frozenset()
            g_eo = gs[0]
            self._mark('bwd phase 1 (ASR, D, fbank)')
            ev_bwd1 = torch.cuda.Event()
            ev_bwd1.record(main)
            phase2_first = lib.exp_env('RE2E_PHASE2_FIRST', '0') == '1' and rdist.world_size() == 1     # experiment: enqueue order only
            if phase2_first:
                enhance_out.backward(g_eo)
                self._mark('bwd phase 2 (enhancer)')
            if cut is not None and gs[1] is not None:
                # the clean branch's conv-stack backward (reached through CORAL and the shared BLSTMP): side stream, not
                # joined into main before the enhancer's backward chain starts -- it runs under that chain
                side = self.side_stream
                side.wait_event(ev_cut if cut_fired else ev_bwd1)
                with torch.cuda.stream(side):
                    gs[1].record_stream(side)
                    # (its weight gradients stay on the weight-gradient stream: inline here, like the D-step's, 46.0 -> 46.7 ms per step,
                    #  profiles/r06_ab_dstep_inline_wgrad.txt)
                    torch.autograd.backward([cut[0]], [gs[1]])
            ev_side_bwd = torch.cuda.Event()          # clean-branch conv backward (ASR gradients) enqueued on the side stream
            ev_side_bwd.record(self.side_stream)
            if self.marks is not None:
                for st_, nm_ in ((self.side_stream, 'side'), (self.wgrad_stream, 'wgrad')):
                    with torch.cuda.stream(st_):
                        self._mark('  %s stream: phase-1 work done' % nm_)
            # D-step (joint_train.py:195-212) on a side stream: it only needs the forward results, so it fills
            # the CUs that the latency-bound enhancer BLSTM backward (1600 dependent launches) leaves idle.
            side = self.side_stream
            side.wait_event(ev_fwd)
            with torch.cuda.stream(side):
                for t_ in (enhance_feat, clean_feat, enhance_cmvn):
                    if isinstance(t_, torch.Tensor) and t_.is_cuda:
                        t_.record_stream(side)
                loss_D = self._d_step(clean_feat, enhance_feat, enhance_cmvn, d_fake=d_fake if reuse else None, fake_stats=fake_stats,
                                      fake_bn=fake_bn_layers, real_part=d_real_part, hold=hold)
            self._mark('D-step enqueued (side)')
            if self.marks is not None:
                with torch.cuda.stream(side):
                    self._mark('  side stream: D-step done')
            if not armed and rdist.world_size() > 1:
                # Data parallel: every ASR gradient kernel has been enqueued (phase 1 on main, the clean-branch / CTC
                # backward on side, the weight gradients on wgrad), so the 116 MB ASR all-reduce starts as soon as those
                # finish and runs over xGMI UNDER the enhancer's backward chain instead of after it.  RCCL orders its
                # stream behind the stream the collective is issued from: issue it from the wgrad stream, behind events
                # of the other two.
                ws = self.wgrad_stream
                with torch.cuda.stream(ws):
                    ws.wait_event(ev_bwd1)
                    ws.wait_event(ev_side_bwd)
                    work = rdist.allreduce_mean_(self.asr_optimizer.grad, async_op=True)
                if work is not None:
                    sync.pending.append(work)
                armed = True
            # Phase 2: the enhancer backward chain on the main stream.
            if not phase2_first:
                enhance_out.backward(g_eo)
                self._mark('bwd phase 2 (enhancer)')
            if self.marks is not None:
                with torch.cuda.stream(self.wgrad_stream):
                    self._mark('  wgrad stream: all weight gradients done')
            torch.cuda.current_stream().wait_event(ev_side_bwd)
        else:
            reuse = self.isGAN and self.reuse_dfake
            ops.FROZEN_PARAMS = frozenset(id(p) for p in self.gan_model.parameters()) if reuse else frozenset()
            # ShareE2E cuts the graph at the clean conv stack's output whenever that stack ran on the side stream
            # (clean_branch is not None), GAN or not: d(loss)/d(leaf) lands in the leaf's .grad and the conv-stack
            # backward (reached through CORAL and the shared BLSTMP) is run here, on the side stream.
            cut = getattr(self.asr_model, 'clean_cut', None) if clean_branch is not None else None
            self.asr_model.clean_cut = None
            try:
                loss.backward(retain_graph=reuse)
            finally:
                ops.FROZEN_PARAMS = frozenset()
            if cut is not None and cut[1].grad is not None:
                side = self.side_stream
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):
                    cut[1].grad.record_stream(side)
                    torch.autograd.backward([cut[0]], [cut[1].grad])
                cut[1].grad = None
            loss_D = None
            if self.overlap_dstep:
                torch.cuda.current_stream().wait_stream(self.side_stream)
        if self.overlap_dstep:
            torch.cuda.current_stream().wait_stream(self.wgrad_stream)      # deferred weight-gradient kernels
        sync.finish([self.enhance_optimizer] if armed else [self.asr_optimizer, self.enhance_optimizer])
        grad_norm = self.asr_optimizer.clip_grad_norm(opt.grad_clip)           # ASR params only (:188)
        # The NaN gate (:189-193), on the device.  Besides the ASR norm it refuses the update when the ENHANCER's gradients are not finite (a
        # give-up in its backward chain leaves the ASR norm finite) or when a persistent kernel gave up anywhere in this step.
        base, holds, delta = self._gate
        dp = rdist.world_size() > 1
        call_gate = lambda *a: lib.call('re2e_step_gate', base.data_ptr(), 0, *a)
        call_gate(self.enhance_optimizer.grad_sumsq().data_ptr(), self.asr_optimizer.stats.data_ptr(), None, None, None, None)
        self.enhance_optimizer.step(self.asr_optimizer.gate_stats())           # unclipped, same gate
        self.asr_optimizer.step()
        nxt = holds[(self._step_no + 1) % 2]
        if self.isGAN:
            if loss_D is None:
                loss_D = self._d_step(clean_feat, enhance_feat, enhance_cmvn, d_fake=d_fake if self.reuse_dfake else None,
                                      fake_stats=fake_stats, fake_bn=fake_bn_layers, hold=hold)
            else:
                torch.cuda.current_stream().wait_stream(self.side_stream)
            # D's update (:211) is the step's LAST kernel, behind the gate: a step that a persistent kernel gave up on is repeated as a
            # whole, so D must not have moved in it.  A NaN of the model's own does not stop D (upstream's D-step is independent of the G-step's
            # gate) -- except in data-parallel runs, where the give-up count is a per-device number and the one thing every replica agrees
            # on is the main gate's flag.
            call_gate(None, None, self.gan_optimizer.stats.data_ptr(), self.asr_optimizer.stats.data_ptr() + 8 if dp else None, nxt.data_ptr(),
                      delta.data_ptr())
            self.gan_optimizer.step()
            out['train/loss_D'] = loss_D.detach()
        else:
            call_gate(None, None, None, None, nxt.data_ptr(), delta.data_ptr())
        self._step_no += 1
        out.update({'train/loss_ctc': loss_ctc.detach().view(()), 'train/acc': acc, 'train/loss_att': loss_att.detach(),
                    'train/enhance_loss': enhance_loss.detach(), 'train/coral_loss': coral_loss.detach(),
                    'grad_norm': grad_norm.clone(),      # a copy: the optimizer's stats buffer is rewritten by the next step (NaN when the gate refused)
                    'aborts': delta.clone()})            # give-ups of persistent kernels since the last acknowledgement, counted at the end of this step
        self.last = dict(enhance_out=enhance_out, enhance_feat=enhance_feat)
        self._mark('optimizers')
        return out

    def _hold_factor(self):
        """The device scalar this step's losses are multiplied by (see ``_step``); first call: allocate the gate's state and acknowledge
        whatever earlier users of the process left in the give-up counters."""
        if self._gate is None:
            dev = next(self.enhance_model.parameters()).device
            base = torch.zeros(1, dtype=torch.int32, device=dev)
            holds = [torch.ones(1, dtype=torch.float32, device=dev) for _ in range(2)]
            delta = torch.zeros(1, dtype=torch.float32, device=dev)
            lib.call('re2e_step_gate', base.data_ptr(), 1, None, None, None, None, None, None)
            self._gate = (base, holds, delta)
        return self._gate[1][self._step_no % 2]

    def _acknowledge_aborts(self):
        """The host has seen the give-ups counted so far (it is about to repeat the step they spoilt): the next step's losses are whole again."""
        base, holds, delta = self._gate
        lib.call('re2e_step_gate', base.data_ptr(), 1, None, None, None, None, holds[self._step_no % 2].data_ptr(), delta.data_ptr())

    def _giveups_since_ack(self):
        """Give-ups of persistent kernels on THIS device since the last acknowledgement, read through the gate (no acknowledgement, nothing
        refused): for work that has no step gate of its own -- the validation pass, the CMVN estimate.  Synchronises with the device."""
        self._hold_factor()
        base = self._gate[0]
        out = torch.zeros(1, dtype=torch.float32, device=base.device)
        lib.call('re2e_step_gate', base.data_ptr(), 0, None, None, None, None, None, out.data_ptr())
        return int(out.item())

    guarded_repeats = 0

    def run_guarded(self, fn, snapshot, restore, what):
        """Non-training work runs the same persistent recurrences / decoder loop as a step but behind no step gate: ``fn()`` is run, the
        replicas agree (collective) whether a persistent kernel gave up in it on ANY of them, and if so every replica puts ``snapshot()``'s state
        back (``restore``), acknowledges and runs ``fn`` again with the launch-per-step kernels -- before the next training step is enqueued, whose
        gate would otherwise refuse its update on the one replica that counted the give-up and let the others apply it.  Raises on every
        replica if the repeated pass gives up as well."""
        self._hold_factor()          # (the gate's first use acknowledges what earlier users of the process left: before fn, not after)
        snap = snapshot()
        res = fn()
        if rdist.any_rank(self._giveups_since_ack()) == 0:
            return res
        logging.warning('a persistent kernel gave up on a peer workgroup during %s: repeated with the launch-per-step kernels', what)
        self.guarded_repeats += 1
        restore(snap)
        self._acknowledge_aborts()
        with stepwise_kernels():
            res = fn()
        if rdist.any_rank(self._giveups_since_ack()):
            raise lib.Re2eError('%s gave up with the launch-per-step kernels as well' % what)
        return res

    def _d_real(self, clean_feat, enhance_cmvn, hold=None):
        """Real half of the discriminator update on the CURRENT stream, ahead of the G-step: forward of D(clean) with the
        BatchNorm running-statistics update deferred (upstream applies it AFTER the G-step's D(fake) pass; ``_d_step``
        replays it there), then the backward of 0.5 * loss_D_real into D's (freshly zeroed) gradient buffers -- the same
        terms, in the same accumulation order (real before fake), as the single backward of 0.5 * (real + fake)."""
        set_requires_grad([self.gan_model], True)
        self.gan_optimizer.zero_grad()
        stats = []
        ops.BN_STATS_SINK, ops.BN_DEFER_RUNNING = stats, True
        try:
            d_real = self.gan_model(clean_feat.detach(), enhance_cmvn)
        finally:
            ops.BN_STATS_SINK, ops.BN_DEFER_RUNNING = None, False
        loss_D_real = self.criterionGAN(d_real, True)
        gan_params = [p for p in self.gan_model.parameters() if p.requires_grad]
        torch.autograd.grad(loss_D_real * 0.5 if hold is None else loss_D_real * 0.5 * hold, gan_params, allow_unused=True)
        self._ev_dreal_wgrad = None
        if ops.WGRAD_STREAM is not None:
            self._ev_dreal_wgrad = torch.cuda.Event()          # behind this half's weight gradients on the weight-gradient stream
            self._ev_dreal_wgrad.record(ops.WGRAD_STREAM)
        return loss_D_real.detach(), stats

    def _d_step(self, clean_feat, enhance_feat, enhance_cmvn, d_fake=None, fake_stats=None, fake_bn=None, real_part=None, hold=None):
        """Discriminator step (joint_train.py:195-212) on the CURRENT stream up to and including the clipping of its gradients; the update
        itself (``gan_optimizer.step()``) is the caller's, behind the step's gate (``_step``).  ``d_fake``: D(enhance_feat) of the G-step
        (same input, same weights as upstream's second evaluation) -- its graph is walked again for the parameter
        gradients instead of recomputing the forward; the BatchNorm running statistics get the update that forward would
        have applied (``fake_stats``), in upstream's order (after the D(real) pass)."""
        opt = self.opt
        set_requires_grad([self.gan_model], True)
        # D's weight gradients of THIS pass run inline, on the stream of the D-step (the side stream), not behind the weight-gradient stream's
        # backlog (round 6): that stream was the last of the three to finish (46.6 ms of the step against 45.7 / 45.6), and the D-step's update
        # waited for all of it.  46.47 -> 45.99 ms per step (five interleaved rounds, profiles/r06_ab_dstep_inline_wgrad.txt).
        # RE2E_DSTEP_WGRAD_STREAM=1 (experiments build): as before.
        wg_keep = ops.WGRAD_STREAM
        if lib.exp_env('RE2E_DSTEP_WGRAD_STREAM') != '1':
            ops.WGRAD_STREAM = None
        if real_part is not None:
            loss_D_real, real_stats = real_part
            replay_running_stats(real_stats)
        else:
            self.gan_optimizer.zero_grad()
            loss_D_real = self.criterionGAN(self.gan_model(clean_feat.detach(), enhance_cmvn), True)
        if d_fake is not None:
            replay_running_stats(fake_stats)
            for bn in fake_bn:
                bn.num_batches_tracked += 1
            loss_D_fake = self.criterionGAN(d_fake, False)
            loss_D = (loss_D_real + loss_D_fake) * 0.5
            gan_params = [p for p in self.gan_model.parameters() if p.requires_grad]
            torch.autograd.grad(loss_D if hold is None else loss_D * hold, gan_params, allow_unused=True)      # the fused ops accumulate the gradients themselves
        else:
            loss_D_fake = self.criterionGAN(self.gan_model(enhance_feat.detach(), enhance_cmvn), False)
            loss_D = (loss_D_real + loss_D_fake) * 0.5
            (loss_D if hold is None else loss_D * hold).backward()
        inline = ops.WGRAD_STREAM is None and wg_keep is not None
        ops.WGRAD_STREAM = wg_keep
        if inline and real_part is not None and getattr(self, '_ev_dreal_wgrad', None) is not None:
            torch.cuda.current_stream().wait_event(self._ev_dreal_wgrad)      # only the early real half went through the weight-gradient stream
        elif ops.WGRAD_STREAM is not None:
            torch.cuda.current_stream().wait_stream(ops.WGRAD_STREAM)
        GradSync().finish([self.gan_optimizer])
        self.gan_optimizer.clip_grad_norm(opt.grad_clip)
        return loss_D

    def validate(self, data, enhance_cmvn, want_attention=False):
        """One validation batch (joint_train.py:245-275): no-grad pass with enhancer / fbank / ASR in eval mode.  As in the
        reference the discriminator is NOT switched to eval (its BatchNorm keeps using batch statistics and moves its
        running statistics).  Returns the ``val/*`` meters as device scalars; ``want_attention`` adds ``att_ws`` =
        ``calculate_all_attentions(enhance_feat, ...)`` (numpy, (B, Lmax+1, T'))."""
        opt = self.opt
        clean_inputs, mix_inputs, mix_log_inputs, targets, input_sizes, target_sizes = data[2], data[4], data[5], data[7], data[8], data[9]
        nets = [self.enhance_model, self.feat_model, self.asr_model]
        modes = [m.training for m in nets]
        for m in nets:
            m.eval()
        ops.MULTI_STREAM, ops.WGRAD_STREAM, ops.AUX_STREAM = False, None, None
        try:
            with torch.no_grad():
                enhance_out = self.enhance_model(mix_inputs, mix_log_inputs, input_sizes)
                enhance_feat = self.feat_model(enhance_out)
                clean_feat = self.feat_model(clean_inputs)
                enhance_loss = ops.mean_loss(enhance_feat, clean_feat, 0.0, _LOSS_KIND[opt.enhance_loss_type])
                errors = {}
                if self.isGAN:
                    gan_loss = self.criterionGAN(self.gan_model(enhance_feat, enhance_cmvn), True)
                    enhance_loss = enhance_loss + opt.gan_loss_lambda * gan_loss
                    errors['val/gan_loss'] = opt.gan_loss_lambda * gan_loss
                loss_ctc, loss_att, acc, _, _ = self.asr_model(clean_feat, enhance_feat, targets, input_sizes, target_sizes, 0.0, enhance_cmvn)
                asr_loss = opt.mtlalpha * loss_ctc.view(()) + (1 - opt.mtlalpha) * loss_att
                enhance_loss = opt.enhance_loss_lambda * enhance_loss
                errors.update({'val/loss': asr_loss + enhance_loss, 'val/loss_ctc': loss_ctc.view(()), 'val/acc': acc, 'val/loss_att': loss_att,
                               'val/enhance_loss': enhance_loss})
                if want_attention and opt.mtlalpha != 1.0:
                    errors['att_ws'] = self.asr_model.calculate_all_attentions(enhance_feat, targets, input_sizes, target_sizes, enhance_cmvn)
        finally:
            for m, t in zip(nets, modes):
                m.train(t)
        return errors

    def fit(self, train_loader, val_loader, visualizer, train_sampler=None, start_epoch=0, iters=0, best_loss=float('inf'), best_acc=0.0,
            max_iters=None, prefetch=False):
        """The reference's training loop (joint_train.py:145-329) around ``step`` / ``validate``: CMVN estimate before
        training and after every validation, scheduled-sampling rate updated only at validation time, ``print_freq``
        logging + 'latest' checkpoint, ``validate_freq`` validation + model selection (``opt.criterion`` 'acc' / 'loss';
        a worse score decays Adadelta's eps, a better one is saved as model.{acc,loss}.best) with the reference's
        checkpoint keys.  The per-step meters are read back one iteration late, after the next step has been enqueued,
        so that logging never drains the GPU.  Returns (iters, best_loss, best_acc).

        ``prefetch``: ``train_loader`` yields UN-collated batches (lists of samples, or ``data.prefetch.Staged`` objects prepared by
        loader workers) and they go through ``data.prefetch.DevicePrefetcher``: pinned staging, H2D and zero padding on a copy
        stream two batches ahead of the step that consumes them (set GPU_MAX_HW_QUEUES=8: the step already runs on four
        streams).  ``prefetch`` may also be a collate callable ``(batch, device, pool) -> 10-tuple`` (e.g. one that wraps
        ``collate_kaldi_device`` for raw Kaldi records)."""
        from .utils import utils
        opt = self.opt
        def loader():
            if not prefetch:
                return train_loader
            from .data.prefetch import DevicePrefetcher, collate_device_pinned
            return DevicePrefetcher(train_loader, next(self.enhance_model.parameters()).device,
                                    collate=prefetch if callable(prefetch) else collate_device_pinned)
        fm = self.feat_model
        cmvn_keys = ('sum', 'sum_sq', 'frame_count', 'cmvn_processed_num')

        def cmvn_estimate():
            # (compute_cmvn's host-side accumulators are never reset upstream: a repeated pass starts from the values this one found)
            return self.run_guarded(lambda: compute_cmvn_epoch(opt, loader(), self.enhance_model, fm),
                                    lambda: {k: copy.deepcopy(getattr(fm, k)) for k in cmvn_keys if hasattr(fm, k)},
                                    lambda snap: [setattr(fm, k, v) for k, v in snap.items()], 'the CMVN estimate')
        enhance_cmvn = cmvn_estimate()
        rampup = utils.ScheSampleRampup(opt.sche_samp_start_iter, opt.sche_samp_final_iter, opt.sche_samp_final_rate)
        sche_samp_rate = rampup.update(iters)
        acc_report = loss_report = None
        for m in (self.enhance_model, self.feat_model, self.asr_model):
            m.train()
        pending = None
        writer = rdist.rank() == 0          # data parallel: one rank writes checkpoints / plots into exp_path

        def flush():
            """Read back the meters of the step enqueued before the current one.  Returns True when that step had to be REPEATED because a
            persistent kernel gave up in it: whatever was enqueued behind it was held on the device (its losses were multiplied by NaN,
            ``_step``) and has to be run again by the caller."""
            nonlocal pending
            repeated = False
            if pending is not None:
                vals = self.finish_read(pending['_read'])
                gn, mine = vals.pop('grad_norm', 0.0), int(vals.pop('aborts', 0.0))
                if not math.isfinite(gn):             # joint_train.py:189-193: the update was skipped on the device
                    # ... either by non-finite numbers of the model's own (upstream: warn and go on), or because a persistent kernel gave up
                    # on a peer workgroup and poisoned its outputs: then the step is REPEATED with the launch-per-step kernels
                    redo = self.recover_aborted_step(pending['_entry'], mine) if pending.get('_entry') is not None else None
                    if redo is None:
                        logging.warning('grad norm is nan. Do not update model.')
                    else:
                        vals, repeated = redo, True
                elif mine:
                    self.unexplained_aborts += mine   # a give-up that did NOT show up as a skipped update: raised at the next print boundary
                visualizer.set_current_errors(vals)
                pending = None
            return repeated

        def check_recurrences():
            # a persistent kernel that gave up on a peer workgroup poisons its outputs with NaN and the step gate refuses the update;
            # flush() repeats such a step; the validation pass and the CMVN estimate, which have no gate, check the device counter themselves
            # (run_guarded).  What is left to check here is a give-up that the gate counted in a step whose norm was finite all the same
            # (collective: all replicas fail together).
            n = rdist.any_rank(self.unexplained_aborts)
            if n != 0:
                raise lib.Re2eError('%d recurrent sequences were aborted by a persistent kernel (a peer workgroup never arrived) in a step whose '
                                    'update was not refused' % n)

        for epoch in range(start_epoch, opt.epochs):
            if train_sampler is not None and epoch > opt.shuffle_epoch:
                train_sampler.shuffle(epoch)
            for data in loader():
                entry = (data, sche_samp_rate, enhance_cmvn, self._bn_snapshot(), self._rng_snapshot())
                errors = self.step(data, sche_samp_rate, enhance_cmvn)
                if flush():                               # previous step's meters, now that this step is queued
                    # the previous step was repeated: THIS one ran behind the give-up and was held (it applied nothing; what it did to D's
                    # running statistics was put back with the previous step's snapshot): run it again, on the repaired state
                    self._uncount_step()
                    entry = (data, sche_samp_rate, enhance_cmvn, self._bn_snapshot(), self._rng_snapshot())
                    errors = self.step(data, sche_samp_rate, enhance_cmvn)
                pending = {k: v for k, v in errors.items() if k.startswith('train/') or k in ('grad_norm', 'aborts')}
                pending['_read'] = self.start_read(pending)        # the copy is enqueued NOW, behind this step; read after the next one is enqueued
                pending['_entry'] = entry
                iters += 1
                if iters % opt.print_freq == 0:
                    flush()
                    check_recurrences()
                    visualizer.print_current_errors(epoch, iters)
                    if self.isGAN:
                        rdist.average_buffers_(self.gan_model)
                    if writer:
                        st = self.state(epoch, iters, best_loss, best_acc)
                        st.update(acc_report=acc_report, loss_report=loss_report)
                        utils.save_checkpoint(st, opt.exp_path, filename='latest')
                if iters % opt.validate_freq == 0:
                    flush()
                    sche_samp_rate = rampup.update(iters)
                    def validation_pass():
                        got, saved = [], 0
                        for vdata in val_loader:
                            want = opt.num_save_attention > 0 and opt.mtlalpha != 1.0 and saved < opt.num_save_attention
                            verr = self.validate(vdata, enhance_cmvn, want_attention=want)
                            got.append((vdata, self.to_floats(verr), verr.get('att_ws') if want else None))
                            if want:
                                saved += len(vdata[0]) if vdata[0] is not None else verr['att_ws'].shape[0]
                        return got
                    # (the meters and plots reach the visualizer only when no persistent kernel gave up in the pass on any replica: NaN validation
                    #  scores would otherwise enter best_loss / best_acc and the model selection below.  D's BatchNorm buffers move in validation,
                    #  as upstream: a repeated pass starts from their values before the first one.)
                    results = self.run_guarded(validation_pass, self._bn_snapshot,
                                               lambda snap: [b.copy_(s_) for b, s_ in zip(self.gan_model.buffers(), snap)] if self.isGAN else None,
                                               'the validation pass')
                    saved = 0
                    for vdata, vfloats, att_ws in results:
                        visualizer.set_current_errors(vfloats)
                        if att_ws is not None:
                            for x in range(len(vdata[0]) if vdata[0] is not None else att_ws.shape[0]):
                                if saved >= opt.num_save_attention:
                                    break
                                name = vdata[0][x] if vdata[0] is not None else 'utt%d' % x
                                visualizer.plot_attention(att_ws[x], int(vdata[9][x]), int(vdata[8][x]), '{}_ep{}_it{}.png'.format(name, epoch, iters))
                                saved += 1
                    visualizer.print_epoch_errors(epoch, iters)
                    acc_report = visualizer.plot_epoch_errors(epoch, iters, 'acc.png')
                    loss_report = visualizer.plot_epoch_errors(epoch, iters, 'loss.png')
                    val_loss, val_acc = visualizer.get_current_errors('val/loss'), visualizer.get_current_errors('val/acc')
                    filename = None
                    if opt.criterion == 'acc' and opt.mtlalpha != 1.0:
                        if val_acc < best_acc:
                            opt.eps = utils.adadelta_eps_decay(self.asr_optimizer, opt.eps_decay)
                        else:
                            filename = 'model.acc.best'
                        best_acc = max(best_acc, val_acc)
                    elif opt.criterion == 'loss':
                        if val_loss > best_loss:
                            opt.eps = utils.adadelta_eps_decay(self.asr_optimizer, opt.eps_decay)
                        else:
                            filename = 'model.loss.best'
                        best_loss = min(val_loss, best_loss)
                    check_recurrences()
                    if self.isGAN:
                        rdist.average_buffers_(self.gan_model)
                    if writer:
                        st = self.state(epoch, iters, best_loss, best_acc)
                        st.update(acc_report=acc_report, loss_report=loss_report)
                        utils.save_checkpoint(st, opt.exp_path, filename=filename)
                    visualizer.reset()
                    enhance_cmvn = cmvn_estimate()
                if max_iters is not None and iters >= max_iters:
                    flush()
                    return iters, best_loss, best_acc
        flush()
        return iters, best_loss, best_acc

    # ---- a persistent kernel that gave up (csrc/lstm.hip, csrc/decloop.hip: bounded spins, NaN outputs, counted on the device) ----------------
    # Protocol (round 5).  The meters of step k are read one step late, when step k + 1 is already enqueued, so everything that must happen
    # BEFORE the host knows is decided on the device by re2e_step_gate at the end of every step:
    #   * the update of a step in which a kernel gave up is refused (also when the NaN did not reach the ASR norm: enhancer backward chain),
    #     D's included -- D's optimizer step is the step's last kernel;
    #   * BatchNorm running statistics are never moved by non-finite batch statistics (bn_finalize_kernel), so a snapshot taken before a
    #     step is always clean;
    #   * the step after it is HELD: its losses are multiplied by NaN (the factor the gate wrote), it applies nothing on any replica;
    #   * the count of give-ups since the last acknowledgement travels with each step's meters, so a give-up is attributed to the step it
    #     happened in, not to whatever the host had enqueued by the time it looked.
    # The host then (``fit``): puts back the aborted step's BatchNorm snapshot (which also undoes the held step's forward), acknowledges the
    # count, repeats the aborted step with the launch-per-step kernels and runs the held step again -- the updates land in the order of an
    # undisturbed run.
    recovered_steps = 0
    unexplained_aborts = 0

    def _bn_snapshot(self):
        """D's BatchNorm buffers (a few KB) before a step: what ``recover_aborted_step`` puts back before it repeats the step."""
        return [b.detach().clone() for b in self.gan_model.buffers()] if self.isGAN else []

    @staticmethod
    def _rng_snapshot():
        """What a step draws from besides its batch: the dropout mask stream (seed, next mask index) and Python's RNG (one draw per output token for
        scheduled sampling, e2e_decoder.py:123).  Put back in front of the repeat of an aborted step: the repeat then draws what the aborted
        attempt drew, and the held step behind it -- run again afterwards -- what it would have drawn in an undisturbed run."""
        import random
        return ops.dropout_state(), random.getstate()

    @staticmethod
    def _rng_restore(snap):
        import random
        (seed, call), pystate = snap
        ops.dropout_seed(seed, call)
        random.setstate(pystate)

    def _uncount_step(self):
        """A step whose update the gate refused because of a give-up is run again: it must not count twice (Adam's bias correction)."""
        for o in (self.enhance_optimizer, self.asr_optimizer, self.gan_optimizer):
            if o is not None:
                o.step_count -= 1

    def recover_aborted_step(self, entry, mine):
        """Called for a step whose update the device-side gate refused.  ``mine``: give-ups counted on this device in it.  If there were any
        (on ANY replica: the averaged gradients carried the NaN to all of them, so every replica is here and the agreement below is a
        collective all of them reach), the step is repeated with the launch-per-step recurrences -- same arithmetic, no in-launch hand-off that
        can time out -- on the state it first ran on.  Returns the repeated step's meters, or None if nothing gave up (the NaN was the
        model's own).  Raises if the repeated step is not finite either."""
        data, rate, cmvn, bn = entry[:4]
        if rdist.any_rank(max(0, int(mine))) == 0:
            return None
        if len(entry) > 4:
            self._rng_restore(entry[4])
        if self.recovered_steps == 0:
            logging.warning('a persistent kernel gave up on a peer workgroup (%d sequence(s) on this rank): the step is repeated with the '
                            'launch-per-step kernels.  Further repeats are counted in JointTrainer.recovered_steps', mine)
        self.recovered_steps += 1
        if self.isGAN:
            for b, s in zip(self.gan_model.buffers(), bn):
                b.copy_(s)
        self._uncount_step()
        self._acknowledge_aborts()
        with stepwise_kernels():
            vals = self.to_floats({k: v for k, v in self.step(data, rate, cmvn).items() if k.startswith('train/') or k in ('grad_norm', 'aborts')})
        gn, again = vals.pop('grad_norm', 0.0), int(vals.pop('aborts', 0.0))
        if rdist.any_rank(1 if (again or not math.isfinite(gn)) else 0):
            raise lib.Re2eError('a step that a persistent kernel had aborted is not finite with the launch-per-step kernels either (grad norm %r)' % gn)
        return vals

    @staticmethod
    def start_read(errors):
        """First half of ``to_floats``: ONE device-to-host copy of the meters into pinned memory, enqueued on the current stream behind what is
        already there, and an event behind it.  ``finish_read`` waits for THAT event only -- ``fit`` enqueues the next step in between, and reading
        the meters of step k must not drain step k + 1 (a ``.cpu()`` at that point waits for everything the stream has been given since: the host
        would start enqueueing step k + 2 only when step k + 1 has ended, 1.5 ms of idle GPU per step, profiles/r06_soak.txt)."""
        keys = [k for k, v in errors.items() if isinstance(v, torch.Tensor)]      # (att_ws is a numpy array)
        if not keys:
            return (keys, None, None, 0, None)
        flags = [f for f in ops.sync_bn_flags() if f.device == errors[keys[0]].device]
        dev = torch.stack([errors[k].detach().double().reshape(()) for k in keys] + flags)
        if not dev.is_cuda:
            return (keys, dev, None, len(flags), None)
        host = torch.empty(dev.shape, dtype=dev.dtype, pin_memory=True)
        host.copy_(dev, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        return (keys, host, ev, len(flags), dev)          # (``dev`` rides along: it must outlive the copy)

    @staticmethod
    def finish_read(handle):
        """Second half of ``to_floats``: the meters as host floats.  The ragged-shard counters of synchronised BatchNorm (ops._sync_rows_poison)
        ride along and raise here, on every rank in the same step."""
        keys, host, ev, nflags, _ = handle
        if host is None:
            return {}
        if ev is not None:
            ev.synchronize()
        vals = host.tolist()
        if nflags:
            ops.check_sync_bn(vals[len(keys):])
        return dict(zip(keys, vals[:len(keys)]))

    @staticmethod
    def to_floats(errors):
        """The meters as host floats: ONE device-to-host copy -- the only host synchronisation of a step (``start_read`` + ``finish_read``)."""
        return JointTrainer.finish_read(JointTrainer.start_read(errors))

    def state(self, epoch, iters, best_loss=float('inf'), best_acc=0.0):
        """checkpoint dict with the reference's keys (joint_train.py:225-233)."""
        st = {'asr_state_dict': self.asr_model.state_dict(), 'fbank_state_dict': self.feat_model.state_dict(),
              'enhance_state_dict': self.enhance_model.state_dict(), 'opt': self.opt, 'epoch': epoch, 'iters': iters,
              'eps': self.opt.eps, 'lr': self.opt.lr, 'best_loss': best_loss, 'best_acc': best_acc, 'acc_report': None, 'loss_report': None}
        if self.isGAN:
            st['gan_state_dict'] = self.gan_model.state_dict()
        # the dropout mask stream: the BASE seed (opt.seed) and the index of the next mask -- not the writing rank's own stream seed, which
        # every replica would otherwise inherit on resume (the index is the same on all ranks: they make the same calls); not an upstream key
        st['dropout_state'] = {'base_seed': int(getattr(self.opt, 'seed', 1234)), 'call': int(ops.dropout_state()[1])}
        return st

    @staticmethod
    def _dropout_stream_seed(base_seed):
        """One mask stream per process: replicas draw different masks, as upstream's per-process RNG would."""
        return (int(base_seed) * 1000003 + rdist.rank()) & 0xFFFFFFFFFFFF

    @classmethod
    def restore_dropout(cls, package):
        """Continue the dropout mask stream of a checkpoint written by ``state()``: this rank's stream (re-derived from the base seed) at the
        saved mask index.  No-op for checkpoints without the key; a (seed, index) pair of an older checkpoint is taken as it is."""
        ds = package.get('dropout_state') if isinstance(package, dict) else None
        if isinstance(ds, dict):
            ops.dropout_seed(cls._dropout_stream_seed(ds['base_seed']), int(ds['call']))
        elif ds is not None:
            ops.dropout_seed(int(ds[0]), int(ds[1]))

    def load_state(self, package):
        """Counterpart of ``state()`` (upstream: the ``--joint_resume`` branch, joint_train.py:98-111 and :127-140): the networks'
        state_dicts, the dropout mask stream, and eps / lr -- into ``opt`` AND into every optimizer of the trainer, because upstream builds
        its three optimizers AFTER reading the package (Adadelta(eps=package eps) / Adam(lr=package lr)): a resume after an
        ``adadelta_eps_decay`` must train with the decayed eps.  As upstream, the optimizers' accumulators are NOT part of a checkpoint: they
        restart from zero (``state()`` has no key for them), which is what a freshly constructed trainer has.
        Returns (epoch, iters, best_loss, best_acc) with ``iters`` = the checkpoint's ``iters`` - 1, as upstream resumes
        (joint_train.py:108: ``int(package.get('iters', 0)) - 1``; the 'latest' checkpoint is written at ``iters % print_freq == 0``, so the
        resumed run reaches the next print / validation boundary one step later than the uninterrupted one would have)."""
        self.asr_model.load_state_dict(package['asr_state_dict'])
        self.feat_model.load_state_dict(package['fbank_state_dict'])
        self.enhance_model.load_state_dict(package['enhance_state_dict'])
        if self.isGAN and 'gan_state_dict' in package:
            self.gan_model.load_state_dict(package['gan_state_dict'])
        for k in ('eps', 'lr'):
            if k in package:
                setattr(self.opt, k, package[k])
        for o in (self.enhance_optimizer, self.asr_optimizer, self.gan_optimizer):
            if o is None:
                continue
            g = o.param_groups[0]
            if o.kind == 'adadelta' and 'eps' in package:
                g['eps'] = float(package['eps'])
            if o.kind == 'adam' and 'lr' in package:
                g['lr'] = float(package['lr'])
        self.restore_dropout(package)
        return (int(package.get('epoch', 0)), int(package.get('iters', 0)) - 1, float(package.get('best_loss', float('inf'))),
                float(package.get('best_acc', 0.0)))


def config4_opt(**over):
    """Namespace with BASELINE.json config-4 architecture."""
    import argparse
    V = 4233
    d = dict(idim=257, odim=V, fbank_dim=80, char_list=[str(i) for i in range(V)], gpu_ids=[0], verbose=0, enhance_type='blstm',
             enhance_layers=2, enhance_units=256, enhance_projs=256, dropout_rate=0.0, subsample_type='skip', subsample='1_1_1_1_1',
             fbank_opti_type='frozen', train_dataset_len=128, num_utt_cmvn=20000, etype='vggblstmp', elayers=3, eunits=512, eprojs=512,
             atype='location', adim=320, aconv_chans=10, aconv_filts=100, awin=5, aheads=4, dlayers=1, dunits=300, mtlalpha=0.5,
             lsm_type='', lsm_weight=0.0, labeldist=None, fusion='', lmtype=None, rnnlm=None, ndf=64, norm_D='batch', input_nc=1,
             n_layers_D=3, no_lsgan=False, netD_type='basic', enhance_loss_type='L2', enhance_loss_lambda=1.0, coral_loss_lambda=0.1,
             gan_loss_lambda=1.0, grad_clip=5.0, eps=1e-8, isGAN=True, opt_type='adadelta', lr=0.005, beta1=0.5, exp_path=None)
    d.update(over)
    return argparse.Namespace(**d)
