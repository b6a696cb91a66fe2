"""E2E (mirror of model/e2e_model.py:20-202) and ShareE2E (S1: missing upstream, contract from
joint_train.py:97,120,170,266,280 -- build-defined)."""
import logging
import sys

import numpy as np
import torch

from .. import ops
from ..lib import Re2eError
from .e2e_attention import AttLoc
from .e2e_common import ModelBase, host_to_dev, lecun_normal_init_parameters, lens_dev, lens_list, set_forget_bias_to_one, to_cuda
from .e2e_ctc import CTC
from .e2e_decoder import Decoder
from .e2e_encoder import Encoder


class E2E(ModelBase):
    def __init__(self, args):
        super(E2E, self).__init__()
        self.opt = args
        idim, odim = args.fbank_dim, args.odim
        self.etype, self.verbose = args.etype, args.verbose
        self.char_list = args.char_list
        self.mtlalpha = args.mtlalpha
        self.sos = self.eos = odim - 1                                   # e2e_model.py:34-35
        subsample = np.ones(args.elayers + 1, dtype=int)
        if args.etype == 'blstmp':
            ss = args.subsample.split('_')
            for j in range(min(args.elayers + 1, len(ss))):
                subsample[j] = int(ss[j])
        self.subsample = subsample
        labeldist = args.labeldist if getattr(args, 'lsm_type', '') else None
        self.enc = Encoder(args.etype, idim, args.elayers, args.eunits, args.eprojs, self.subsample, args.subsample_type,
                           args.dropout_rate)
        self.ctc = CTC(odim, args.eprojs, args.dropout_rate)
        if args.atype == 'location':
            self.att = AttLoc(args.eprojs, args.dunits, args.adim, args.aconv_chans, args.aconv_filts, 'softmax')
        elif args.atype in ('noatt', 'dot', 'add', 'location2d', 'location_recurrent', 'coverage', 'coverage_location', 'multi_head_dot',
                            'multi_head_add', 'multi_head_loc', 'multi_head_multi_res_loc'):
            raise Re2eError('atype %s is out of scope: only the default location-aware attention is on the hot path' % args.atype)
        else:
            logging.error('Error: need to specify an appropriate attention archtecture')
            sys.exit()
        if getattr(args, 'fusion', '') in ('deep_fusion', 'cold_fusion'):
            raise Re2eError('LM fusion is decode-time only and out of scope (SURVEY section 2, row 15)')
        self.dec = Decoder(args.eprojs, odim, args.dlayers, args.dunits, self.sos, self.eos, self.att, self.verbose, self.char_list,
                           labeldist, args.lsm_weight)
        self.init_like_chainer()

    def init_like_chainer(self):
        """e2e_model.py:148-166: LeCun normal, embed ~ N(0,1), decoder forget-gate bias 1."""
        lecun_normal_init_parameters(self)
        self.dec.embed.weight.data.normal_(0, 1)
        for l in range(len(self.dec.decoder)):
            set_forget_bias_to_one(self.dec.decoder[l].bias_ih)

    @staticmethod
    def _split_targets(targets, target_sizes):
        ys, off = [], 0
        for s in lens_list(target_sizes):
            ys.append(targets[off:off + s])
            off += s
        return ys

    def recognize(self, x, recog_args, char_list=None, rnnlm=None, fstlm=None):
        """e2e_model.py:204-236: beam search for ONE utterance ``x`` (1, T, fbank_dim) -> n-best [{'yseq', 'score'}]."""
        prev = self.training
        self.eval()
        try:
            with torch.no_grad():
                xs = to_cuda(self, x)
                hpad, _ = self.enc(xs, [xs.shape[1]])
                lpz = self.ctc.log_softmax(hpad)[0] if recog_args.ctc_weight > 0.0 else None
                return self.dec.recognize_beam(hpad[0], lpz, recog_args, char_list, rnnlm, fstlm)
        finally:
            if prev:
                self.train()

    def forward(self, inputs, targets, input_sizes, target_sizes, scheduled_sampling_rate=0.0):
        """-> (loss_ctc, loss_att, acc)   (e2e_model.py:169-202)"""
        xpad = to_cuda(self, inputs)
        ilens = lens_list(input_sizes)
        ys = self._split_targets(targets, target_sizes)
        h_tm, hlens = self.enc.forward_tm(xpad, ilens)
        # CTC and the attention decoder only share the encoder output: with a filler stream available (trainers.StepStreams, JointTrainer) the
        # CTC head -- ctc_lo product, softmax, alpha / beta; autograd runs its backward on the same stream -- goes beside the decoder's
        # latency-bound token loop instead of in front of it (ShareE2E.forward does the same with its two heads)
        aux = ops.AUX_STREAM if (ops.MULTI_STREAM and self.mtlalpha not in (0, 1)) else None
        if aux is not None:
            cur = torch.cuda.current_stream()
            aux.wait_stream(cur)
            with torch.cuda.stream(aux):
                h_tm.record_stream(aux)
                loss_ctc = self.ctc.forward_tm(h_tm, hlens, ys)
        else:
            loss_ctc = self.ctc.forward_tm(h_tm, hlens, ys) if self.mtlalpha != 0 else None
        if self.mtlalpha == 1:
            loss_att, acc = None, None
        else:
            loss_att, acc = self.dec(ops.transpose01(h_tm), hlens, ys, scheduled_sampling_rate)
        if aux is not None:
            cur.wait_stream(aux)
            loss_ctc.record_stream(cur)
        return loss_ctc, loss_att, acc

    def calculate_all_attentions(self, inputs, targets, input_sizes, target_sizes):
        with torch.no_grad():
            hpad, hlens = self.enc(to_cuda(self, inputs), lens_list(input_sizes))
            return self.dec.calculate_all_attentions(hpad, hlens, self._split_targets(targets, target_sizes))


class ShareE2E(E2E):
    """S1.  forward(clean_feat, enhance_feat, targets, input_sizes, target_sizes, ss_rate, cmvn) ->
    (loss_ctc, loss_att, acc, clean_context, mix_context): both inputs are CMVN-normalised
    ``(x + cmvn[0]) * cmvn[1]`` (cf. joint_recog.py:148), the SHARED encoder runs on both branches as
    one 2B batch, losses come from the enhanced branch exactly as E2E.forward, contexts are the
    encoder states of the valid frames of each branch."""

    def encode_clean(self, clean_feat, cmvn, input_sizes=None):
        """Clean-branch CMVN + VGG conv stack, to be enqueued on a SIDE stream while the enhancer's recurrent chain
        occupies the main stream (the clean branch does not depend on the enhancer).  Returns a handle for
        ``forward(..., clean_branch=handle)``."""
        cln = to_cuda(self, clean_feat)
        if cmvn is not None:
            cln = ops.cmvn_pair(cln, None, to_cuda(self, cmvn).float().contiguous())
        h = self.enc.enc1.conv_stack(cln, input_sizes)
        ev = torch.cuda.Event()
        ev.record()
        return h, ev

    def forward(self, clean_feat, enhance_feat, targets, input_sizes, target_sizes, scheduled_sampling_rate=0.0, cmvn=None,
                clean_branch=None, context_loss=None):
        enh = to_cuda(self, enhance_feat)
        cln = to_cuda(self, clean_feat)
        ilens = lens_list(input_sizes)
        B = enh.shape[0]
        ys = self._split_targets(targets, target_sizes)
        cm = to_cuda(self, cmvn).float().contiguous() if cmvn is not None else None
        if clean_branch is not None and self.etype in ('vggblstmp', 'vggblstm'):
            # the two branches share only the recurrent stack: conv stacks separately (possibly on different streams),
            # then ONE (T', 2B, .) tensor for the BLSTMP so that the sequential chain is paid once
            h_cln, ev = clean_branch
            h_enh = self.enc.enc1.conv_stack(ops.cmvn_pair(enh, None, cm) if cm is not None else enh, ilens)
            torch.cuda.current_stream().wait_event(ev)
            h_cln.record_stream(torch.cuda.current_stream())
            if torch.is_grad_enabled() and h_cln.requires_grad:
                # cut the graph at the clean conv stack's output: the caller runs that backward itself
                # (``clean_cut`` = (graph tensor, leaf); d(loss)/d(leaf) -> backward of the graph tensor) so that it can put
                # it on the side stream WITHOUT the end-of-backward join autograd would add between the two streams
                leaf = h_cln.detach().requires_grad_(True)
                self.clean_cut = (h_cln, leaf)
                h_cln = leaf
            ops.mark_grad(h_enh, 'h_enh (BLSTMP bwd done, main)')
            ops.mark_grad(h_cln, 'h_cln (side: clean conv bwd starts)')
            ops.mark('  conv stack (enhanced branch) fwd done')
            h_in, hl2 = self.enc.enc1.pack_tm([h_enh, h_cln], [ilens, ilens])
            h_tm2 = self.enc.enc2.forward_tm(h_in, lens_dev(hl2, enh.device))
            ops.mark('  shared BLSTMP fwd done')
        else:
            x2 = ops.cmvn_pair(enh, cln, cm) if cm is not None else torch.cat([enh, cln], 0)
            h_tm2, hl2 = self.enc.forward_tm(x2, ilens + ilens)           # (T', 2B, E)
        hlens = hl2[:B]
        hpad2 = ops.transpose01(h_tm2)                                     # (2B, T', E)
        ops.mark_grad(hpad2, 'hpad2 (decoder, CTC, CORAL bwd done)')
        hpad_enh, hpad_cln = hpad2[:B], hpad2[B:]
        Tq, E = hpad2.shape[1], hpad2.shape[2]
        # CTC and the attention decoder only share the encoder output: with a filler stream available the CTC branch
        # (ctc_lo GEMM, softmax, alpha/beta; autograd runs its backward on the same stream) goes beside the decoder's
        # 41 latency-bound steps instead of in front of them.
        aux = ops.AUX_STREAM if (ops.MULTI_STREAM and self.mtlalpha != 1) else None
        # ``context_loss`` (optional callable (clean_context, mix_context) -> scalar, e.g. the trainer's CORAL term): the contexts and
        # that loss depend only on the encoder output, so with a filler stream they are enqueued there, in front of the decoder --
        # forward AND backward (autograd runs a node's backward on its forward stream) then run beside the decoder's latency-bound
        # loop instead of between its forward and backward on the critical stream (round 3: ~0.45 ms of small kernels).  The value is
        # left in ``self.last_context_loss``.
        # (Measured and rejected, round 3: differentiating the two heads right here, behind their forward, and re-injecting their
        # gradient at the encoder output -- the engine enqueues their backward only after the decoder's ~250 backward launches, so
        # the critical stream waits ~0.9 ms for it -- removes that wait but puts the CTC / CORAL backward beside the decoder's
        # FORWARD loop, which it slows by more: 65.0 against 64.7 ms per step.)
        def heads_on_aux():
            l_ctc = self.ctc.forward(hpad_enh, hlens, ys) if self.mtlalpha != 0 else None
            idx = host_to_dev(np.concatenate([b * Tq + np.arange(hlens[b], dtype=np.int32) for b in range(B)]).astype(np.int32), enh.device)
            mix_c = ops.gather_rows(hpad_enh.reshape(B * Tq, E), idx)
            cln_c = ops.gather_rows(hpad_cln.reshape(B * Tq, E), idx)
            l_ctx = context_loss(cln_c, mix_c) if context_loss is not None else None
            return l_ctc, cln_c, mix_c, l_ctx
        # (... and so does enqueueing the heads BEHIND the decoder, which makes their backward nodes the first the engine enqueues:
        # beside the decoder's backward loop they cost it more than the wait they remove, 65.3 against 64.75 ms.)
        if aux is not None:
            cur = torch.cuda.current_stream()
            aux.wait_stream(cur)
            with torch.cuda.stream(aux):
                hpad2.record_stream(aux)
                loss_ctc, clean_context, mix_context, self.last_context_loss = heads_on_aux()
                ops.mark('aux stream: CTC + CORAL heads fwd done')
        else:
            loss_ctc, clean_context, mix_context, self.last_context_loss = heads_on_aux()
        if self.mtlalpha == 1:
            loss_att, acc = None, None
        else:
            loss_att, acc = self.dec(hpad_enh, hlens, ys, scheduled_sampling_rate)
        if aux is not None:
            cur = torch.cuda.current_stream()
            cur.wait_stream(aux)
            for t_ in (loss_ctc, clean_context, mix_context, self.last_context_loss):
                if t_ is not None:
                    t_.record_stream(cur)
        return loss_ctc, loss_att, acc, clean_context, mix_context

    def calculate_all_attentions(self, enhance_feat, targets, input_sizes, target_sizes, cmvn=None):
        with torch.no_grad():
            enh = to_cuda(self, enhance_feat)
            if cmvn is not None:
                enh = ops.cmvn_pair(enh, None, to_cuda(self, cmvn).float().contiguous())
            hpad, hlens = self.enc(enh, lens_list(input_sizes))
            return self.dec.calculate_all_attentions(hpad, hlens, self._split_targets(targets, target_sizes))
