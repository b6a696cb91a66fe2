"""Attention decoder (mirror of Decoder.__init__/forward, model/e2e_decoder.py:27-168)."""
import random

import numpy as np
import torch

from .. import ops
from ..lib import Re2eError
from .e2e_common import LinearParams, host_to_dev, lens_dev, lens_list


class LSTMCellParams(torch.nn.Module):
    """nn.LSTMCell's parameter names: weight_ih (4H, I), weight_hh (4H, H), bias_ih, bias_hh."""

    def __init__(self, idim, hdim):
        super().__init__()
        b = 1.0 / np.sqrt(hdim)
        self.weight_ih = torch.nn.Parameter(torch.empty(4 * hdim, idim).uniform_(-b, b))
        self.weight_hh = torch.nn.Parameter(torch.empty(4 * hdim, hdim).uniform_(-b, b))
        self.bias_ih = torch.nn.Parameter(torch.empty(4 * hdim).uniform_(-b, b))
        self.bias_hh = torch.nn.Parameter(torch.empty(4 * hdim).uniform_(-b, b))


class EmbeddingParams(torch.nn.Module):
    def __init__(self, n, d):
        super().__init__()
        self.weight = torch.nn.Parameter(torch.randn(n, d))


def decoder_forward_hip(p, hpad, hlens, ys, eos, ss_rate=0.0, return_att=False, prefix='', greedy=False, labeldist=None, lsm_weight=0.0):
    """Decoder pass on the GPU.  ``p`` maps reference state_dict names (att.* / dec.*) to Parameters; ``ys`` is a
    list of 1-D label tensors (host or device).  Teacher forced, except at the steps where scheduled sampling
    fires (e2e_decoder.py:123: ``random.random() < rate and i > 0`` -- one draw of Python's RNG per step, for the
    whole batch) or, with ``greedy`` (calculate_all_attentions :408-412), at every step i > 0."""
    dev = hpad.device
    B, T, E = hpad.shape
    hl = lens_list(hlens)
    hl_dev = lens_dev(hl, dev)
    ylist = [[int(v) for v in y.tolist()] for y in ys]
    L1 = max(len(y) for y in ylist) + 1
    ids_in = np.full((B, L1), eos, np.int32)         # pad_list(ys_in, eos)   :97
    ids_out = np.full((B, L1), -1, np.int32)         # pad_list(ys_out, -1)   :98
    for b, y in enumerate(ylist):
        ids_in[b, 1:len(y) + 1] = y
        ids_out[b, :len(y)] = y
        ids_out[b, len(y)] = eos
    ids_tm = host_to_dev(np.ascontiguousarray(ids_in.T).reshape(-1), dev)
    tgt_tm = host_to_dev(np.ascontiguousarray(ids_out.T).reshape(-1), dev)
    hmask = ops.mask_rows(hpad, hl_dev)                                            # :85
    pre = ops.linear(hmask, p[prefix + 'att.mlp_enc.weight'], p[prefix + 'att.mlp_enc.bias'])
    Pm = dict(embed=p[prefix + 'dec.embed.weight'], w_ih=p[prefix + 'dec.decoder.0.weight_ih'], w_hh=p[prefix + 'dec.decoder.0.weight_hh'],
              b_ih=p[prefix + 'dec.decoder.0.bias_ih'], b_hh=p[prefix + 'dec.decoder.0.bias_hh'], mlp_dec=p[prefix + 'att.mlp_dec.weight'],
              mlp_att=p[prefix + 'att.mlp_att.weight'], loc_conv=p[prefix + 'att.loc_conv.weight'], gvec_w=p[prefix + 'att.gvec.weight'],
              gvec_b=p[prefix + 'att.gvec.bias'])
    if greedy:
        sample_steps = tuple(i > 0 for i in range(L1))
    else:
        sample_steps = tuple((random.random() < ss_rate) and i > 0 for i in range(L1)) if ss_rate > 0.0 else None
    if sample_steps is not None and any(sample_steps):
        Pm.update(out_w=p[prefix + 'dec.output.weight'], out_b=p[prefix + 'dec.output.bias'])
    else:
        sample_steps = None
    ops.mark_grad(pre, 'pre (decoder loop bwd done)')
    ops.mark('decoder loop starts')
    z_all, w_all = ops.decoder_loop(hmask, pre, ids_tm, hl_dev, L1, Pm, sample_steps)   # (L1,B,D), (L1,B,T)
    ops.mark('decoder loop fwd done')
    ops.mark_grad(z_all, 'decoder states (output layer + CE bwd done: loop bwd starts)')
    D = z_all.shape[2]
    logits = ops.linear(z_all.reshape(L1 * B, D), p[prefix + 'dec.output.weight'], p[prefix + 'dec.output.bias'])
    scale = float(np.mean([len(y) + 1 for y in ylist])) - 1.0                      # :159
    loss, stats = ops.cross_entropy(logits, tgt_tm, scale)
    if labeldist is not None:                                                      # :162-166 label smoothing
        reg = ops.label_smoothing(logits, labeldist, B)
        loss = (1.0 - lsm_weight) * loss + lsm_weight * reg
    acc = stats[1] / stats[2]                                                      # th_accuracy (device scalar)
    if return_att:
        return loss.view(()), acc, w_all.transpose(0, 1)
    return loss.view(()), acc


class Decoder(torch.nn.Module):
    def __init__(self, eprojs, odim, dlayers, dunits, sos, eos, att, verbose=0, char_list=None, labeldist=None, lsm_weight=0.,
                 fusion=None, rnnlm=None, model_unit='char', space_loss_weight=0.1):
        super(Decoder, self).__init__()
        if dlayers != 1:
            raise Re2eError('dlayers > 1 is outside the round-1 hot path')
        self.labeldist, self.lsm_weight, self.vlabeldist = labeldist, lsm_weight, None
        self.dunits, self.dlayers = dunits, dlayers
        self.embed = EmbeddingParams(odim, dunits)
        self.decoder = torch.nn.ModuleList([LSTMCellParams(dunits + eprojs, dunits)])
        self.output = LinearParams(dunits, odim)
        self.att = att
        self.sos, self.eos = sos, eos
        self.ignore_id = -1
        self.loss = None
        self.return_acc_tensor = False

    def forward(self, hpad, hlen, ys, scheduled_sampling_rate=0.0, att_params=None):
        p = {'dec.' + k: v for k, v in self.named_parameters() if not k.startswith('att.')}
        p.update({'att.' + k: v for k, v in self.att.named_parameters()})
        if self.labeldist is not None and (self.vlabeldist is None or self.vlabeldist.device != hpad.device):
            self.vlabeldist = torch.as_tensor(np.asarray(self.labeldist), dtype=torch.float32).to(hpad.device).contiguous()
        loss, acc = decoder_forward_hip(p, hpad, hlen, ys, self.eos, scheduled_sampling_rate, labeldist=self.vlabeldist,
                                        lsm_weight=self.lsm_weight)
        self.loss = loss
        return loss, (acc if self.return_acc_tensor else float(acc))

    def recognize_beam(self, h, lpz, recog_args, char_list=None, rnnlm=None, fstlm=None):
        """e2e_decoder.py:171-369 (no LM): n-best list of {'yseq', 'score'} for the encoder states ``h`` (T', eprojs) of one
        utterance; ``lpz`` = CTC log posteriors (T', V) or None.  All live hypotheses are advanced as one GPU batch."""
        if rnnlm is not None or fstlm is not None:
            raise Re2eError('LM fusion at decode time is out of scope (SURVEY section 2)')
        from .beam_search import recognize_beam
        p = {'dec.' + k: v for k, v in self.named_parameters() if not k.startswith('att.')}
        p.update({'att.' + k: v for k, v in self.att.named_parameters()})
        lp = lpz.detach().cpu().numpy() if isinstance(lpz, torch.Tensor) else lpz
        return recognize_beam(p, h, lp, recog_args, self.eos, lpz_dev=lpz if isinstance(lpz, torch.Tensor) and lpz.is_cuda else None)

    def calculate_all_attentions(self, hpad, hlen, ys):
        """e2e_decoder.py:371-461 -- attention weights (B, Lmax+1, T').  NB the reference's pass is GREEDY: for i > 0 it
        feeds the arg-max of its own previous output (:408-412), the labels only fix the number of steps."""
        p = {'dec.' + k: v for k, v in self.named_parameters() if not k.startswith('att.')}
        p.update({'att.' + k: v for k, v in self.att.named_parameters()})
        with torch.no_grad():
            _, _, att = decoder_forward_hip(p, hpad, hlen, ys, self.eos, 0.0, return_att=True, greedy=True)
        return att.cpu().numpy()
