"""Mask-based enhancement net (mirror of EnhanceModel, model/enhance_model.py:43-217; blstm variant)."""
import logging
import sys

import numpy as np
import torch

from .. import lib, ops
from ..lib import Re2eError
from .e2e_common import LinearParams, ModelBase, lecun_normal_init_parameters, lens_dev, lens_list, to_cuda
from .e2e_encoder import BLSTM, BLSTMP


class SequenceWise(torch.nn.Module):
    """enhance_model.py:19-40 (parameter container; the (B*T, H) collapse is implicit in the GEMM)."""

    def __init__(self, module):
        super(SequenceWise, self).__init__()
        self.module = module


class EnhanceModel(ModelBase):
    def __init__(self, args):
        super(EnhanceModel, self).__init__()
        self.opt = args
        idim = odim = args.idim
        self.enhance_type = args.enhance_type
        self.verbose = args.verbose
        self.subsample = np.ones(args.enhance_layers + 1, dtype=int)
        if self.enhance_type == 'blstm':
            self.enc1 = BLSTM(idim, args.enhance_layers, args.enhance_units, args.enhance_projs, args.dropout_rate)
        elif self.enhance_type == 'blstmp':
            ss = args.subsample.split('_')
            for j in range(min(args.enhance_layers + 1, len(ss))):
                self.subsample[j] = int(ss[j])
            if any(int(v) > 1 for v in self.subsample):
                raise Re2eError('a mask enhancer cannot subsample frames: the mask has to cover every frame of mix_inputs '
                                '(upstream fails on the shape mismatch at enhance_model.py:164)')
            self.enc1 = BLSTMP(idim, args.enhance_layers, args.enhance_units, args.enhance_projs, self.subsample, args.subsample_type,
                               args.dropout_rate)
        elif self.enhance_type in ('unet_128', 'unet_256', 'vggblstmp', 'vggblstm'):
            raise Re2eError('enhance_type %s is out of scope (U-Net: "next" row N4; vgg*: dead code upstream, '
                            'enhance_model.py:94,100 reference an undefined name)' % self.enhance_type)
        else:
            logging.error('Error: need to specify an appropriate enhance_mask archtecture')
            sys.exit()
        self.fc = torch.nn.Sequential(SequenceWise(torch.nn.Sequential(LinearParams(args.enhance_projs, odim, bias=False))))
        self.loss_kind = 'L1'        # the active loss upstream is L1 (:170); 'L2' = the commented-out MSE (:169)
        lecun_normal_init_parameters(self)

    def _mask_net(self, mix_inputs, mix_log_inputs, ilens):
        lens = lens_list(ilens)
        T = max(lens)
        dev = mix_log_inputs.device
        ld = lens_dev(lens, dev)
        x_tm = ops.transpose01(mix_log_inputs)[:T]
        proj_tm = self.enc1.forward_tm(x_tm, ld)                        # (T,B,projs)
        proj = ops.transpose01(proj_tm)                                 # (B,T,projs)
        mix = mix_inputs if mix_inputs.shape[1] == T else mix_inputs[:, :T].contiguous()
        W = self.fc[0].module[0].weight
        return ops.mask_fc(proj, W, mix, ld, T)                         # (enhance_out, mask)

    def forward(self, mix_inputs, mix_log_inputs, input_sizes, clean_inputs=None, cos_angles=None):
        """enhance_out = sigmoid(fc(BLSTM(mix_log))) * [t < len] * mix  (+ mask-L1 loss when clean is given)."""
        mix_inputs = to_cuda(self, mix_inputs)
        mix_log_inputs = to_cuda(self, mix_log_inputs)
        enhance_out, _ = self._mask_net(mix_inputs, mix_log_inputs, input_sizes)
        if clean_inputs is None:
            return enhance_out
        clean = to_cuda(self, clean_inputs)
        cos = to_cuda(self, cos_angles)
        tgt = ops.mul_const(clean, cos)
        n = float(sum(lens_list(input_sizes)))
        if self.loss_kind == 'L1':       # F.l1_loss(size_average=False) / sum(ilens)
            loss = ops.mean_loss(enhance_out, tgt, 0.0, lib.LOSS_L1) * (enhance_out.numel() / n)
        else:
            loss = ops.mean_loss(enhance_out, tgt, 0.0, lib.LOSS_L2) * (enhance_out.numel() / n)
        return loss, enhance_out

    def calculate_all_specgram(self, mix_inputs, mix_log_inputs, input_sizes):
        """enhance_model.py:188-217: same as forward but WITHOUT the length mask."""
        with torch.no_grad():
            mix_inputs = to_cuda(self, mix_inputs)
            lens = lens_list(input_sizes)
            full = [mix_inputs.shape[1]] * len(lens)
            out, _ = self._mask_net(mix_inputs, to_cuda(self, mix_log_inputs), input_sizes)
            # frames beyond each length: sigmoid(fc(tanh(l_last.bias))) * mix, recomputed without the mask
            last = self.enc1.l_last if self.enhance_type == 'blstm' else getattr(self.enc1, 'bt%d' % (self.enc1.elayers - 1))
            proj_pad = ops.linear(torch.zeros(1, last.weight.shape[1], device=out.device), last.weight, last.bias, 'tanh')
            padrow = ops.linear(proj_pad, self.fc[0].module[0].weight, None, 'sigmoid')      # (1, F)
            out = out.clone()
            for b, l in enumerate(lens):
                if l < out.shape[1]:
                    out[b, l:] = ops.mul_const(mix_inputs[b, l:out.shape[1]].contiguous(), padrow.expand(out.shape[1] - l, -1).contiguous())
            return out
