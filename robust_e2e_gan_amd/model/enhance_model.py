"""Mask-based enhancement net (mirror of EnhanceModel, model/enhance_model.py:43-303: blstm / blstmp / U-Net variants)."""
import logging
import sys

import numpy as np
import torch

from .. import lib, ops
from ..lib import Re2eError
from .e2e_common import ConvParams, LinearParams, ModelBase, lecun_normal_init_parameters, lens_dev, lens_list, to_cuda
from .e2e_encoder import BLSTM, BLSTMP
from .gan_model import BatchNormParams, InstanceNormParams, _norm, init_net


class SequenceWise(torch.nn.Module):
    """enhance_model.py:19-40 (parameter container; the (B*T, H) collapse is implicit in the GEMM)."""

    def __init__(self, module):
        super(SequenceWise, self).__init__()
        self.module = module


class ConvTransposeParams(torch.nn.Module):
    """Parameter holder with nn.ConvTranspose2d's names/shapes: weight (Cin, Cout, kh, kw), bias (Cout)."""

    def __init__(self, cin, cout, k, bias=True):
        super().__init__()
        self.weight = torch.nn.Parameter(torch.empty(cin, cout, k, k).normal_(0.0, 0.02))
        self.bias = torch.nn.Parameter(torch.zeros(cout)) if bias else None


class _Mod(torch.nn.Module):
    """parameter-free member of a block's nn.Sequential (LeakyReLU / ReLU / Sigmoid / Dropout): keeps upstream's numbering"""

    def __init__(self, kind, p=0.0):
        super().__init__()
        self.kind, self.p = kind, p


class UnetSkipConnectionBlock(torch.nn.Module):
    """enhance_model.py:249-303 with norm_layer = BatchNorm2d (use_bias False) or InstanceNorm2d(affine=False) (use_bias True, :258-261:
    --enhance_norm instance).  ``self.model`` is an nn.Sequential with
    upstream's member order, so state_dict keys agree (model.1.weight = downconv, model.2.* = downnorm, ...).  NHWC inside;
    the skip connection concatenates along the channel (= last) axis."""

    def __init__(self, outer_nc, inner_nc, input_nc=None, submodule=None, outermost=False, innermost=False, use_dropout=0.0, norm='batch'):
        super().__init__()
        self.outermost = outermost
        input_nc = outer_nc if input_nc is None else input_nc
        ub = norm == 'instance'
        downconv = ConvParams(input_nc, inner_nc, 4, bias=ub, stride=2, padding=1)
        if outermost:
            model = [downconv, submodule, _Mod('relu'), ConvTransposeParams(inner_nc * 2, outer_nc, 4, bias=True), _Mod('sigmoid')]
        elif innermost:
            model = [_Mod('lrelu'), downconv, _Mod('relu'), ConvTransposeParams(inner_nc, outer_nc, 4, bias=ub), _norm(norm, outer_nc)]
        else:
            model = [_Mod('lrelu'), downconv, _norm(norm, inner_nc), submodule, _Mod('relu'),
                     ConvTransposeParams(inner_nc * 2, outer_nc, 4, bias=ub), _norm(norm, outer_nc)]
            if use_dropout > 0.0:
                model.append(_Mod('dropout', use_dropout))
        self.model = torch.nn.Sequential(*model)

    def forward(self, x):
        h = x
        for m in self.model:
            if isinstance(m, _Mod):
                if m.kind == 'dropout':
                    h = ops.dropout(h, m.p) if self.training else h
                else:
                    h = ops.activation(h, m.kind)
            elif isinstance(m, ConvParams):
                h = ops.conv2d(h, m.weight, m.bias, m.stride, m.padding, None)
            elif isinstance(m, ConvTransposeParams):
                h = ops.conv_transpose2d(h, m.weight, m.bias, 2, 1)
            elif isinstance(m, BatchNormParams):
                h = ops.bn_lrelu(h, m.weight, m.bias, m.running_mean, m.running_var, self.training, m.momentum, m.eps, slope=1.0)
                if self.training:
                    m.num_batches_tracked += 1
            elif isinstance(m, InstanceNormParams):
                h = ops.instance_norm_lrelu(h, m.eps, slope=1.0)
            else:
                h = m(h)
        return h if self.outermost else torch.cat([x, h], -1)


class UnetGenerator(torch.nn.Module):
    """enhance_model.py:224-246: ``num_downs`` stride-2 stages (5 = unet_128, 8 = unet_256)."""

    def __init__(self, input_nc, output_nc, num_downs, ngf=64, use_dropout=0.0, norm='batch'):
        super().__init__()
        block = UnetSkipConnectionBlock(ngf * 8, ngf * 8, innermost=True, norm=norm)
        for _ in range(num_downs - 5):
            block = UnetSkipConnectionBlock(ngf * 8, ngf * 8, submodule=block, use_dropout=use_dropout, norm=norm)
        block = UnetSkipConnectionBlock(ngf * 4, ngf * 8, submodule=block, norm=norm)
        block = UnetSkipConnectionBlock(ngf * 2, ngf * 4, submodule=block, norm=norm)
        block = UnetSkipConnectionBlock(ngf, ngf * 2, submodule=block, norm=norm)
        self.model = UnetSkipConnectionBlock(output_nc, ngf, input_nc=input_nc, submodule=block, outermost=True, norm=norm)
        self.num_downs = num_downs

    def forward(self, x_nhwc, ilens):
        return self.model(x_nhwc), ilens


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
        elif self.enhance_type in ('unet_128', 'unet_256'):
            # enhance_model.py:58-63,94-101: pix2pix U-Net over the (T, F) log-spectrogram image
            unorm = getattr(args, 'enhance_norm', 'batch')
            if unorm not in ('batch', 'instance'):
                raise Re2eError('enhance_norm=%s: normalization layer is not usable (e2e_common.get_norm_layer: batch | instance)' % unorm)
            if getattr(args, 'enhance_input_nc', 1) != 1 or getattr(args, 'enhance_output_nc', 1) != 1:
                raise Re2eError('the U-Net enhancer maps ONE log-spectrogram image to ONE mask (enhance_model.py:153 squeezes channel 1)')
            self.enc1 = UnetGenerator(1, 1, 5 if self.enhance_type == 'unet_128' else 8, getattr(args, 'enhance_ngf', 64), args.dropout_rate, norm=unorm)
            init_net(self.enc1, 0.02)
        elif self.enhance_type in ('vggblstmp', 'vggblstm'):
            raise Re2eError('enhance_type %s is dead code upstream (enhance_model.py:94,100 reference an undefined name)' % self.enhance_type)
        else:
            logging.error('Error: need to specify an appropriate enhance_mask archtecture')
            sys.exit()
        self.fc = torch.nn.Sequential(SequenceWise(torch.nn.Sequential(LinearParams(args.enhance_projs, odim, bias=False))))
        self.loss_kind = 'L1'        # the active loss upstream is L1 (:170); 'L2' = the commented-out MSE (:169)
        lecun_normal_init_parameters(self)

    def _unet_mask(self, mix_inputs, mix_log_inputs, ilens):
        """enhance_model.py:139-142,152-164 for the U-Net types: the caller passes mix_log as (B,1,T,F) (enhance_fbank_train.py
        :110-117 trims T and unsqueezes); T and F must be multiples of 2^num_downs.  sigmoid is applied to the U-Net's output,
        which already went through the outermost block's own nn.Sigmoid -- upstream's double sigmoid is kept."""
        x = mix_log_inputs
        if x.dim() == 4:
            if x.shape[1] != 1:
                raise Re2eError('U-Net enhancer input must be (B,1,T,F)')
            x = x[:, 0]
        B, T, F_ = x.shape
        q = 1 << self.enc1.num_downs
        if T % q or F_ % q:
            raise Re2eError('U-Net enhancer: T=%d and F=%d must be multiples of %d (the skip connections need matching sizes)' % (T, F_, q))
        y, _ = self.enc1(x.contiguous().view(B, T, F_, 1), ilens)         # NCHW (B,1,T,F) == NHWC (B,T,F,1)
        out = ops.activation(y.view(B, T, F_), 'sigmoid')
        out = ops.mask_rows(out, lens_dev(lens_list(ilens), x.device))
        return ops.mul_const(out, mix_inputs.contiguous()), out

    def _mask_net(self, mix_inputs, mix_log_inputs, ilens):
        if self.enhance_type in ('unet_128', 'unet_256'):
            return self._unet_mask(mix_inputs, mix_log_inputs, ilens)
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
