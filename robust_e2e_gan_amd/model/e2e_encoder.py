"""Encoder stacks (mirror of model/e2e_encoder.py): VGG2L :231-279, BLSTMP :101-150, BLSTM :153-177.

Internal layouts are MI355X-first: NHWC through the conv stack, time-major (T,B,F) through the
recurrent stack; module boundaries stay batch-first (B,T,F) like the reference."""
import logging
import math
import os
import sys

import torch

from .. import ops
from .. import lib
from ..lib import Re2eError
from .e2e_common import ConvParams, LinearParams, LSTMParams, _get_vgg2l_odim, lens_dev, lens_list

FUSE_RELU_POOL_BWD = lib.exp_env('RE2E_NO_RELU_POOL_FUSION') is None      # A/B switches (ops.conv2d relu_bwd_in_pool / relu_bwd_in_next)
FUSE_RELU_CONV_BWD = lib.exp_env('RE2E_NO_RELU_CONV_FUSION') is None
FUSE_CONV_POOL = lib.exp_env('RE2E_NO_CONV_POOL_FUSION') is None


class BLSTM(torch.nn.Module):
    """nn.LSTM(idim, cdim, elayers, bidirectional) + tanh(Linear(2*cdim, hdim))."""

    def __init__(self, idim, elayers, cdim, hdim, dropout):
        super(BLSTM, self).__init__()
        self.dropout = float(dropout or 0.0)
        self.nblstm = LSTMParams(idim, cdim, elayers)
        self.l_last = LinearParams(cdim * 2, hdim)
        self.elayers = elayers

    def forward_tm(self, x_tm, lens_d):
        """time-major (T,B,I) -> (T,B,hdim)"""
        for l in range(self.elayers):
            x_tm = ops.bilstm(x_tm, lens_d, self.nblstm.layer_weights(l))
            # nn.LSTM(dropout=p) (e2e_encoder.py:156-157): on the outputs of every layer but the last, training mode only.
            # Upstream draws the mask over the packed (valid) frames; padded rows are zero here, so masking the padded
            # tensor is the same arithmetic on the valid frames.
            if self.dropout and self.training and l + 1 < self.elayers:
                x_tm = ops.dropout(x_tm, self.dropout)
        # (upstream applies it to the padded rows too, :173-176: tanh(bias) there -- here the product runs over the valid rows and the padded
        #  ones get tanh(bias) written, ops.gemm_rows fill)
        return ops.linear(x_tm, self.l_last.weight, self.l_last.bias, 'tanh', maps=ops.row_maps(lens_d, x_tm.shape[0], x_tm.shape[1]))

    def forward(self, xpad, ilens):
        lens = lens_list(ilens)
        T = max(lens)
        x_tm = ops.transpose01(xpad)[:T]
        y = self.forward_tm(x_tm, lens_dev(lens, xpad.device))
        return ops.transpose01(y), lens


class BLSTMP(torch.nn.Module):
    def __init__(self, idim, elayers, cdim, hdim, subsample, subsample_type, dropout):
        super(BLSTMP, self).__init__()
        # every layer is its own nn.LSTM(num_layers=1, dropout=p) upstream (e2e_encoder.py:109): inter-layer dropout of a
        # one-layer LSTM never fires, so ``dropout`` is accepted and has no effect here either.
        for i in range(elayers):
            setattr(self, 'bilstm%d' % i, LSTMParams(idim if i == 0 else hdim, cdim, 1))
            setattr(self, 'bt%d' % i, LinearParams(2 * cdim, hdim))
        self.elayers, self.cdim = elayers, cdim
        self.subsample, self.subsample_type = subsample, subsample_type
        self.subsampling = any(int(s) > 1 for s in subsample[1:elayers + 1])
        if self.subsampling and subsample_type not in ('skip', 'maxpooling'):
            raise Re2eError('BLSTMP subsample_type %r is not known (e2e_encoder.py:137-143: skip | maxpooling)' % subsample_type)
        if self.subsampling and subsample_type == 'maxpooling' and any(int(s) not in (1, 2) for s in subsample[1:elayers + 1]):
            raise Re2eError('BLSTMP maxpooling subsampling is built for factor 2 (the 2x2 pooling kernel over a width-1 image)')

    def forward_tm(self, x_tm, lens_d, lens=None):
        """Time-major (T,B,idim) -> (T',B,hdim).  With frame subsampling configured (``subsample[l+1] > 1``: every sub-th
        frame of layer l's output is kept, lengths become (len+1)//sub, e2e_encoder.py:133-139) the host length list
        ``lens`` is required and the result is ``(y, new_lens)``; without it the lengths do not change and ``y`` is returned."""
        if self.subsampling and lens is None:
            raise Re2eError('BLSTMP with frame subsampling needs the host length list (forward_tm(x, lens_dev, lens))')
        cur = list(lens) if lens is not None else None
        for l in range(self.elayers):
            y = ops.bilstm(x_tm, lens_d, getattr(self, 'bilstm%d' % l).layer_weights(0))
            sub = int(self.subsample[l + 1])
            if sub > 1 and self.subsample_type == 'skip':
                y = y[::sub].contiguous()
                cur = [(i + 1) // sub for i in cur]
                lens_d = lens_dev(cur, y.device)
            elif sub > 1:
                # F.max_pool1d(kernel 2, stride 2) over time (:140-143; floor mode: an odd last frame is dropped).  The
                # time-major (T, B*2H) tensor is a width-1 NHWC image with B*2H channels for the 2x2 pooling kernel.
                T2, Bc, Cc = y.shape[0] // 2, y.shape[1], y.shape[2]
                y = ops.maxpool2(y[:2 * T2].contiguous().view(1, 2 * T2, 1, Bc * Cc)).view(T2, Bc, Cc)
                cur = [i // sub for i in cur]
                lens_d = lens_dev(cur, y.device)
            bt = getattr(self, 'bt%d' % l)
            # (upstream applies it to the padded rows too, :145-147: tanh(bias) there -- the mapped product writes the same, ops.gemm_rows fill)
            x_tm = ops.linear(y, bt.weight, bt.bias, 'tanh', maps=ops.row_maps(lens_d, y.shape[0], y.shape[1]))
        return (x_tm, cur) if lens is not None else x_tm

    def forward(self, xpad, ilens):
        lens = lens_list(ilens)
        T = max(lens)
        y, nl = self.forward_tm(ops.transpose01(xpad)[:T], lens_dev(lens, xpad.device), lens)
        return ops.transpose01(y), nl


ROW_LIMITS = lib.exp_env('RE2E_NO_VGG_ROW_LIMITS') is None      # (experiments) every row of every image, as rounds 1-5 did


class VGG2L(torch.nn.Module):
    def __init__(self, in_channel=1):
        super(VGG2L, self).__init__()
        self.conv1_1 = ConvParams(in_channel, 64, 3, stride=1, padding=1)
        self.conv1_2 = ConvParams(64, 64, 3, stride=1, padding=1)
        self.conv2_1 = ConvParams(64, 128, 3, stride=1, padding=1)
        self.conv2_2 = ConvParams(128, 128, 3, stride=1, padding=1)
        self.in_channel = in_channel

    def row_limits(self, ilens, T, device):
        """Per-layer row limits of a ragged batch for ``conv_stack`` -> (lims of conv1_2, conv2_1, conv2_2) or None.

        Upstream convolves the whole zero-padded (B, 1, Tmax, idim) batch and cuts every utterance at its pooled length afterwards (:272-278):
        a pooled frame p reads input rows < 4 p + 10, so rows further than the stack's reach beyond an utterance's end are computed and never read.
        With P pooled frames, conv2_2 is needed in rows < 2 P, conv2_1 in < 2 P + 1, pool1 in < 2 P + 2, conv1_2 in < 4 P + 4, conv1_1 in < 4 P + 5,
        and the gradients are exactly zero beyond the same rows.  Every limit is its layer's need rounded up to 16 (the kernels skip whole 8- /
        16-row patches).  A layer's rows between its need and its limit are computed from whatever the layer in front holds there -- real values
        below that layer's limit, zeros beyond it (ops.RowLims: zero-filled wherever a kernel reads all rows) -- finite, unneeded, multiplied
        by zero gradients.  Config 4 (lengths 0.7 .. 1.0 Tmax): 12-14 % of the three Winograd layers' rows, forward, data and weight gradient."""
        lens = lens_list(ilens)
        H1, H2 = T, (T + 1) // 2
        a16 = lambda v: (v + 15) // 16 * 16
        P = self.pooled_lens(lens)
        L22 = [min(H2, a16(2 * p)) for p in P]
        L21 = [min(H2, a16(2 * p + 1)) for p in P]
        L12 = [min(H1, a16(4 * p + 4)) for p in P]
        L11 = [min(H1, a16(4 * p + 5)) for p in P]
        Lp1 = [(l + 1) // 2 for l in L12]                       # pooled rows of conv1_2's output = conv2_1's input
        saved = sum(H1 - l for l in L12) / float(max(1, len(lens)) * H1)
        if saved < 0.03:
            return None
        dv = lambda v: lens_dev(v, device)
        tail = lambda v, H: max(H - min(v), 0)
        return (ops.RowLims(dv(L12), dv(L11), tail(L12, H1), tail(L11, H1)), ops.RowLims(dv(L21), dv(Lp1), tail(L21, H2), tail(Lp1, H2)),
                ops.RowLims(dv(L22), dv(L21), tail(L22, H2), tail(L21, H2)))

    def conv_stack(self, xs, ilens=None):
        """(B,T,idim) batch-first -> pooled NHWC (B, ceil(ceil(T/2)/2), ceil(ceil(idim/2)/2), 128)  (e2e_encoder.py:259-266).
        ``ilens``: the utterance lengths -- rows the cut + re-pad of :272-278 never reads are then not computed (``row_limits``)."""
        if self.in_channel != 1:
            raise Re2eError('VGG2L with in_channel != 1 is not on the hot path')
        B, T, Fd = xs.shape
        lm = self.row_limits(ilens, T, xs.device) if (ilens is not None and ROW_LIMITS and FUSE_CONV_POOL) else None
        l12, l21, l22 = lm if lm is not None else (None, None, None)
        h = xs.contiguous().view(B, T, Fd, 1)                                     # NCHW (B,1,T,F) == NHWC (B,T,F,1)
        # No ReLU-backward pass anywhere in the stack: conv -> ReLU -> conv takes the data gradient of the second convolution through
        # the ReLU in that kernel's epilogue (x_is_relu_out), conv -> ReLU -> pool folds the ReLU's mask into the pool's index
        # byte (relu_in).  RE2E_NO_RELU_POOL_FUSION / RE2E_NO_RELU_CONV_FUSION restore the separate passes.
        fp, fc = FUSE_RELU_POOL_BWD, FUSE_RELU_CONV_BWD
        h = ops.conv2d(h, self.conv1_1.weight, self.conv1_1.bias, 1, 1, 'relu', relu_bwd_in_next=fc)
        if FUSE_CONV_POOL:          # conv -> ReLU -> pool in one launch: the full-resolution activation is never written
            h = ops.conv2d(h, self.conv1_2.weight, self.conv1_2.bias, 1, 1, 'relu', x_is_relu_out=fc, pool=True, lims=l12)
            h = ops.conv2d(h, self.conv2_1.weight, self.conv2_1.bias, 1, 1, 'relu', relu_bwd_in_next=fc, lims=l21)
            return ops.conv2d(h, self.conv2_2.weight, self.conv2_2.bias, 1, 1, 'relu', x_is_relu_out=fc, pool=True, lims=l22)
        h = ops.conv2d(h, self.conv1_2.weight, self.conv1_2.bias, 1, 1, 'relu', relu_bwd_in_pool=fp, x_is_relu_out=fc)
        h = ops.maxpool2(h, relu_in=fp)
        h = ops.conv2d(h, self.conv2_1.weight, self.conv2_1.bias, 1, 1, 'relu', relu_bwd_in_next=fc)
        h = ops.conv2d(h, self.conv2_2.weight, self.conv2_2.bias, 1, 1, 'relu', relu_bwd_in_pool=fp, x_is_relu_out=fc)
        return ops.maxpool2(h, relu_in=fp)

    @staticmethod
    def pooled_lens(ilens):
        return [int(math.ceil(math.ceil(l / 2.0) / 2.0)) for l in lens_list(ilens)]

    def pack_tm(self, hs, ilens_per_branch):
        """pooled NHWC branches -> ONE time-major (T', sum B_k, 128*F') tensor with the cut + zero re-pad of :272-278."""
        nls = [self.pooled_lens(il) for il in ilens_per_branch]
        lens_d = [lens_dev(nl, hs[0].device) for nl in nls]
        out = ops.vgg_pack_multi(hs, lens_d)
        nl = sum(nls, [])
        # (a collated batch is as long as its longest utterance: no slice then -- a slice's backward is a zero-fill + copy of the whole gradient)
        return (out if max(nl) == out.shape[0] else out[:max(nl)]), nl

    def forward_tm(self, xs, ilens):
        """(B,T,idim) batch-first -> time-major (T',B,128*F')."""
        return self.pack_tm([self.conv_stack(xs, ilens)], [ilens])

    def forward(self, xs, ilens):
        y, nl = self.forward_tm(xs, ilens)
        return ops.transpose01(y), nl


class Encoder(torch.nn.Module):
    """model/e2e_encoder.py:17-98 dispatcher (blstm / blstmp / vggblstm / vggblstmp)."""

    def __init__(self, etype, idim, elayers, eunits, eprojs, subsample, subsample_type, dropout, in_channel=1):
        super(Encoder, self).__init__()
        if etype == 'blstm':
            self.enc1 = BLSTM(idim, elayers, eunits, eprojs, dropout)
        elif etype == 'blstmp':
            self.enc1 = BLSTMP(idim, elayers, eunits, eprojs, subsample, subsample_type, dropout)
        elif etype == 'vggblstmp':
            self.enc1 = VGG2L(in_channel)
            self.enc2 = BLSTMP(_get_vgg2l_odim(idim, in_channel=in_channel), elayers, eunits, eprojs, subsample, subsample_type, dropout)
        elif etype == 'vggblstm':
            self.enc1 = VGG2L(in_channel)
            self.enc2 = BLSTM(_get_vgg2l_odim(idim, in_channel=in_channel), elayers, eunits, eprojs, dropout)
        elif etype in ('cnnblstmp', 'cnnblstm'):
            raise Re2eError('etype %s (CNN2L) is out of scope for the hot path (SURVEY section 2, row 5)' % etype)
        else:
            logging.error('Error: need to specify an appropriate encoder archtecture')
            sys.exit()
        self.etype = etype

    def forward_tm(self, xs, ilens):
        """-> time-major (T',B,eprojs), lens'"""
        if self.etype in ('blstm', 'blstmp'):
            lens = lens_list(ilens)
            x_tm = ops.transpose01(xs)[:max(lens)]
            if self.etype == 'blstmp':
                return self.enc1.forward_tm(x_tm, lens_dev(lens, xs.device), lens)
            return self.enc1.forward_tm(x_tm, lens_dev(lens, xs.device)), lens
        h_tm, nl = self.enc1.forward_tm(xs, ilens)
        if self.etype == 'vggblstmp':
            return self.enc2.forward_tm(h_tm, lens_dev(nl, xs.device), nl)
        return self.enc2.forward_tm(h_tm, lens_dev(nl, xs.device)), nl

    def forward(self, xs, ilens):
        y, lens = self.forward_tm(xs, ilens)
        return ops.transpose01(y), lens
