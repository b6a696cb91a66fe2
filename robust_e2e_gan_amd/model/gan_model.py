"""Discriminator + GAN loss (mirror of model/gan_model.py:55-171) and CORAL (S2, missing upstream)."""
import torch

from .. import lib, ops
from ..lib import Re2eError
from .e2e_common import ConvParams, ModelBase, to_cuda


class BatchNormParams(torch.nn.Module):
    """nn.BatchNorm2d's state: weight, bias, running_mean, running_var, num_batches_tracked."""

    def __init__(self, c, momentum=0.1, eps=1e-5):
        super().__init__()
        self.weight = torch.nn.Parameter(torch.ones(c))
        self.bias = torch.nn.Parameter(torch.zeros(c))
        self.register_buffer('running_mean', torch.zeros(c))
        self.register_buffer('running_var', torch.ones(c))
        self.register_buffer('num_batches_tracked', torch.tensor(0, dtype=torch.long))
        self.momentum, self.eps = momentum, eps


class InstanceNormParams(torch.nn.Module):
    """nn.InstanceNorm2d(affine=False, track_running_stats=False) (gan_model.py:46): no parameters, no buffers -- a slot in the
    Sequential numbering and the eps."""

    def __init__(self, num_features, eps=1e-5):
        super(InstanceNormParams, self).__init__()
        self.num_features, self.eps = num_features, eps


def _norm(kind, ch):
    return BatchNormParams(ch) if kind == 'batch' else InstanceNormParams(ch)


class _Act(torch.nn.Module):
    """placeholder that keeps nn.Sequential's numbering (LeakyReLU is fused into the conv / BN kernels)."""


class _Sigmoid(torch.nn.Module):
    """nn.Sigmoid() appended by ``use_sigmoid`` (--no_lsgan): fused into the last convolution's epilogue."""


def init_NLayerDiscriminator(input_nc, ndf=64, n_layers=3, use_sigmoid=False, norm='batch'):
    """gan_model.py:55-95; norm_layer = BatchNorm2d (use_bias False on the normalised convs) or InstanceNorm2d (use_bias True, :57)."""
    ub = norm == 'instance'
    seq = [ConvParams(input_nc, ndf, 4, stride=2, padding=1), _Act()]
    nf_mult = 1
    for n in range(1, n_layers):
        nf_prev, nf_mult = nf_mult, min(2 ** n, 8)
        seq += [ConvParams(ndf * nf_prev, ndf * nf_mult, 4, bias=ub, stride=2, padding=1), _norm(norm, ndf * nf_mult), _Act()]
    nf_prev, nf_mult = nf_mult, min(2 ** n_layers, 8)
    seq += [ConvParams(ndf * nf_prev, ndf * nf_mult, 4, bias=ub, stride=1, padding=1), _norm(norm, ndf * nf_mult), _Act()]
    seq += [ConvParams(ndf * nf_mult, 1, 4, stride=1, padding=1)]
    if use_sigmoid:
        seq += [_Sigmoid()]
    return torch.nn.Sequential(*seq)


def init_PixelDiscriminator(input_nc, ndf=64, use_sigmoid=False, norm='batch'):
    """gan_model.py:98-116 (1x1 PatchGAN): with BatchNorm2d only the first conv has a bias, with InstanceNorm2d all three (:100)."""
    ub = norm == 'instance'
    seq = [ConvParams(input_nc, ndf, 1, stride=1, padding=0), _Act(),
           ConvParams(ndf, ndf * 2, 1, bias=ub, stride=1, padding=0), _norm(norm, ndf * 2), _Act(),
           ConvParams(ndf * 2, 1, 1, bias=ub, stride=1, padding=0)]
    if use_sigmoid:
        seq += [_Sigmoid()]
    return torch.nn.Sequential(*seq)


def init_net(net, gain=0.02):
    """gan_model.py:18-39 ('normal'): conv weights ~ N(0, gain), biases 0, BN gamma ~ N(1, gain)."""
    for m in net.modules():
        if isinstance(m, ConvParams):
            m.weight.data.normal_(0.0, gain)
            if m.bias is not None:
                m.bias.data.zero_()
        elif isinstance(m, BatchNormParams):
            m.weight.data.normal_(1.0, gain)
            m.bias.data.zero_()


class GANModel(ModelBase):
    def __init__(self, args):
        super(GANModel, self).__init__()
        self.opt = args
        norm = getattr(args, 'norm_D', 'batch')
        if norm not in ('batch', 'instance'):
            # gan_model.py:46-49: 'none' gives norm_layer = None, which upstream's constructors then call (TypeError); anything else
            # raises NotImplementedError there
            raise Re2eError('norm_D=%s: normalization layer is not usable (gan_model.py:42-49: batch | instance)' % norm)
        use_sigmoid = bool(args.no_lsgan)                   # gan_model.py:126: the plain-GAN discriminator ends in a Sigmoid
        if args.netD_type == 'basic':
            self.model = init_NLayerDiscriminator(args.input_nc, args.ndf, n_layers=3, use_sigmoid=use_sigmoid, norm=norm)
        elif args.netD_type == 'n_layers':
            self.model = init_NLayerDiscriminator(args.input_nc, args.ndf, args.n_layers_D, use_sigmoid=use_sigmoid, norm=norm)
        elif args.netD_type == 'pixel':
            # the module itself is built; upstream's joint loop cannot drive it (joint_train.py:178 reads an undefined
            # mix_feat at train time and applies the 80-wide CMVN to the 160-wide concatenation)
            self.model = init_PixelDiscriminator(args.input_nc, args.ndf, use_sigmoid=use_sigmoid, norm=norm)
        else:
            raise NotImplementedError('Discriminator model name [%s] is not recognized' % args.netD_type)
        init_net(self.model, 0.02)

    def forward(self, input, cmvn=None):
        """(B,T,80) [-> CMVN, S3] -> PatchGAN logits (B,1,H',W')   (gan_model.py:141-145)"""
        x = to_cuda(self, input)
        if cmvn is not None:
            x = ops.cmvn_pair(x, None, to_cuda(self, cmvn).float().contiguous())
        if x.dim() == 3:
            B, T, Fd = x.shape
            h = x.contiguous().view(B, T, Fd, 1)          # NCHW (B,1,T,F) and NHWC (B,T,F,1) share memory
        else:
            h = x.permute(0, 2, 3, 1).contiguous()
        mods = list(self.model)
        self._bn_layers_last = []
        i = 0
        while i < len(mods):
            m = mods[i]
            if isinstance(m, ConvParams):
                nxt = mods[i + 1] if i + 1 < len(mods) else None
                if isinstance(nxt, _Act):
                    h = ops.conv2d(h, m.weight, m.bias, m.stride, m.padding, 'lrelu')
                    i += 2
                elif isinstance(nxt, BatchNormParams):
                    h = ops.conv2d(h, m.weight, m.bias, m.stride, m.padding, None)
                    h = ops.bn_lrelu(h, nxt.weight, nxt.bias, nxt.running_mean, nxt.running_var, self.training, nxt.momentum, nxt.eps)
                    if self.training:
                        nxt.num_batches_tracked += 1
                        self._bn_layers_last.append(nxt)
                    i += 3
                elif isinstance(nxt, InstanceNormParams):
                    h = ops.conv2d(h, m.weight, m.bias, m.stride, m.padding, None)
                    h = ops.instance_norm_lrelu(h, nxt.eps)
                    i += 3
                else:
                    h = ops.conv2d(h, m.weight, m.bias, m.stride, m.padding, 'sigmoid' if isinstance(nxt, _Sigmoid) else None)
                    i += 1
            else:
                i += 1
        return h.permute(0, 3, 1, 2)                      # logical NCHW like the reference (view, Cout == 1)


def replay_running_stats(stats):
    """Apply once more the BatchNorm running-statistics update of a forward pass whose batch statistics were captured in
    ``ops.BN_STATS_SINK`` (momentum update with the unbiased batch variance, as bn_finalize does).  The trainer uses the
    G-step's D(fake) activations again for the D-step instead of recomputing them; upstream runs that forward twice and
    therefore moves the running statistics twice."""
    with torch.no_grad():
        for rm, rv, mean, invstd, P, momentum, eps in stats:
            var_b = (1.0 / (invstd * invstd) - eps).contiguous()
            # the forward's own finalize kernel (csrc/elementwise.hip bn_finalize_kernel): same arithmetic as the update it replays, and the
            # same guard -- batch statistics that are not finite (a recurrence that gave up poisoned the features) move nothing
            junk_m, junk_i = torch.empty_like(mean), torch.empty_like(mean)
            lib.call('re2e_bn_sync_finalize', mean.data_ptr(), var_b.data_ptr(), int(P), mean.numel(), float(momentum), float(eps), rm.data_ptr(),
                     rv.data_ptr(), junk_m.data_ptr(), junk_i.data_ptr())


class GANLoss(torch.nn.Module):
    """LSGAN: MSE(D(x), 1.0 or 0.0 broadcast); ``use_lsgan=False`` (--no_lsgan): nn.BCELoss on the discriminator's
    sigmoid outputs against the same broadcast constants (gan_model.py:152-171)."""

    def __init__(self, use_lsgan=True, target_real_label=1.0, target_fake_label=0.0):
        super(GANLoss, self).__init__()
        self.register_buffer('real_label', torch.tensor(target_real_label))
        self.register_buffer('fake_label', torch.tensor(target_fake_label))
        self._real, self._fake = float(target_real_label), float(target_fake_label)
        self._kind = lib.LOSS_L2 if use_lsgan else lib.LOSS_BCE

    def __call__(self, input, target_is_real):
        return ops.mean_loss(input, None, self._real if target_is_real else self._fake, self._kind)


def CORAL(source, target):
    """S2 (build-defined): Deep-CORAL ||Cov(src) - Cov(tgt)||_F^2 / (4 d^2) over valid-frame rows."""
    return ops.coral(source, target)
