"""Functional CPU restatement of the four networks on the joint_train hot path.

TEST INFRASTRUCTURE (see oracle/__init__.py).  Every function takes a dict ``p`` of tensors
keyed exactly like the reference ``state_dict()`` so golden parameters
load straight in.  All file:line citations are to /root/reference.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F
from torch.nn.utils.rnn import pack_padded_sequence, pad_packed_sequence

from .fbank_tables import mel_matrix  # noqa: F401  (re-export)


# --------------------------------------------------------------------------------------
# helpers (model/e2e_common.py)
# --------------------------------------------------------------------------------------
def pad_list(xs, pad_value):
    """e2e_common.py:208-217"""
    n = len(xs)
    ml = max(x.size(0) for x in xs)
    out = xs[0].new_full((n, ml) + tuple(xs[0].shape[1:]), pad_value)
    for i, x in enumerate(xs):
        out[i, :x.size(0)] = x
    return out


def mask_by_length(xs, lens, fill=0.0):
    """e2e_common.py:190-195"""
    T = xs.size(1)
    m = (torch.arange(T).unsqueeze(0) < torch.as_tensor(lens).view(-1, 1))
    return torch.where(m.unsqueeze(-1), xs, torch.full_like(xs, fill))


def th_accuracy(y_all, pad_target, ignore_label=-1):
    """e2e_common.py:198-205"""
    pred = y_all.view(pad_target.size(0), pad_target.size(1), -1).argmax(2)
    mask = pad_target != ignore_label
    num = (pred[mask] == pad_target[mask]).sum().item()
    return float(num) / float(mask.sum().item())


def _bilstm(x, lens, w, prefix, layer):
    """One bidirectional nn.LSTM layer on a padded batch with packed semantics
    (e2e_encoder.py:128-132,168-170): reverse direction starts at each utterance's last valid
    frame, padded outputs are zero, h0=c0=0.  Gate order i,f,g,o, two bias vectors."""
    sfx = '_l%d' % layer
    H = w[prefix + 'weight_hh' + sfx].size(1)
    flat = [w[prefix + 'weight_ih' + sfx], w[prefix + 'weight_hh' + sfx],
            w[prefix + 'bias_ih' + sfx], w[prefix + 'bias_hh' + sfx],
            w[prefix + 'weight_ih' + sfx + '_reverse'], w[prefix + 'weight_hh' + sfx + '_reverse'],
            w[prefix + 'bias_ih' + sfx + '_reverse'], w[prefix + 'bias_hh' + sfx + '_reverse']]
    packed = pack_padded_sequence(x, torch.as_tensor(lens).cpu(), batch_first=True)
    B = x.size(0)
    h0 = x.new_zeros(2, B, H)
    out, _, _ = torch._VF.lstm(packed.data, packed.batch_sizes, (h0, h0), flat, True, 1, 0.0,
                               False, True)
    y, _ = pad_packed_sequence(type(packed)(out, packed.batch_sizes, None, None), batch_first=True)
    return y


# --------------------------------------------------------------------------------------
# F2  EnhanceModel (blstm)   model/enhance_model.py:125-174, e2e_encoder.py:153-177
# --------------------------------------------------------------------------------------
def tm_mask(mask_flat, B, T, W):
    """A dropout mask drawn over the TIME-MAJOR (T,B,W) tensor (the product's layout, oracle/philox.py) as a batch-first
    (B,T,W) tensor."""
    return torch.as_tensor(mask_flat).view(T, B, W).transpose(0, 1)


def enhance_forward(p, mix, mix_log, lens, layers, clean=None, cos=None, inter_masks=None, kind='blstm'):
    """``inter_masks[l]``: dropout mask (already scaled by 1/(1-p), batch-first) applied to the output of LSTM layer l < layers-1
    -- nn.LSTM(dropout=p) in training mode (e2e_encoder.py:156-157).  ``kind='blstmp'``: the BLSTMP enhancer
    (enhance_model.py:90-93: per-layer projection, no l_last)."""
    if kind == 'blstmp':
        proj3, _ = blstmp_forward(p, mix_log, lens, layers, pre='enc1.')
        B, T, _ = proj3.shape
        return _enhance_head(p, proj3.reshape(B * T, -1), B, T, mix, lens, clean, cos)
    x = mix_log
    for l in range(layers):
        x = _bilstm(x, lens, {k.replace('enc1.nblstm.', ''): v for k, v in p.items()}, '', l)
        if inter_masks is not None and l + 1 < layers:
            x = x * inter_masks[l]
    B, T, _ = x.shape
    proj = torch.tanh(F.linear(x.reshape(B * T, -1), p['enc1.l_last.weight'], p['enc1.l_last.bias']))
    return _enhance_head(p, proj, B, T, mix, lens, clean, cos)


def _enhance_head(p, proj, B, T, mix, lens, clean, cos):
    lin = F.linear(proj, p['fc.0.module.0.weight']).view(B, T, -1)
    out = torch.sigmoid(lin)
    valid = (torch.arange(T).unsqueeze(0) < torch.as_tensor(lens).view(-1, 1)).unsqueeze(-1)
    out = out * valid.to(out.dtype)                      # :158-163 masked_fill(mask, 0)
    enhance_out = out * mix[:, :T]                       # :164
    if clean is not None:                                # :166-172 (active loss is L1)
        loss = (enhance_out - clean * cos).abs().sum() / float(sum(int(l) for l in lens))
        return loss, enhance_out
    return enhance_out


# --------------------------------------------------------------------------------------
# N4  U-Net enhancer   model/enhance_model.py:224-303 (UnetGenerator / UnetSkipConnectionBlock, BatchNorm)
# --------------------------------------------------------------------------------------
def _unet_block(p, buf, pre, x, depth, num_downs, train, drop_masks):
    """One UnetSkipConnectionBlock (enhance_model.py:249-303); ``pre`` = its state_dict prefix ('enc1.model.' for the outermost)."""
    outermost, innermost = depth == 0, depth == num_downs - 1
    bn = lambda h, k: F.batch_norm(h, buf[pre + 'model.%d.running_mean' % k], buf[pre + 'model.%d.running_var' % k],
                                   p[pre + 'model.%d.weight' % k], p[pre + 'model.%d.bias' % k], train, 0.1, 1e-5)
    if outermost:
        h = F.conv2d(x, p[pre + 'model.0.weight'], None, stride=2, padding=1)
        h = _unet_block(p, buf, pre + 'model.1.', h, depth + 1, num_downs, train, drop_masks)
        h = F.conv_transpose2d(F.relu(h), p[pre + 'model.3.weight'], p[pre + 'model.3.bias'], stride=2, padding=1)
        return torch.sigmoid(h)
    h = F.conv2d(F.leaky_relu(x, 0.2), p[pre + 'model.1.weight'], None, stride=2, padding=1)
    if innermost:
        h = F.conv_transpose2d(F.relu(h), p[pre + 'model.3.weight'], None, stride=2, padding=1)
        h = bn(h, 4)
    else:
        h = bn(h, 2)
        h = _unet_block(p, buf, pre + 'model.3.', h, depth + 1, num_downs, train, drop_masks)
        h = F.conv_transpose2d(F.relu(h), p[pre + 'model.5.weight'], None, stride=2, padding=1)
        h = bn(h, 6)
        if drop_masks is not None and (pre + 'model.7') in drop_masks:
            h = h * drop_masks[pre + 'model.7']
    return torch.cat([x, h], 1)


def unet_enhance_forward(p, buf, mix, mix_log, lens, num_downs=5, clean=None, cos=None, train=True, drop_masks=None):
    """EnhanceModel.forward for enhance_type unet_128 / unet_256 (enhance_model.py:139-142,152-172): the U-Net output (already
    through the outermost nn.Sigmoid) is squeezed and passed through sigmoid AGAIN, masked by length and multiplied with mix."""
    y = _unet_block(p, buf, 'enc1.model.', mix_log.unsqueeze(1), 0, num_downs, train, drop_masks)
    out = torch.sigmoid(y.squeeze(1))
    B, T, _ = out.shape
    valid = (torch.arange(T).unsqueeze(0) < torch.as_tensor(lens).view(-1, 1)).unsqueeze(-1)
    enhance_out = out * valid.to(out.dtype) * mix
    if clean is not None:
        loss = (enhance_out - clean * cos).abs().sum() / float(sum(int(l) for l in lens))
        return loss, enhance_out
    return enhance_out


# --------------------------------------------------------------------------------------
# F3/F4  FbankModel   model/feat_model.py:118-135, :62-90
# --------------------------------------------------------------------------------------
def fbank_forward(x, W, cmvn=None):
    n, t = x.size(0), x.size(1)
    y = (x ** 2).reshape(n * t, -1).mm(W).view(n, t, -1)
    keep = ~(y <= 1e-7)                                 # :130 in-place clamp `out[out <= 1e-7] = 1e-7` => zero grad there; a NaN is NOT replaced
    y = torch.where(keep, y, torch.full_like(y, 1e-7))
    y = torch.log(y)
    if cmvn is not None:
        y = (y + cmvn[0, :]) * cmvn[1, :]
    return y


class CmvnAccumulator:
    """feat_model.py:62-90 -- host-side running sums over valid frames; returns None until
    ``cmvn_num`` utterances were seen, then [-mean; 1/sqrt(var)] on the NEXT call."""

    def __init__(self, dim, cmvn_num):
        self.sum = np.zeros((1, dim), np.float32)
        self.sum_sq = np.zeros((1, dim), np.float32)
        self.cmvn_num = cmvn_num
        self.n_done = 0
        self.frames = 0

    def update(self, feats, lens):
        if self.n_done < self.cmvn_num:
            for b in range(len(lens)):
                m = feats[b, :int(lens[b])].detach().cpu().numpy()
                self.sum = np.add(self.sum, m.sum(0))
                self.sum_sq = np.add(self.sum_sq, np.square(m).sum(0))
                self.frames += m.shape[0]
                self.n_done += 1
            return None
        mean = self.sum / self.frames
        var = self.sum_sq / self.frames - np.square(mean)
        out = np.zeros((2, mean.shape[1]), np.float32)
        out[0] = -mean
        out[1] = 1.0 / np.sqrt(var)
        return out


# --------------------------------------------------------------------------------------
# F5  VGG2L   model/e2e_encoder.py:242-279
# --------------------------------------------------------------------------------------
def vgg2l_forward(p, x, lens, pre='enc.enc1.'):
    h = x.unsqueeze(1)
    h = F.relu(F.conv2d(h, p[pre + 'conv1_1.weight'], p[pre + 'conv1_1.bias'], padding=1))
    h = F.relu(F.conv2d(h, p[pre + 'conv1_2.weight'], p[pre + 'conv1_2.bias'], padding=1))
    h = F.max_pool2d(h, 2, stride=2, ceil_mode=True)
    h = F.relu(F.conv2d(h, p[pre + 'conv2_1.weight'], p[pre + 'conv2_1.bias'], padding=1))
    h = F.relu(F.conv2d(h, p[pre + 'conv2_2.weight'], p[pre + 'conv2_2.bias'], padding=1))
    h = F.max_pool2d(h, 2, stride=2, ceil_mode=True)
    nl = [int(math.ceil(math.ceil(int(l) / 2.0) / 2.0)) for l in lens]
    h = h.transpose(1, 2).contiguous()
    h = h.view(h.size(0), h.size(1), h.size(2) * h.size(3))
    h = pad_list([h[i, :nl[i]] for i in range(len(nl))], 0.0)      # cut + re-pad with zeros
    return h, nl


# --------------------------------------------------------------------------------------
# F6  BLSTMP (subsample all 1)   model/e2e_encoder.py:119-150
# --------------------------------------------------------------------------------------
def blstmp_forward(p, x, lens, elayers, pre='enc.enc2.', subsample=None, subsample_type='skip'):
    """BLSTMP.forward (model/e2e_encoder.py:119-150); ``subsample[l+1] > 1`` keeps every sub-th output frame of layer l
    ('skip' type, :137-139)."""
    lens = [int(l) for l in lens]
    for l in range(elayers):
        w = {k.replace(pre + 'bilstm%d.' % l, ''): v for k, v in p.items()
             if k.startswith(pre + 'bilstm%d.' % l)}
        y = _bilstm(x, lens, w, '', 0)
        sub = int(subsample[l + 1]) if subsample is not None else 1
        if sub > 1 and subsample_type == 'skip':
            y = y[:, ::sub]
            lens = [(i + 1) // sub for i in lens]
        elif sub > 1:                                   # 'maxpooling' (:140-143)
            y = F.max_pool1d(y.transpose(1, 2), sub, stride=sub).transpose(1, 2)
            lens = [i // sub for i in lens]
        B, T, _ = y.shape
        x = torch.tanh(F.linear(y.reshape(B * T, -1), p[pre + 'bt%d.weight' % l],
                                p[pre + 'bt%d.bias' % l])).view(B, T, -1)
    return x, lens


def encoder_forward(p, x, lens, elayers):
    h, hl = vgg2l_forward(p, x, lens)
    return blstmp_forward(p, h, hl, elayers)


# --------------------------------------------------------------------------------------
# F7  CTC   model/e2e_ctc.py:33-66  (warp-ctc restated: softmax inside, blank 0, sum/B)
# --------------------------------------------------------------------------------------
def ctc_forward(p, hpad, hlens, ys, dropout_mask=None):
    """``dropout_mask``: (B,T',E) mask scaled by 1/(1-p) -- F.dropout(hs_pad, p) of e2e_ctc.py:51 (always on)."""
    if dropout_mask is not None:
        hpad = hpad * dropout_mask
    logits = F.linear(hpad, p['ctc.ctc_lo.weight'], p['ctc.ctc_lo.bias']).transpose(0, 1)
    olens = torch.tensor([len(y) for y in ys], dtype=torch.long)
    nll = F.ctc_loss(logits.log_softmax(2), torch.cat(ys).long(), torch.as_tensor(hlens).long(),
                     olens, blank=0, reduction='sum')
    return (nll / hpad.size(0)).view(1)


# --------------------------------------------------------------------------------------
# F8/F9  AttLoc + Decoder   model/e2e_attention.py:236-299, model/e2e_decoder.py:78-168
# --------------------------------------------------------------------------------------
def decoder_forward(p, hpad, hlens, ys, sos_eos, ss_rate=0.0, return_att=False, sample_steps=None, labeldist=None, lsm_weight=0.0):
    """Decoder.forward (model/e2e_decoder.py:78-168).  ``sample_steps[i]`` True: step i feeds the arg-max of step i-1's
    output (scheduled sampling :123-127 fires for the whole batch on one draw; calculate_all_attentions :408-412 does it
    at every i > 0).  The caller decides the steps -- the reference draws ``random.random() < rate`` once per step."""
    assert ss_rate == 0.0, 'pass the drawn steps as sample_steps'
    hlens = [int(l) for l in hlens]
    hpad = mask_by_length(hpad, hlens, 0.0)
    B, T, _ = hpad.shape
    eos = ys[0].new_tensor([sos_eos])
    ys_in = pad_list([torch.cat([eos, y]) for y in ys], sos_eos)
    ys_out = pad_list([torch.cat([y, eos]) for y in ys], -1)
    olen = ys_out.size(1)
    dunits = p['dec.decoder.0.weight_hh'].size(1)
    z = hpad.new_zeros(B, dunits)
    c = hpad.new_zeros(B, dunits)
    pre = F.linear(hpad, p['att.mlp_enc.weight'], p['att.mlp_enc.bias'])
    att_w = pad_list([hpad.new_full((l,), 1.0 / l) for l in hlens], 0.0)
    if att_w.size(1) < T:
        att_w = F.pad(att_w, (0, T - att_w.size(1)))
    eys = F.embedding(ys_in, p['dec.embed.weight'])
    K = p['att.loc_conv.weight'].size(3)
    zs, ws = [], []
    for i in range(olen):
        conv = F.conv2d(att_w.view(B, 1, 1, T), p['att.loc_conv.weight'], padding=(0, (K - 1) // 2))
        conv = conv.squeeze(2).transpose(1, 2)
        conv = F.linear(conv, p['att.mlp_att.weight'])
        dec = F.linear(z, p['att.mlp_dec.weight']).view(B, 1, -1)
        e = F.linear(torch.tanh(conv + pre + dec), p['att.gvec.weight'], p['att.gvec.bias']).squeeze(2)
        att_w = F.softmax(2.0 * e, dim=1)
        att_c = (hpad * att_w.unsqueeze(2)).sum(1)
        if sample_steps is not None and i > 0 and sample_steps[i]:
            y_prev = F.linear(zs[-1], p['dec.output.weight'], p['dec.output.bias'])
            e_i = F.embedding(y_prev.argmax(1), p['dec.embed.weight'])
        else:
            e_i = eys[:, i]
        ey = torch.cat([e_i, att_c], 1)
        gates = F.linear(ey, p['dec.decoder.0.weight_ih'], p['dec.decoder.0.bias_ih']) + \
            F.linear(z, p['dec.decoder.0.weight_hh'], p['dec.decoder.0.bias_hh'])
        gi, gf, gg, go = gates.chunk(4, 1)
        c = torch.sigmoid(gf) * c + torch.sigmoid(gi) * torch.tanh(gg)
        z = torch.sigmoid(go) * torch.tanh(c)
        zs.append(z)
        ws.append(att_w)
    zall = torch.stack(zs, 1).reshape(B * olen, -1)
    y_all = F.linear(zall, p['dec.output.weight'], p['dec.output.bias'])
    loss = F.cross_entropy(y_all, ys_out.reshape(-1), ignore_index=-1, reduction='mean')
    loss = loss * (float(np.mean([len(y) + 1 for y in ys])) - 1.0)
    acc = th_accuracy(y_all, ys_out, -1)
    if labeldist is not None:                           # :162-166 label smoothing (sum over ALL rows / number of utterances)
        reg = -(F.log_softmax(y_all, dim=1) * labeldist).sum() / len(ys)
        loss = (1.0 - lsm_weight) * loss + lsm_weight * reg
    if return_att:
        return loss, acc, torch.stack(ws, 1)
    return loss, acc


def split_targets(targets, target_sizes):
    ys, off = [], 0
    for s in target_sizes:
        ys.append(targets[off:off + int(s)])
        off += int(s)
    return ys


def e2e_forward(p, feats, targets, lens, tlens, elayers, mtlalpha=0.5, sample_steps=None, ctc_dropout=None):
    """E2E.forward   model/e2e_model.py:169-202.  ``ctc_dropout`` = (p, seed, call): the CTC head's input dropout with the
    product's counter-based mask (oracle/philox.py)."""
    ys = split_targets(targets, tlens)
    V = p['dec.output.weight'].size(0)
    hpad, hlens = encoder_forward(p, feats, lens, elayers)
    cmask = None
    if ctc_dropout is not None:
        from .philox import dropout_mask
        B, T, E = hpad.shape
        cmask = tm_mask(dropout_mask(B * T * E, *ctc_dropout), B, T, E)
    loss_ctc = ctc_forward(p, hpad, hlens, ys, cmask) if mtlalpha != 0 else None
    loss_att, acc = decoder_forward(p, hpad, hlens, ys, V - 1, sample_steps=sample_steps) if mtlalpha != 1 else (None, None)
    return loss_ctc, loss_att, acc, hpad, hlens


def valid_rows(h, hlens):
    return torch.cat([h[i, :int(hlens[i])] for i in range(h.size(0))], 0)


def share_e2e_forward(p, clean_feat, enhance_feat, targets, lens, tlens, elayers, cmvn, mtlalpha=0.5):
    """S1 (build-defined): both branches normalised, shared encoder, losses on the
    enhanced branch, contexts = valid encoder frames of each branch."""
    nf = (lambda z: z) if cmvn is None else (lambda z: (z + cmvn[0, :]) * cmvn[1, :])
    loss_ctc, loss_att, acc, h_mix, hl = e2e_forward(p, nf(enhance_feat), targets, lens, tlens,
                                                      elayers, mtlalpha)
    h_cln, _ = encoder_forward(p, nf(clean_feat), lens, elayers)
    return loss_ctc, loss_att, acc, valid_rows(h_cln, hl), valid_rows(h_mix, hl)


def coral(src, tgt):
    """S2 (build-defined): Deep-CORAL  ||Cov(src)-Cov(tgt)||_F^2 / (4 d^2), unbiased cov."""
    d = src.size(1)

    def cov(x):
        xm = x - x.mean(0, keepdim=True)
        return xm.t().mm(xm) / (x.size(0) - 1)

    diff = cov(src) - cov(tgt)
    return (diff * diff).sum() / (4.0 * d * d)


# --------------------------------------------------------------------------------------
# F11/F12  GANModel 'basic' + GANLoss   model/gan_model.py:55-95,141-145,152-171
# --------------------------------------------------------------------------------------
def discriminator_forward(p, buf, x, cmvn=None, train=True, momentum=0.1, eps=1e-5, use_sigmoid=False):
    """``buf`` holds running_mean/var/num_batches_tracked and is updated in place (train).  ``use_sigmoid``: the
    --no_lsgan discriminator ends in nn.Sigmoid (gan_model.py:90-91,126)."""
    if cmvn is not None:                                 # S3
        x = (x + cmvn[0, :]) * cmvn[1, :]
    h = x.unsqueeze(1) if x.dim() == 3 else x
    h = F.leaky_relu(F.conv2d(h, p['model.0.weight'], p['model.0.bias'], stride=2, padding=1), 0.2)
    for conv, bn, stride in (('model.2', 'model.3', 2), ('model.5', 'model.6', 2), ('model.8', 'model.9', 1)):
        h = F.conv2d(h, p[conv + '.weight'], None, stride=stride, padding=1)
        h = F.batch_norm(h, buf[bn + '.running_mean'], buf[bn + '.running_var'], p[bn + '.weight'],
                         p[bn + '.bias'], train, momentum, eps)
        if train:
            buf[bn + '.num_batches_tracked'] += 1
        h = F.leaky_relu(h, 0.2)
    h = F.conv2d(h, p['model.11.weight'], p['model.11.bias'], stride=1, padding=1)
    return torch.sigmoid(h) if use_sigmoid else h


def gan_loss(d_out, target_is_real, use_lsgan=True):
    """GANLoss (gan_model.py:152-171): MSE to the broadcast label, or nn.BCELoss on probabilities with --no_lsgan."""
    t = 1.0 if target_is_real else 0.0
    if use_lsgan:
        return ((d_out - t) ** 2).mean()
    return F.binary_cross_entropy(d_out, torch.full_like(d_out, t))
