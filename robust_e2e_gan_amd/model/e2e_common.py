"""Helpers that fix padding / init / freezing semantics (mirror of model/e2e_common.py)."""
import collections
import math
import weakref

import numpy as np
import torch


class ModelBase(torch.nn.Module):
    """model/e2e_common.py:16-38 -- ``load_model(path, state_dict_key, opt)`` classmethod."""

    def forward(self, x):
        raise NotImplementedError

    @classmethod
    def load_model(cls, path, state_dict, opt=None):
        if path is not None:
            package = torch.load(path, map_location=lambda storage, loc: storage, weights_only=False)
            model = cls(args=package['opt'])
            if state_dict in package and package[state_dict] is not None:
                model.load_state_dict(package[state_dict])
                print('checkpoint found at {} {}'.format(path, state_dict))
        else:
            model = cls(opt)
            print('no checkpoint found, so init model')
        if opt is not None and len(getattr(opt, 'gpu_ids', [])) > 0:
            model = model.cuda()
        return model

    @staticmethod
    def get_param_size(model):
        return sum(p.numel() for p in model.parameters())


def set_requires_grad(nets, requires_grad=False):
    """model/e2e_common.py:71-77"""
    if not isinstance(nets, list):
        nets = [nets]
    for net in nets:
        if net is not None:
            for param in net.parameters():
                param.requires_grad = requires_grad


def module_device(m):
    return next(m.parameters()).device


def to_cuda(m, x):
    """model/e2e_common.py:80-85 -- move ``x`` to the module's device (no-op when already there)."""
    dev = module_device(m)
    if isinstance(x, torch.Tensor) and x.device != dev:
        return x.to(dev, non_blocking=True)
    return x


def lens_list(lens):
    if isinstance(lens, torch.Tensor):
        return [int(v) for v in lens.tolist()]
    return [int(v) for v in lens]


def host_to_dev(values, device, dtype=torch.int32):
    """Small host array -> device tensor WITHOUT stalling the host: staged in pinned memory and copied with
    non_blocking=True.  (``torch.tensor(list, device=cuda)`` is a blocking copy: it waits for everything already
    enqueued on the stream, which serialises the host behind the GPU in the middle of the training step.)"""
    t = torch.as_tensor(np.asarray(values), dtype=dtype)
    if torch.device(device).type != 'cuda':
        return t.to(device)
    return t.pin_memory().to(device, non_blocking=True)


_LENS_CACHE = collections.OrderedDict()     # least recently used first: a step touches ~25 entries, so nothing a step in flight uses is ever evicted
_LENS_CACHE_MAX = 256
LENS_HOST = {}        # data_ptr of a cached length tensor -> the host tuple it was made from (ops.row_maps: the valid rows of a ragged batch)


def dev_cached(key, make, device):
    """Device copy of a small host array ``make()`` (int32), cached under ``key``.  An entry remembers the stream its
    (non-blocking) upload was enqueued on and an event behind that copy: a user on ANOTHER stream (the CTC branch on the
    aux stream, the decoder on main) waits for the event and tells the caching allocator about its use, so neither a read
    ahead of the copy nor a reuse of the block under a late reader can happen."""
    ent = _LENS_CACHE.get(key)
    cuda = torch.device(device).type == 'cuda'
    if ent is None:
        if len(_LENS_CACHE) >= _LENS_CACHE_MAX:
            # evict the least recently used quarter (never everything: the length tensors of the step being enqueued must keep their host
            # tuples -- ops.row_maps would silently fall back to all rows for the rest of the step -- and the tensors of steps still in flight
            # stay referenced by their autograd graphs; users on other streams have told the allocator, ``record_stream`` below / ops._touch)
            for _ in range(_LENS_CACHE_MAX // 4):
                _LENS_CACHE.popitem(last=False)
            for k_ in [k_ for k_, (_, ref) in LENS_HOST.items() if ref() is None]:
                del LENS_HOST[k_]
        t = host_to_dev(np.asarray(make(), np.int32), device)
        if cuda:
            ev = torch.cuda.Event()
            ev.record()
            ent = (t, ev, torch.cuda.current_stream().cuda_stream, set())
        else:
            ent = (t, None, None, set())
        _LENS_CACHE[key] = ent
        return t
    _LENS_CACHE.move_to_end(key)
    t, ev, sid, seen = ent
    if cuda:
        cur = torch.cuda.current_stream()
        if cur.cuda_stream != sid and cur.cuda_stream not in seen:
            cur.wait_event(ev)
            t.record_stream(cur)
            seen.add(cur.cuda_stream)
    return t


def lens_dev(lens, device):
    """int32 device copy of a length list; the same lists recur many times per step, so uploads are cached (``dev_cached``)."""
    key = (tuple(lens_list(lens)), str(device))
    t = dev_cached(key, lambda: key[0], device)
    LENS_HOST[t.data_ptr()] = (key[0], weakref.ref(t))
    return t


def host_lens_of(lens_d):
    """The host tuple a cached length tensor was made from, or None for any other tensor (also one that merely reuses a cached tensor's address)."""
    ent = LENS_HOST.get(lens_d.data_ptr())
    return ent[0] if ent is not None and ent[1]() is lens_d else None


def lecun_normal_init_parameters(module):
    """model/e2e_common.py:135-154"""
    for p in module.parameters():
        data = p.data
        if data.dim() == 1:
            data.zero_()
        elif data.dim() == 2:
            data.normal_(0, 1.0 / math.sqrt(data.size(1)))
        elif data.dim() == 4:
            n = data.size(1)
            for k in data.size()[2:]:
                n *= k
            data.normal_(0, 1.0 / math.sqrt(n))
        else:
            raise NotImplementedError


def set_forget_bias_to_one(bias):
    """model/e2e_common.py:220-223"""
    n = bias.size(0)
    bias.data[n // 4:n // 2].fill_(1.0)


def _get_vgg2l_odim(idim, in_channel=1, out_channel=128):
    """model/e2e_common.py:164-168"""
    idim = idim / in_channel
    idim = np.ceil(np.array(idim, dtype=np.float32) / 2)
    idim = np.ceil(np.array(idim, dtype=np.float32) / 2)
    return int(idim) * out_channel


def pad_list(xs, pad_value):
    """model/e2e_common.py:208-217 (host-side helper for label tensors)"""
    n = len(xs)
    ml = max(x.size(0) for x in xs)
    out = xs[0].new_full((n, ml) + tuple(xs[0].shape[1:]), pad_value)
    for i, x in enumerate(xs):
        out[i, :x.size(0)] = x
    return out


class LinearParams(torch.nn.Module):
    """Parameter holder with nn.Linear's names/shapes (weight (out,in), bias (out))."""

    def __init__(self, idim, odim, bias=True):
        super().__init__()
        self.weight = torch.nn.Parameter(torch.empty(odim, idim))
        self.bias = torch.nn.Parameter(torch.empty(odim)) if bias else None
        b = 1.0 / math.sqrt(idim)
        self.weight.data.uniform_(-b, b)
        if bias:
            self.bias.data.uniform_(-b, b)


class ConvParams(torch.nn.Module):
    """Parameter holder with nn.Conv2d's names/shapes (weight (Cout,Cin,kh,kw), bias (Cout))."""

    def __init__(self, cin, cout, kh, kw=None, bias=True, stride=1, padding=0):
        super().__init__()
        kw = kh if kw is None else kw
        self.weight = torch.nn.Parameter(torch.empty(cout, cin, kh, kw))
        self.bias = torch.nn.Parameter(torch.empty(cout)) if bias else None
        b = 1.0 / math.sqrt(cin * kh * kw)
        self.weight.data.uniform_(-b, b)
        if bias:
            self.bias.data.uniform_(-b, b)
        self.stride, self.padding = stride, padding


class LSTMParams(torch.nn.Module):
    """Parameter holder with nn.LSTM(bidirectional=True)'s names: weight_ih_l{k}[_reverse] ..."""

    def __init__(self, idim, hdim, layers):
        super().__init__()
        self.hdim, self.layers = hdim, layers
        b = 1.0 / math.sqrt(hdim)
        for l in range(layers):
            for sfx in ('', '_reverse'):
                i = idim if l == 0 else 2 * hdim
                for name, shape in (('weight_ih', (4 * hdim, i)), ('weight_hh', (4 * hdim, hdim)), ('bias_ih', (4 * hdim,)),
                                    ('bias_hh', (4 * hdim,))):
                    p = torch.nn.Parameter(torch.empty(*shape).uniform_(-b, b))
                    setattr(self, '%s_l%d%s' % (name, l, sfx), p)

    def layer_weights(self, l):
        out = []
        for sfx in ('', '_reverse'):
            for name in ('weight_ih', 'weight_hh', 'bias_ih', 'bias_hh'):
                out.append(getattr(self, '%s_l%d%s' % (name, l, sfx)))
        return out
