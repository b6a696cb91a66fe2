"""torch.autograd.Function wrappers around the re2e C ABI (include/re2e.h).

Every arithmetic step of the hot path goes through ``lib.call`` into libre2e_hip.so; torch is used
for device memory (``torch.empty``), views and the autograd graph only.  Weight gradients are
accumulated by the kernels straight into ``param.grad`` (GEMM/conv epilogue ``beta=1``), which is
what lets the optimizer and the RCCL all-reduce work on one flat buffer per network
(robust_e2e_gan_amd/optim.py, dist.py).
"""
import os
import time
import torch

from . import lib
from .lib import call, ptr, query, workspace

ACT = {None: lib.ACT_NONE, 'tanh': lib.ACT_TANH, 'relu': lib.ACT_RELU, 'lrelu': lib.ACT_LRELU, 'sigmoid': lib.ACT_SIGMOID}


def _f32(t):
    if t.dtype != torch.float32:
        t = t.float()
    return t if t.is_contiguous() else t.contiguous()


def _need_gpu(t):
    if not t.is_cuda:
        raise lib.Re2eError('robust_e2e_gan_amd ops run on the GPU only (no CPU fallback); got a %s tensor' % t.device)


def empty(shape, like):
    return torch.empty(shape, dtype=torch.float32, device=like.device)


def zeros(shape, like):
    return torch.zeros(shape, dtype=torch.float32, device=like.device)


# ---------------------------------------------------------------------------------------------
# raw helpers
# ---------------------------------------------------------------------------------------------
def gemm(A, B, C, M, N, K, transa=False, transb=False, lda=None, ldb=None, ldc=None, bias=None, bias2=None,
         act=lib.ACT_NONE, beta=0.0, mul=None, mask_out=None, lens=None, T=0, dev=None):
    """C[M,N] = act(op(A) op(B) + bias + bias2) + beta*C   (see re2e_gemm)."""
    lda = lda if lda is not None else (M if transa else K)
    ldb = ldb if ldb is not None else (K if transb else N)
    ldc = ldc if ldc is not None else N
    wsb = query('re2e_gemm_workspace_bytes', int(transa), int(transb), M, N, K)
    ws = workspace(wsb, dev if dev is not None else A.device, 'gemm') if wsb else None
    _p = lambda t: t if isinstance(t, int) else t.data_ptr()       # tensors or raw device addresses (sub-matrix views)
    call('re2e_gemm', int(transa), int(transb), M, N, K, _p(A), lda, _p(B), ldb, _p(C), ldc,
         ptr(bias), ptr(bias2), act, float(beta), ptr(mul), ptr(mask_out), ptr(lens), T, ptr(ws), wsb)
    return C


NT_INPUT_GRAD = lib.exp_env('RE2E_NO_NT_INPUT_GRAD') is None
INLINE_LAST_WGRAD = lib.exp_env('RE2E_NO_INLINE_LAST_WGRAD') is None
FUSED_DBIAS = lib.exp_env('RE2E_NO_FUSED_DBIAS') is None


def gemm_input_grad(dz, W, dx, M, K, N, beta=0.0):
    """dx[M,K] = dz[M,N] W[N,K] (+ beta dx): the input gradient of y = x W^T.  For many rows the weight is transposed first (one
    small gather, N*K elements) and the product runs as dz (W^T)^T on the engine's k-major path -- 256x128 tiles with the
    interleaved k-step and the round-filling row tail -- instead of the row-major-B path (MI355X, 12800 x 2560 x 2048:
    113 -> 128 TFLOP/s; RE2E_NO_NT_INPUT_GRAD=1 keeps the direct form)."""
    if NT_INPUT_GRAD and M >= 2048 and K % 4 == 0 and N % 4 == 0 and W.is_contiguous():
        wt = empty((K, N), dz)
        call('re2e_conv_weight_gather', W.data_ptr(), wt.data_ptr(), N, K, 1, 1, 1, 1, 1, 0, 0, 1)       # wt[k][n] = W[n][k]
        return gemm(dz, wt, dx, M, K, N, transb=True, beta=beta)
    return gemm(dz, W, dx, M, K, N, beta=beta)


# ---------------------------------------------------------------------------------------------
# Ragged time-major batches: products over the VALID rows only (re2e_gemm_nt_rows)
# ---------------------------------------------------------------------------------------------
# The reference packs its sequences (pack_padded_sequence) and never computes a padded (t, b) row; the recurrences here run on the padded
# (T, B, .) layout, where a config-4 batch (lengths 0.7 T .. T) is 15 % padding.  ``row_maps(lens_dev, T, B)`` gives the physical rows t * B + b
# with t < len_b (and the others): the x W^T products around the recurrences then run over those rows only, gathered / scattered through the
# per-lane offsets of the LDS-DMA pipeline.  RE2E_NO_ROW_MAPS=1 (experiments): all rows, as rounds 1-5 did.
ROW_MAPS = lib.exp_env('RE2E_NO_ROW_MAPS') is None
TN_ROW_MAPS = lib.exp_env('RE2E_NO_TN_ROW_MAPS') is None      # (experiments) the weight gradients over all rows
ROW_IDENT = lib.exp_env('RE2E_NO_ROW_IDENT') is None           # (experiments) every tile looks its rows up
ROW_MAPS_MIN_K = 384
ROW_MAPS_MIN_PAD = 0.04      # below this share of padded rows the maps are not worth their per-tile lookups


class RowMaps(object):
    """``ident``: valid[r] == r for r < ident (the first padded physical row: about min(len) * B of a time-major batch): tiles of those rows skip the table."""
    __slots__ = ('valid', 'invalid', 'nv', 'ni', 'rows', 'ident')

    def __init__(self, valid, invalid, nv, ni, rows, ident):
        self.valid, self.invalid, self.nv, self.ni, self.rows, self.ident = valid, invalid, nv, ni, rows, ident


_ROW_MAPS = {}


def row_maps(lens_d, T, B):
    """RowMaps of a (T, B, .) time-major tensor whose utterance lengths are the cached length tensor ``lens_d`` (model.e2e_common.lens_dev), or
    None: maps off, lengths not known on the host, too little padding, too few rows."""
    if not ROW_MAPS or lens_d is None:
        return None
    from .model import e2e_common as ec
    lens = ec.host_lens_of(lens_d)
    if lens is None or len(lens) != B:
        return None
    key = (lens, T, B, str(lens_d.device))
    hit = _ROW_MAPS.get(key)
    if hit is None:
        import numpy as np
        ln = np.minimum(np.asarray(lens, np.int64), T)
        ok = (np.arange(T, dtype=np.int64)[:, None] < ln[None, :]).reshape(-1)
        idx = np.arange(T * B, dtype=np.int32)
        v_ = idx[ok]
        neq = np.nonzero(v_ != idx[:v_.size])[0]
        hit = (v_, idx[~ok], int(neq[0]) if neq.size else int(v_.size))
        while len(_ROW_MAPS) >= 64:
            _ROW_MAPS.pop(next(iter(_ROW_MAPS)))          # oldest first (dicts keep insertion order)
        _ROW_MAPS[key] = hit
    v, iv, ident = hit
    if v.size < 256 or iv.size < ROW_MAPS_MIN_PAD * T * B:
        return None
    dev = lens_d.device
    return RowMaps(ec.dev_cached(('rows_v',) + key, lambda: v, dev), ec.dev_cached(('rows_i',) + key, lambda: iv, dev), int(v.size), int(iv.size), T * B,
                   ident if ROW_IDENT else 0)


def gemm_rows(A, B, C, N, K, maps, lda=None, ldb=None, ldc=None, bias=None, bias2=None, act=lib.ACT_NONE, beta=0.0, fill=False):
    """C[r] = act(A[r] B^T + bias + bias2) + beta C[r] over the rows r of ``maps.valid`` (B stored (N, K): x W^T); ``fill``: the other rows of C
    get what the product over a zero row of A leaves there -- act(bias), zeros without a bias (only with beta = 0, at most one bias): upstream's
    Linear + tanh over the zero-padded frames (e2e_encoder.py:145-147,173-176).  Falls back to the product over all rows when the pipeline
    declines the shape."""
    if fill and bias2 is not None:
        raise lib.Re2eError('gemm_rows(fill=True) takes one bias')
    lda = lda if lda is not None else K
    ldb = ldb if ldb is not None else K
    ldc = ldc if ldc is not None else N
    if K < ROW_MAPS_MIN_K and not fill:
        # a short contraction is over before the per-tile row lookups have paid (25600 x 1024 x 260: 141 us mapped against 131 over all rows,
        # profiles/r05_rows_alone.txt): all rows
        return gemm(A, B, C, maps.rows, N, K, transb=True, lda=lda, ldb=ldb, ldc=ldc, bias=bias, bias2=bias2, act=act, beta=beta)
    wsb = query('re2e_gemm_workspace_bytes', 0, 1, maps.nv, N, K)
    ws = workspace(wsb, A.device, 'gemm') if wsb else None
    _p = lambda t: t if isinstance(t, int) else t.data_ptr()
    if lib.call_supported('re2e_gemm_nt_rows', maps.nv, N, K, _p(A), lda, _p(B), ldb, _p(C), ldc, ptr(bias), ptr(bias2), act, float(beta),
                          maps.valid.data_ptr(), maps.ident, maps.rows, ptr(ws), wsb):
        if fill and N % 4 == 0 and ldc % 4 == 0:
            call('re2e_fill_rows', _p(C), ldc, N, maps.invalid.data_ptr(), maps.ni, 0.0, ptr(bias), act if bias is not None else lib.ACT_NONE)
        return C
    return gemm(A, B, C, maps.rows, N, K, transb=True, lda=lda, ldb=ldb, ldc=ldc, bias=bias, bias2=bias2, act=act, beta=beta)


def gemm_input_grad_rows(dz, W, dx, K, N, maps, beta=0.0, fill=False):
    """``gemm_input_grad`` over the valid rows: dx[r] = dz[r] W (+ beta dx[r]); ``fill``: zeros in the padded rows of dx."""
    if K % 4 == 0 and N % 4 == 0 and W.is_contiguous():
        wt = empty((K, N), dz)
        call('re2e_conv_weight_gather', W.data_ptr(), wt.data_ptr(), N, K, 1, 1, 1, 1, 1, 0, 0, 1)       # wt[k][n] = W[n][k]
        return gemm_rows(dz, wt, dx, K, N, maps, beta=beta, fill=fill)
    return gemm_input_grad(dz, W, dx, maps.rows, K, N, beta=beta)


def gemm_tn_rows(A, B, C, M, N, maps, beta=0.0, lda=None, ldb=None, ldc=None):
    """C[M,N] = sum over the rows r of ``maps.valid`` of A[r][:M]^T B[r][:N] + beta C (the weight gradient dy^T x of a ragged time-major batch:
    dy is zero in the padded rows, so the sum over the valid rows is the whole sum).  Falls back to the contraction over all rows."""
    lda = lda if lda is not None else M
    ldb = ldb if ldb is not None else N
    ldc = ldc if ldc is not None else N
    wsb = query('re2e_gemm_workspace_bytes', 1, 0, M, N, maps.nv)
    ws = workspace(wsb, A.device if not isinstance(A, int) else C.device, 'gemm') if wsb else None
    _p = lambda t: t if isinstance(t, int) else t.data_ptr()
    if TN_ROW_MAPS and lib.call_supported('re2e_gemm_tn_rows', M, N, maps.nv, _p(A), lda, _p(B), ldb, _p(C), ldc, float(beta), maps.valid.data_ptr(), maps.ident,
                          maps.rows, ptr(ws), wsb):
        return C
    wsb = query('re2e_gemm_workspace_bytes', 1, 0, M, N, maps.rows)
    ws = workspace(wsb, C.device, 'gemm') if wsb else None
    call('re2e_gemm', 1, 0, M, N, maps.rows, _p(A), lda, _p(B), ldb, _p(C), ldc, None, None, lib.ACT_NONE, float(beta), None, None, None, 0, ptr(ws), wsb)
    return C


def colsum_into(A2d, M, N, out, beta, lda=None):
    wsb = query('re2e_colsum_workspace_bytes', M, N)
    ws = workspace(wsb, A2d.device, 'colsum')
    call('re2e_colsum', A2d.data_ptr(), M, N, lda if lda is not None else N, out.data_ptr(), float(beta), ws.data_ptr(), wsb)


MULTI_STREAM = False     # set by JointTrainer when branches run on side streams
WGRAD_STREAM = None      # optional stream for weight-gradient kernels (see ``param_grads``)
AUX_STREAM = None        # optional filler stream for independent branches inside a module (ShareE2E: the CTC branch)
FROZEN_PARAMS = frozenset()   # id()s of parameters whose gradients must NOT be produced by the backward now running (the trainer
#                               builds D's graph once with trainable parameters and walks it twice: G-step = input gradient only)
MARKS = None             # RE2E_TIMELINE: list of (label, host time, event) shared with JointTrainer (see ``mark_grad``)
SYNC_BN = False          # data-parallel runs that shard ONE global batch: BatchNorm statistics over all ranks' rows (BnLreluFn)
BN_DEFER_RUNNING = False  # with a sink: leave the running statistics alone in this forward (the owner replays the update later, in order)
BN_STATS_SINK = None     # optional list: every BatchNorm forward appends (running_mean, running_var, mean, invstd, P, momentum, eps)


def mark_grad(t, label):
    """RE2E_TIMELINE: note when the backward pass has produced the gradient of ``t`` -- an event on the stream autograd
    runs that node on, i.e. behind everything enqueued on it so far."""
    if MARKS is not None and isinstance(t, torch.Tensor) and t.requires_grad:
        marks = MARKS

        def hook(g):
            ev = torch.cuda.Event(enable_timing=True)
            ev.record()
            marks.append(('  grad of ' + label, time.perf_counter(), ev))
        t.register_hook(hook)
    return t


def mark(label):
    """RE2E_TIMELINE: an event on the current stream, behind everything enqueued on it so far (forward-side twin of ``mark_grad``)."""
    if MARKS is not None:
        ev = torch.cuda.Event(enable_timing=True)
        ev.record()
        MARKS.append(('  ' + label, time.perf_counter(), ev))


def _wants(ctx, i, p):
    return ctx.needs_input_grad[i] and (p is None or id(p) not in FROZEN_PARAMS)


class param_grads(object):
    """``with param_grads(t1, t2, ...):`` runs the enclosed weight/bias-gradient kernels on WGRAD_STREAM.

    Nothing downstream in the backward pass depends on a weight gradient, so those kernels form a reservoir of
    MFMA work that fills the CUs left idle by the latency-bound parts of the step (recurrent chains, decoder).
    ``t*`` are the tensors the kernels read: they are kept alive for that stream.  Ordering of the accumulation
    into shared gradients is handled by ``accumulate``; the trainer joins the stream before the optimizer."""
    __slots__ = ('ts', 'ctx', 'inline')

    def __init__(self, *tensors, inline=False):
        self.ts = tensors
        self.ctx = None
        self.inline = inline        # keep the kernels on the current stream (see BiLstmFn.backward)

    def __enter__(self):
        ws = WGRAD_STREAM
        if not MULTI_STREAM or ws is None or self.inline:
            return self
        cur = torch.cuda.current_stream()
        if cur == ws:
            return self
        ev = torch.cuda.Event()
        ev.record(cur)
        ws.wait_event(ev)
        for t in self.ts:
            if isinstance(t, torch.Tensor):
                t.record_stream(ws)
            elif t is not None:
                # RowMaps / RowLims: small cached index tensors (model.e2e_common.dev_cached) that the weight-gradient kernels read on THIS
                # stream -- the cache may evict them while such a kernel is still queued
                for name in t.__slots__:
                    v = getattr(t, name)
                    if isinstance(v, torch.Tensor):
                        v.record_stream(ws)
        self.ctx = torch.cuda.stream(ws)
        self.ctx.__enter__()
        return self

    def __exit__(self, *exc):
        if self.ctx is not None:
            self.ctx.__exit__(*exc)
            self.ctx = None
        return False


class accumulate(object):
    """``with accumulate(p) as (g, beta): <kernels that compute g = beta*g + dL/dp>``.

    The kernels write straight into ``p.grad`` (autograd never sees parameter gradients), so when the two ASR
    branches or the D passes run on different HIP streams the read-modify-write of a shared gradient must be
    ordered by hand: wait for the event of the previous writer, record a new one when this writer is enqueued."""
    __slots__ = ('p',)

    def __init__(self, p):
        self.p = p

    def __enter__(self):
        p = self.p
        if MULTI_STREAM:
            ev = getattr(p, '_re2e_ev', None)
            if ev is not None:
                torch.cuda.current_stream().wait_event(ev)
        if p.grad is None:
            p.grad = torch.empty_like(p)
            return p.grad, 0.0
        return p.grad, 1.0

    def __exit__(self, *exc):
        if MULTI_STREAM:
            ev = torch.cuda.Event()
            ev.record()
            self.p._re2e_ev = ev
        return False


def act_bwd(dy, y, act):
    """dz = dy * act'(y) (out of place: autograd may share dy with other consumers)."""
    if act == lib.ACT_NONE:
        return dy
    dz = torch.empty_like(dy)
    call('re2e_act_bwd', dy.data_ptr(), y.data_ptr(), dz.data_ptr(), dy.numel(), act)
    return dz


def act_bwd_bias(dy2, y, act, b, b_needs_grad=None):
    """dz = dy * act'(y) for a (rows, N) gradient and, when the bias ``b`` needs a gradient, b.grad += colsum(dz) in the
    same pass (re2e_act_bwd_colsum).  Returns (dz, bias_done)."""
    if b_needs_grad is None:
        b_needs_grad = b is not None and b.requires_grad
    if act == lib.ACT_NONE or b is None or not b_needs_grad:
        return act_bwd(dy2, y, act), False
    M, N = dy2.shape
    dz = torch.empty_like(dy2)
    wsb = query('re2e_colsum_workspace_bytes', M, N)
    ws = workspace(wsb, dy2.device, 'colsum')
    with accumulate(b) as (gb, beta):
        call('re2e_act_bwd_colsum', dy2.data_ptr(), y.data_ptr(), dz.data_ptr(), M, N, act, gb.data_ptr(), float(beta), ws.data_ptr(), wsb)
    return dz, True


# ---------------------------------------------------------------------------------------------
# Linear (+ bias + activation)     torch.nn.Linear call sites, see include/re2e.h K3
# ---------------------------------------------------------------------------------------------
_PADDED_GRADS = {}        # data_ptr -> (rows, N, padded N): gradients written with zero-padded rows by their producer (CtcFn.backward)


def _linear_backward_padded(ctx, dy, x2, W, b, M, K, N, Np, need_w, need_b):
    """LinearFn.backward for a gradient whose rows are zero-padded to Np floats: both products over the padded width."""
    dzp = dy.as_strided((M, Np), (Np, 1))
    dx = None
    if ctx.needs_input_grad[0]:
        wt = zeros((K, Np), dzp)
        wt[:, :N].copy_(W.t())                                   # W^T, zero-padded: the k-major B operand
        dx = empty((M, K), x2)
        gemm(dzp, wt, dx, M, K, Np, transb=True)                 # dx = dz W
        dx = dx.view(ctx.xshape)
    with param_grads(dzp, x2):
        if need_w:
            tmp = empty((Np, K), x2)
            gemm(dzp, x2, tmp, Np, K, M, transa=True)            # rows N .. Np-1 are zero
            with accumulate(W) as (gw, beta):
                if beta:
                    gw.add_(tmp[:N])
                else:
                    gw.copy_(tmp[:N])
        if b is not None and need_b:
            with accumulate(b) as (gb, beta):
                colsum_into(dzp, M, N, gb, beta, lda=Np)
    return dx, None, None, None, None


class LinearFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, W, b, act, maps=None):
        _need_gpu(x)
        x2 = _f32(x).view(-1, x.shape[-1])
        M, K = x2.shape
        N = W.shape[0]
        y = empty((M, N), x)
        if maps is not None and maps.rows == M and W.is_contiguous() and N % 4 == 0:
            # ragged time-major rows: the product over the valid (t, b) only; the padded rows get act(b) written -- what upstream's Linear over
            # the zero-padded frames leaves there (nothing downstream reads them: the next recurrence packs them away, attention and CTC mask them)
            gemm_rows(x2, W, y, N, K, maps, bias=b, act=act, fill=True)
        else:
            maps = None
            gemm(x2, W, y, M, N, K, transb=True, bias=b, act=act)
        ctx.maps = maps
        ctx.act, ctx.W, ctx.b = act, W, b
        ctx.save_for_backward(x2, y if act != lib.ACT_NONE else None)
        ctx.xshape = x.shape
        return y.view(*x.shape[:-1], N)

    @staticmethod
    def backward(ctx, dy):
        x2, y = ctx.saved_tensors
        W, b = ctx.W, ctx.b
        M, K = x2.shape
        N = W.shape[0]
        # which parameters get gradients was fixed when the graph was built (ctx.needs_input_grad), NOT by the
        # parameters' requires_grad flags at backward time: the trainer re-enables D's parameters for the D-step
        # while the G-step backward through D may still be pending
        need_w, need_b = _wants(ctx, 1, W), _wants(ctx, 2, b)
        pad = _PADDED_GRADS.pop(dy.data_ptr(), None) if _PADDED_GRADS else None
        if (pad is not None and pad[:2] == (M, N) and dy.dim() >= 2 and dy.stride(-1) == 1 and dy.stride(-2) == pad[2]      # really that buffer
                and ctx.act == lib.ACT_NONE and dy.dtype == torch.float32 and W.is_contiguous()):
            return _linear_backward_padded(ctx, dy, x2, W, b, M, K, N, pad[2], need_w, need_b)
        dz, bias_done = act_bwd_bias(_f32(dy).reshape(M, N), y, ctx.act, b, need_b)
        dx = None
        if ctx.needs_input_grad[0]:
            dx = empty((M, K), x2)
            if ctx.maps is not None and K % 4 == 0:
                gemm_input_grad_rows(dz, W, dx, K, N, ctx.maps, fill=True)      # (dz is zero in the padded rows: so is dx)
            else:
                gemm_input_grad(dz, W, dx, M, K, N)            # dx = dz[M,N] * W[N,K]
            dx = dx.view(ctx.xshape)
        with param_grads(dz, x2, ctx.maps):
            if need_w:
                with accumulate(W) as (gw, beta):
                    if ctx.maps is not None:
                        gemm_tn_rows(dz, x2, gw, N, K, ctx.maps, beta=beta)       # dW = dz^T x over the valid rows (dz is zero in the others)
                    else:
                        gemm(dz, x2, gw, N, K, M, transa=True, beta=beta)   # dW = dz^T x
            if b is not None and need_b and not bias_done:
                with accumulate(b) as (gb, beta):
                    colsum_into(dz, M, N, gb, beta)
        return dx, None, None, None, None


def linear(x, W, b=None, act=None, maps=None):
    """``maps``: RowMaps of a ragged time-major ``x`` (T, B, .): the product runs over its valid rows only (see ``row_maps``)."""
    return LinearFn.apply(x, W, b, ACT[act] if not isinstance(act, int) else act, maps)


class Transpose01Fn(torch.autograd.Function):
    """(D0, D1, W) -> (D1, D0, W): batch-first <-> time-major."""

    @staticmethod
    def forward(ctx, x):
        _need_gpu(x)
        x = _f32(x)
        D0, D1, W = x.shape
        y = empty((D1, D0, W), x)
        call('re2e_transpose01', x.data_ptr(), y.data_ptr(), D0, D1, W)
        return y

    @staticmethod
    def backward(ctx, dy):
        dy = _f32(dy)
        D1, D0, W = dy.shape
        dx = empty((D0, D1, W), dy)
        call('re2e_transpose01', dy.data_ptr(), dx.data_ptr(), D1, D0, W)
        return dx


transpose01 = Transpose01Fn.apply


# ---------------------------------------------------------------------------------------------
# K2 fbank
# ---------------------------------------------------------------------------------------------
class FbankFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, band, cmvn, want_raw, want_norm):
        _need_gpu(x)
        x = _f32(x)
        off, ln, w, maxw, NF = band[:5]
        rows, F = x.numel() // x.shape[-1], x.shape[-1]
        raw = empty(x.shape[:-1] + (NF,), x) if want_raw else None
        nrm = empty(x.shape[:-1] + (NF,), x) if want_norm else None
        pw = empty(x.shape[:-1] + (NF,), x) if x.requires_grad else None      # the band power itself, for the backward kernel
        call('re2e_fbank_fwd', x.data_ptr(), rows, F, NF, off.data_ptr(), ln.data_ptr(), w.data_ptr(), maxw, ptr(raw), ptr(nrm), ptr(cmvn),
             ptr(pw))
        ctx.band, ctx.cmvn = band, cmvn
        if pw is not None:
            ctx.save_for_backward(x, pw)
        outs = (raw if want_raw else x.new_empty(0), nrm if want_norm else x.new_empty(0))
        ctx.mark_non_differentiable(*[o for o, w_ in zip(outs, (want_raw, want_norm)) if not w_])
        ctx.want = (want_raw, want_norm)
        return outs

    @staticmethod
    def backward(ctx, draw, dnorm):
        x, pw = ctx.saved_tensors
        NF, toff, tln, tw, maxc = ctx.band[4:]
        rows, F = x.numel() // x.shape[-1], x.shape[-1]
        draw = _f32(draw) if (ctx.want[0] and draw is not None) else None
        dnorm = _f32(dnorm) if (ctx.want[1] and dnorm is not None) else None
        if draw is None and dnorm is None:
            return None, None, None, None, None
        dx = empty(x.shape, x)
        call('re2e_fbank_bwd', x.data_ptr(), rows, F, NF, toff.data_ptr(), tln.data_ptr(), tw.data_ptr(), maxc, pw.data_ptr(), ptr(draw),
             ptr(dnorm), ptr(ctx.cmvn), dx.data_ptr())
        return dx, None, None, None, None


def fbank(x, band, cmvn=None, want_raw=True, want_norm=False):
    raw, nrm = FbankFn.apply(x, band, cmvn, want_raw, want_norm)
    return (raw if want_raw else None), (nrm if want_norm else None)


class FbankDenseFn(torch.autograd.Function):
    """Trainable dense filterbank (fbank_opti_type 'train', feat_model.py:105-109,118-135): y = log(max((x^2) W, 1e-7))
    [-> (y + cmvn0) * cmvn1] with W a (F, NF) Parameter: GEMM engine for x^2 W, dz W^T and (x^2)^T dz."""

    @staticmethod
    def forward(ctx, x, W, cmvn):
        _need_gpu(x)
        x = _f32(x)
        F_, NF = W.shape
        rows = x.numel() // F_
        x2 = x.reshape(rows, F_)
        sq = empty((rows, F_), x)
        call('re2e_mul', x2.data_ptr(), x2.data_ptr(), sq.data_ptr(), sq.numel())
        z = empty((rows, NF), x)
        gemm(sq, W, z, rows, NF, F_)                               # (x^2) W, W stored (K=F, N=NF)
        y = empty(x.shape[:-1] + (NF,), x)
        call('re2e_logclamp_fwd', z.data_ptr(), ptr(cmvn), rows, NF, y.data_ptr())
        ctx.W, ctx.cmvn = W, cmvn
        ctx.save_for_backward(x2, sq, z)
        return y

    @staticmethod
    def backward(ctx, dy):
        x2, sq, z = ctx.saved_tensors
        W, cmvn = ctx.W, ctx.cmvn
        rows, F_ = x2.shape
        NF = W.shape[1]
        dy = _f32(dy)
        dz = empty((rows, NF), dy)
        call('re2e_logclamp_bwd', z.data_ptr(), ptr(cmvn), rows, NF, dy.data_ptr(), dz.data_ptr())
        dx = None
        if ctx.needs_input_grad[0]:
            ds = empty((rows, F_), dy)
            gemm(dz, W, ds, rows, F_, NF, transb=True)            # dz W^T
            dx = empty((rows, F_), dy)
            call('re2e_mul', x2.data_ptr(), ds.data_ptr(), dx.data_ptr(), dx.numel())
            call('re2e_axpby', 2.0, dx.data_ptr(), 0.0, dx.data_ptr(), dx.numel())      # d(x^2) = 2 x
            dx = dx.view(dy.shape[:-1] + (F_,))
        if _wants(ctx, 1, W):
            with param_grads(dz, sq), accumulate(W) as (gw, beta):
                gemm(sq, dz, gw, F_, NF, rows, transa=True, beta=beta)       # (x^2)^T dz
        return dx, None, None


def fbank_dense(x, W, cmvn=None):
    return FbankDenseFn.apply(x, W, cmvn)


# ---------------------------------------------------------------------------------------------
# enhancer mask epilogue: out = sigmoid(proj W^T) * [t < len] * mix
# ---------------------------------------------------------------------------------------------
class MaskFcFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, proj, W, mix, lens_dev, T):
        _need_gpu(proj)
        p2 = _f32(proj).view(-1, proj.shape[-1])
        M, K = p2.shape
        N = W.shape[0]
        mix = _f32(mix)
        out = empty((M, N), p2)
        mask = empty((M, N), p2)
        gemm(p2, W, out, M, N, K, transb=True, act=lib.ACT_SIGMOID_MASK_MUL, mul=mix, mask_out=mask, lens=lens_dev, T=T)
        ctx.W = W
        ctx.save_for_backward(p2, mix, mask)
        ctx.oshape = proj.shape[:-1] + (N,)
        return out.view(ctx.oshape), mask.view(ctx.oshape)

    @staticmethod
    def backward(ctx, dout, dmask_unused):
        p2, mix, mask = ctx.saved_tensors
        W = ctx.W
        M, K = p2.shape
        N = W.shape[0]
        dout = _f32(dout)
        if N % 4 and M >= 2048 and K % 4 == 0 and W.is_contiguous() and dout.is_contiguous():
            # odd width (the 257 STFT bins): d(linear) with rows zero-padded to a multiple of 4 floats, both products over the padded
            # width on the engine's 16-byte-load path (the same move as for the CTC projection, _linear_backward_padded)
            Np = (N + 3) // 4 * 4
            dlin = empty((M, Np), p2)
            call('re2e_mask_mul_bwd_ld', dout.data_ptr(), mix.data_ptr(), mask.data_ptr(), dlin.data_ptr(), M, N, Np)
            dp = None
            if ctx.needs_input_grad[0]:
                wt = zeros((K, Np), p2)
                wt[:, :N].copy_(W.t())
                dp = empty((M, K), p2)
                gemm(dlin, wt, dp, M, K, Np, transb=True)
                dp = dp.view(ctx.oshape[:-1] + (K,))
            if ctx.needs_input_grad[1]:
                with param_grads(dlin, p2):
                    tmp = empty((Np, K), p2)
                    gemm(dlin, p2, tmp, Np, K, M, transa=True)
                    with accumulate(W) as (gw, beta):
                        if beta:
                            gw.add_(tmp[:N])
                        else:
                            gw.copy_(tmp[:N])
            return dp, None, None, None, None
        dlin = empty((M, N), p2)
        call('re2e_mask_mul_bwd', dout.data_ptr(), mix.data_ptr(), mask.data_ptr(), dlin.data_ptr(), dlin.numel())
        dp = None
        if ctx.needs_input_grad[0]:
            dp = empty((M, K), p2)
            gemm_input_grad(dlin, W, dp, M, K, N)
            dp = dp.view(ctx.oshape[:-1] + (K,))
        if ctx.needs_input_grad[1]:
            with param_grads(dlin, p2), accumulate(W) as (gw, beta):
                gemm(dlin, p2, gw, N, K, M, transa=True, beta=beta)
        return dp, None, None, None, None


mask_fc = MaskFcFn.apply


# ---------------------------------------------------------------------------------------------
# K10 mean losses
# ---------------------------------------------------------------------------------------------
class MeanLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b, target, kind):
        _need_gpu(a)
        a = _f32(a)
        b = _f32(b) if b is not None else None
        n = a.numel()
        out = empty((1,), a)
        wsb = query('re2e_reduce_workspace_bytes', n)
        ws = workspace(wsb, a.device, 'reduce')
        call('re2e_loss_fwd', a.data_ptr(), ptr(b), float(target), n, kind, out.data_ptr(), ws.data_ptr(), wsb)
        ctx.save_for_backward(a, b)
        ctx.target, ctx.kind = float(target), kind
        return out.view(())

    @staticmethod
    def backward(ctx, g):
        a, b = ctx.saved_tensors
        g = _f32(g).reshape(1)
        da = empty(a.shape, a)
        call('re2e_loss_bwd', a.data_ptr(), ptr(b), ctx.target, a.numel(), ctx.kind, g.data_ptr(), 1.0, da.data_ptr(), 0.0)
        return da, None, None, None


def mean_loss(a, b=None, target=0.0, kind=lib.LOSS_L2):
    return MeanLossFn.apply(a, b, target, kind)


# ---------------------------------------------------------------------------------------------
# K5/K9 convolution (NHWC) + bias + activation
# ---------------------------------------------------------------------------------------------
def _conv_out(h, k, s, p):
    return (h + 2 * p - k) // s + 1


WINOGRAD = lib.exp_env('RE2E_NO_WINOGRAD') is None
WINO_WGRAD = lib.exp_env('RE2E_NO_WINO_WGRAD') is None      # A/B switch (RE2E_EXPERIMENTS=1): the direct halo-patch / engine kernels instead


_WINO_DECLINED = set()


def _wino_note(kind, N, H, Wd, C, Cout):
    """One line on stderr the first time a 3x3 / 4x4 layer that looks like a Winograd layer is declined for its SIZE (>= 2 GiB tensors:
    B >= 64 per GPU at T = 800): the direct kernels that take it over are 1.5 - 2x slower and nothing else would say so."""
    if kind not in _WINO_DECLINED:
        _WINO_DECLINED.add(kind)
        import sys
        print('[re2e] note: %s convolution %dx%dx%d, %d -> %d channels is left to the direct kernels (tensors of 2 GiB or more: the '
              'fused Winograd kernels address with 31-bit offsets); shard the batch further to get them back' % (kind, N, H, Wd, C, Cout),
              file=sys.stderr, flush=True)


def _wino_ok(N, H, Wd, C, Cout, k, stride, pad):
    """3x3 / stride-1 / pad-1 layers the fused Winograd F(2x2,3x3) kernel covers (re2e_conv3x3_wino: C % 8 == 0, Cout / 64 a power of
    two -- 192 or 320 output channels stay with the direct engine).  The kernels address with 31-bit offsets; a tensor of 2 GiB or more
    (a per-GPU batch of 64: 128 images x 800 x 80 x 64 channels) is cut along the image axis by the callers (``_wino_images``)."""
    g = Cout // 64
    if not (WINOGRAD and k == (3, 3) and stride == 1 and pad == 1 and C % 8 == 0 and Cout % 64 == 0 and g & (g - 1) == 0):
        return False
    if H * Wd * max(C, Cout) * 4 >= 2 ** 31 - 256:            # one image alone is too large
        _wino_note('3x3', N, H, Wd, C, Cout)
        return False
    return True


def _wino_images(N, H, Wd, C, Cout):
    """Images per launch of the fused Winograd kernels: all of them, or as many as keep both tensors under 2 GiB."""
    return max(1, min(N, (2 ** 31 - 256) // (H * Wd * max(C, Cout) * 4)))


class RowLims(object):
    """Row limits of one 3x3 layer of a ragged image batch (VGG2L.conv_stack): ``out`` / ``inp`` = int32 device tensors, per image, of the rows
    of the layer's output / input that are computed (the rest is zero where someone reads it), ``out_tail`` / ``inp_tail`` = the largest
    number of rows beyond them in the batch (sizes the fill launches)."""
    __slots__ = ('out', 'inp', 'out_tail', 'inp_tail')

    def __init__(self, out, inp, out_tail, inp_tail):
        self.out, self.inp, self.out_tail, self.inp_tail = out, inp, out_tail, inp_tail


def fill_image_rows(t, lim, div, max_tail):
    """zeros in rows >= ceil(lim[n] / div) of the NHWC tensor ``t`` (see re2e_fill_image_rows)."""
    N, H = t.shape[0], t.shape[1]
    if max_tail > 0:
        call('re2e_fill_image_rows', t.data_ptr(), N, H, t.shape[2] * t.shape[3], lim.data_ptr(), div, int(max_tail), 0.0)


_LOG_CALLS = os.environ.get('RE2E_IGEMM_LOG') is not None


def _note_row_limits(row_lim, H=0, W=0):
    """bench.py's executed-FLOP meter (flops.py) counts a row-limited launch with the rows below its limits: the host copy of the limits.  With
    RE2E_IGEMM_LOG (tools/igemm_table.py) the pixels the next Winograd launch really computes go to the engine's call log."""
    if row_lim is None or (lib.FLOP_METER is None and not _LOG_CALLS):
        return
    from .model import e2e_common as ec
    host = ec.host_lens_of(row_lim)
    if host is None:
        return
    if lib.FLOP_METER is not None:
        from . import flops
        flops.note_rows(row_lim.data_ptr(), host)
    if _LOG_CALLS and H:
        import sys
        sys.stderr.write('[igemm-rows] M=%d\n' % (sum(min(H, (min(int(l), H) + 7) // 8 * 8) for l in host) * W))
        sys.stderr.flush()


def conv3x3_wino(x, W, Cout, dgrad=False, bias=None, relu=False, mask=None, pool=False, row_lim=None):
    """re2e_conv3x3_wino on NHWC ``x`` with the layer's weight ``W`` in PyTorch layout: forward (dgrad=False: bias / ReLU / fused
    2x2 max pool -> (pooled, index bytes)) or data gradient (dgrad=True: ``x`` is dy; ``mask``: the ReLU output in front).  Tensors of
    2 GiB or more run as several launches over slices of the image axis (same stream, same workspace: the launches are ordered)."""
    N, H, Wd, C = x.shape
    wsb = query('re2e_conv3x3_wino_workspace_bytes', C, Cout)
    ws = workspace(wsb, x.device, 'wino')
    nb = _wino_images(N, H, Wd, C, Cout)
    _note_row_limits(row_lim, H, Wd)
    if pool:
        yp = empty((N, (H + 1) // 2, (Wd + 1) // 2, Cout), x)
        idx = torch.empty(yp.shape, dtype=torch.uint8, device=x.device)
        for i in range(0, N, nb):
            n = min(nb, N - i)
            if row_lim is not None:
                call('re2e_conv3x3_wino_rows', x[i:i + n].data_ptr(), n, H, Wd, C, W.data_ptr(), Cout, 0, ptr(bias), 1, None, None,
                     yp[i:i + n].data_ptr(), idx[i:i + n].data_ptr(), row_lim.data_ptr() + 4 * i, ws.data_ptr(), wsb)
            else:
                call('re2e_conv3x3_wino', x[i:i + n].data_ptr(), n, H, Wd, C, W.data_ptr(), Cout, 0, ptr(bias), 1, None, None, yp[i:i + n].data_ptr(),
                     idx[i:i + n].data_ptr(), ws.data_ptr(), wsb)
        return yp, idx
    y = empty((N, H, Wd, Cout), x)
    for i in range(0, N, nb):
        n = min(nb, N - i)
        if row_lim is not None:
            call('re2e_conv3x3_wino_rows', x[i:i + n].data_ptr(), n, H, Wd, C, W.data_ptr(), Cout, int(bool(dgrad)), ptr(bias), int(bool(relu)),
                 None if mask is None else mask[i:i + n].data_ptr(), y[i:i + n].data_ptr(), None, None, row_lim.data_ptr() + 4 * i, ws.data_ptr(), wsb)
        else:
            call('re2e_conv3x3_wino', x[i:i + n].data_ptr(), n, H, Wd, C, W.data_ptr(), Cout, int(bool(dgrad)), ptr(bias), int(bool(relu)),
                 None if mask is None else mask[i:i + n].data_ptr(), y[i:i + n].data_ptr(), None, None, ws.data_ptr(), wsb)
    return y


def _wino44_ok(N, H, Wd, C, Cout, k, stride, pad):
    """4x4 / stride-1 / pad-1 layers with enough channels and pixels for Winograd F(2x2,4x4) to pay (re2e_conv4x4_wino: the
    discriminator's conv4 and its data gradient -- both directions need C % 16 == 0 and Cout % 16 == 0)."""
    return (WINOGRAD and k == (4, 4) and stride == 1 and pad == 1 and C % 16 == 0 and Cout % 16 == 0 and min(C, Cout) >= 64
            and N * H * Wd >= 4096 and H >= 3 and Wd >= 3)


def conv4x4_wino(x, W, Cout, pad, dgrad=False):
    """re2e_conv4x4_wino on NHWC ``x`` with the layer's weight ``W`` in PyTorch layout: forward (pad = the layer's padding) or data
    gradient (``x`` is dy, ``Cout`` the layer's input channels, pad = 3 - the layer's padding)."""
    N, H, Wd, C = x.shape
    wsb = query('re2e_conv4x4_wino_workspace_bytes', N, H, Wd, C, Cout, pad)
    ws = workspace(wsb, x.device, 'wino44')
    y = empty((N, H + 2 * pad - 3, Wd + 2 * pad - 3, Cout), x)
    call('re2e_conv4x4_wino', x.data_ptr(), N, H, Wd, C, W.data_ptr(), Cout, pad, int(bool(dgrad)), y.data_ptr(), ws.data_ptr(), wsb)
    return y


class Conv2dFn(torch.autograd.Function):
    """x: (N,H,W,Cin) NHWC; W: (Cout,Cin,KH,KW) PyTorch layout; returns (N,OH,OW,Cout)."""

    @staticmethod
    def forward(ctx, x, W, b, stride, pad, act, act_bwd_done=False, x_is_relu_out=False, pool=False, lims=None):
        _need_gpu(x)
        x = _f32(x)
        N, H, Wd, Cin = x.shape
        Cout, _, KH, KW = W.shape
        OH, OW = _conv_out(H, KH, stride, pad), _conv_out(Wd, KW, stride, pad)
        ctx.act_bwd_done = bool(act_bwd_done or pool)  # the ONLY consumer (maxpool2(relu_in=True) / conv2d(x_is_relu_out=True)) returns d(pre-activation)
        ctx.x_is_relu_out = bool(x_is_relu_out)        # x = ReLU output of the layer in front: dx is taken through that ReLU (dx = 0 where x <= 0)
        ctx.pool = bool(pool)                          # the result is maxpool2(relu(conv)), 2x2 / stride 2 / ceil mode
        ctx.W, ctx.b, ctx.cfg = W, b, (stride, pad, act)
        ctx.lims = None
        if act in (lib.ACT_NONE, lib.ACT_RELU) and _wino_ok(N, H, Wd, Cin, Cout, (KH, KW), stride, pad) and W.is_contiguous():
            # 3x3 / stride-1 VGG layers: fused Winograd F(2x2,3x3), 2.25x fewer matrix instructions than the direct kernels below
            # ``lims`` (RowLims, ragged image batches): rows beyond an utterance's reach are not computed; zeros where a later kernel reads all rows
            ctx.lims = lims
            lo = lims.out if lims is not None else None
            if pool:
                yp, idx = conv3x3_wino(x, W, Cout, bias=b, relu=True, pool=True, row_lim=lo)
                if lims is not None:
                    fill_image_rows(yp, lims.out, 2, (lims.out_tail + 1) // 2)
                ctx.save_for_backward(x, idx)
                return yp
            y = conv3x3_wino(x, W, Cout, bias=b, relu=act == lib.ACT_RELU, row_lim=lo)
            if lims is not None:
                fill_image_rows(y, lims.out, 1, lims.out_tail)
            ctx.save_for_backward(x, y if (act != lib.ACT_NONE and not ctx.act_bwd_done) else None)
            return y
        if act == lib.ACT_NONE and b is None and not pool and _wino44_ok(N, H, Wd, Cin, Cout, (KH, KW), stride, pad) and W.is_contiguous():
            y = conv4x4_wino(x.contiguous(), W, Cout, pad)            # the discriminator's conv4: F(2x2,4x4), 2.56x fewer matrix FLOPs
            ctx.save_for_backward(x, None)
            return y
        wg = empty((Cout, KH, KW, Cin), x)
        call('re2e_conv_weight_gather', W.data_ptr(), wg.data_ptr(), Cout, Cin, KH, KW, 0, KH, KW, 0, 0, 1)
        if pool:
            # one launch: only the pooled activation and its index bytes are written (re2e_conv3x3_relu_pool); geometries the fused
            # kernel does not cover run the convolution and the pool (with the ReLU mask in its index byte) one after the other
            yp = empty((N, (OH + 1) // 2, (OW + 1) // 2, Cout), x)
            idx = torch.empty(yp.shape, dtype=torch.uint8, device=x.device)
            fused = KH == 3 and KW == 3 and stride == 1 and pad == 1 and Cin % 16 == 0 and Cout % 64 == 0 and x.numel() * 4 < 2 ** 31 - 256 \
                and N * OH * OW * Cout * 4 < 2 ** 31 - 256
            if fused:
                call('re2e_conv3x3_relu_pool', x.data_ptr(), N, H, Wd, Cin, wg.data_ptr(), Cout, ptr(b), yp.data_ptr(), idx.data_ptr())
            else:
                y = empty((N, OH, OW, Cout), x)
                call('re2e_conv_igemm', x.data_ptr(), N, H, Wd, Cin, wg.data_ptr(), Cout, KH, KW, OH, OW, stride, stride, 1, 1, -pad, -pad,
                     y.data_ptr(), OH, OW, 1, 1, 0, 0, ptr(b), act, 0.0)
                call('re2e_maxpool2_fwd', y.data_ptr(), N, OH, OW, Cout, yp.data_ptr(), idx.data_ptr(), 1)
            ctx.save_for_backward(x, idx)
            return yp
        y = empty((N, OH, OW, Cout), x)
        call('re2e_conv_igemm', x.data_ptr(), N, H, Wd, Cin, wg.data_ptr(), Cout, KH, KW, OH, OW, stride, stride, 1, 1, -pad, -pad,
             y.data_ptr(), OH, OW, 1, 1, 0, 0, ptr(b), act, 0.0)
        ctx.save_for_backward(x, y if (act != lib.ACT_NONE and not ctx.act_bwd_done) else None)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, y = ctx.saved_tensors
        W, b = ctx.W, ctx.b
        stride, pad, act = ctx.cfg
        if ctx.act_bwd_done:
            act = lib.ACT_NONE                       # dy already is d(pre-activation): no pass over dy and y, the bias gradient is a
            #                                          plain column sum on the weight-gradient stream
        N, H, Wd, Cin = x.shape
        Cout, _, KH, KW = W.shape
        OH, OW = _conv_out(H, KH, stride, pad), _conv_out(Wd, KW, stride, pad)
        if ctx.pool:                                  # y holds the pool's index bytes: scatter the pooled gradient (ReLU mask included)
            full = empty((N, OH, OW, Cout), x)
            call('re2e_maxpool2_bwd', _f32(dy).contiguous().data_ptr(), y.data_ptr(), N, OH, OW, Cout, full.data_ptr())
            dy, y = full, None
        need_w, need_b = _wants(ctx, 1, W), _wants(ctx, 2, b)      # fixed at graph construction (see LinearFn)
        dz, bias_done = act_bwd_bias(_f32(dy).reshape(N * OH * OW, Cout), y, act, b, need_b)
        dz = dz.view(N, OH, OW, Cout)
        dx = None
        lims = ctx.lims
        if ctx.needs_input_grad[0]:
            dx = conv_dgrad(dz, W, (N, H, Wd, Cin), stride, pad, relu_out=x if ctx.x_is_relu_out else None, row_lim=lims.inp if lims is not None else None)
            if lims is not None and lims.inp is not None:
                fill_image_rows(dx, lims.inp, 1, lims.inp_tail)         # the gradient is zero there: written, not computed
        with param_grads(dz, x, lims):
            if need_w and WINO_WGRAD and _wino_ok(N, H, Wd, Cin, Cout, (KH, KW), stride, pad) and Cin % 64 == 0 and x.is_contiguous() \
                    and dz.is_contiguous():
                # 3x3 / stride-1 VGG layers: the sum over pixels in the Winograd domain (re2e_conv3x3_wino_wgrad), 2.25x fewer matrix FLOPs
                nb = _wino_images(N, H, Wd, Cin, Cout)                # tensors of 2 GiB or more: slices of the image axis, accumulated
                wsb = query('re2e_conv3x3_wino_wgrad_workspace_bytes', nb, H, Wd, Cin, Cout)
                ws = workspace(wsb, x.device, 'winow')
                _note_row_limits(lims.out if lims is not None else None, H, Wd)
                with accumulate(W) as (gw, beta):
                    for i in range(0, N, nb):
                        n = min(nb, N - i)
                        if lims is not None:      # dz is zero in rows >= lims.out: patches there are skipped
                            call('re2e_conv3x3_wino_wgrad_rows', x[i:i + n].data_ptr(), n, H, Wd, Cin, dz[i:i + n].data_ptr(), Cout, gw.data_ptr(),
                                 beta if i == 0 else 1.0, lims.out.data_ptr() + 4 * i, ws.data_ptr(), wsb)
                        else:
                            call('re2e_conv3x3_wino_wgrad', x[i:i + n].data_ptr(), n, H, Wd, Cin, dz[i:i + n].data_ptr(), Cout, gw.data_ptr(),
                                 beta if i == 0 else 1.0, ws.data_ptr(), wsb)
            elif need_w and _wino44_ok(N, H, Wd, Cin, Cout, (KH, KW), stride, pad) and x.is_contiguous() and dz.is_contiguous():
                wsb = query('re2e_conv4x4_wino_wgrad_workspace_bytes', N, H, Wd, Cin, Cout, pad)
                ws = workspace(wsb, x.device, 'wino44w')
                with accumulate(W) as (gw, beta):
                    call('re2e_conv4x4_wino_wgrad', x.data_ptr(), N, H, Wd, Cin, dz.data_ptr(), Cout, pad, gw.data_ptr(), beta, ws.data_ptr(), wsb)
            elif need_w:
                wsb = query('re2e_conv_wgrad_workspace_bytes', N, OH, OW, Cin, Cout, KH, KW)
                ws = workspace(wsb, x.device, 'wgrad')
                with accumulate(W) as (gw, beta):
                    call('re2e_conv_wgrad', x.data_ptr(), N, H, Wd, Cin, dz.data_ptr(), Cout, KH, KW, OH, OW, stride, stride, -pad, -pad,
                         gw.data_ptr(), beta, ws.data_ptr(), wsb)
            if b is not None and need_b and not bias_done:
                with accumulate(b) as (gb, beta):
                    colsum_into(dz, N * OH * OW, Cout, gb, beta)
        return dx, None, None, None, None, None, None, None, None, None


def conv_dgrad(dz, W, xshape, stride, pad, relu_out=None, row_lim=None):
    """Data gradient of an NHWC convolution (see re2e_conv_igemm).  ``relu_out`` (stride 1): the convolution's input, which was
    the ReLU output of the layer in front -- the gradient is taken through that ReLU as well (re2e_conv_igemm_masked)."""
    N, H, Wd, Cin = xshape
    Cout, _, KH, KW = W.shape
    OH, OW = dz.shape[1], dz.shape[2]
    if stride == 1 and _wino_ok(N, H, Wd, Cout, Cin, (KH, KW), stride, pad) and (OH, OW) == (H, Wd) and W.is_contiguous():
        return conv3x3_wino(_f32(dz), W, Cin, dgrad=True, mask=relu_out, row_lim=row_lim)
    if relu_out is None and _wino44_ok(N, OH, OW, Cout, Cin, (KH, KW), stride, pad) and (OH, OW) == (H - 1, Wd - 1) and W.is_contiguous():
        return conv4x4_wino(_f32(dz).contiguous(), W, Cin, 3 - pad, dgrad=True)
    if stride == 1:
        wt = empty((Cin, KH, KW, Cout), dz)
        call('re2e_conv_weight_gather', W.data_ptr(), wt.data_ptr(), Cout, Cin, KH, KW, 1, KH, KW, 0, 0, 1)
        dx = empty((N, H, Wd, Cin), dz)
        if relu_out is not None:
            call('re2e_conv_igemm_masked', dz.data_ptr(), N, OH, OW, Cout, wt.data_ptr(), Cin, KH, KW, H, Wd, 1, 1, -1, -1, pad, pad,
                 dx.data_ptr(), H, Wd, 1, 1, 0, 0, relu_out.data_ptr())
        else:
            call('re2e_conv_igemm', dz.data_ptr(), N, OH, OW, Cout, wt.data_ptr(), Cin, KH, KW, H, Wd, 1, 1, -1, -1, pad, pad,
                 dx.data_ptr(), H, Wd, 1, 1, 0, 0, None, lib.ACT_NONE, 0.0)
        return dx
    if relu_out is not None:
        raise lib.Re2eError('conv_dgrad: relu_out needs stride 1')
    if stride != 2 or KH % 2 or KW % 2:
        raise lib.Re2eError('conv data gradient supports stride 1, or stride 2 with even kernels')
    if Cin != 1:    # all four output parity classes in one launch
        dx = empty((N, H, Wd, Cin), dz)
        wt = empty((4, Cin, KH // 2, KW // 2, Cout), dz)
        call('re2e_conv_dgrad_s2', dz.data_ptr(), N, OH, OW, Cout, W.data_ptr(), Cin, KH, KW, H, Wd, pad, dx.data_ptr(), wt.data_ptr())
        return dx
    # Cin == 1 (thin direct kernel): one launch per output parity class (ph,pw); taps a -> kh = 2a + ((ph+pad) % 2)
    dx = torch.zeros((N, H, Wd, Cin), dtype=torch.float32, device=dz.device) if (H % 2 or Wd % 2) else empty((N, H, Wd, Cin), dz)
    TA, TB = KH // 2, KW // 2
    wt = empty((Cin, TA, TB, Cout), dz)
    for ph in range(2):
        for pw in range(2):
            PH, PW = (H - ph + 1) // 2, (Wd - pw + 1) // 2
            if PH <= 0 or PW <= 0:
                continue
            kh0, kw0 = (ph + pad) % 2, (pw + pad) % 2
            call('re2e_conv_weight_gather', W.data_ptr(), wt.data_ptr(), Cout, Cin, KH, KW, 1, TA, TB, kh0, kw0, 2)
            # oh = (2i + ph + pad - kh)/2 = i + (ph + pad - kh0)/2 - a
            oy0, ox0 = (ph + pad - kh0) // 2, (pw + pad - kw0) // 2
            call('re2e_conv_igemm', dz.data_ptr(), N, OH, OW, Cout, wt.data_ptr(), Cin, TA, TB, PH, PW, 1, 1, -1, -1, oy0, ox0,
                 dx.data_ptr(), H, Wd, 2, 2, ph, pw, None, lib.ACT_NONE, 0.0)
    return dx


def conv2d(x, W, b=None, stride=1, pad=1, act=None, relu_bwd_in_pool=False, relu_bwd_in_next=False, x_is_relu_out=False, pool=False, lims=None):
    """``relu_bwd_in_pool`` / ``relu_bwd_in_next``: act is 'relu' and the result goes ONLY into ``maxpool2(y, relu_in=True)`` /
    ``conv2d(y, ..., x_is_relu_out=True)``, whose backward applies the ReLU's derivative (a pooled maximum <= 0 passes nothing back;
    the next convolution's data gradient is masked by its input > 0 in the kernel's epilogue): this convolution's backward then
    skips the pass that would read dy and y and write dz (1.6 GB for a 64-channel VGG layer at config 4) and takes its bias
    gradient as a column sum of dy.  ``x_is_relu_out``: the counterpart flag on the consuming convolution (stride 1).
    ``pool``: return maxpool2(relu(conv(x))) (2x2, stride 2, ceil mode) -- for 3x3 / stride-1 / pad-1 layers with C % 16 == 0 and
    Cout % 64 == 0 in ONE launch that never writes the full-resolution activation (re2e_conv3x3_relu_pool)."""
    if (relu_bwd_in_pool or relu_bwd_in_next) and act != 'relu':
        raise lib.Re2eError('relu_bwd_in_pool / relu_bwd_in_next need act="relu"')
    if x_is_relu_out and stride != 1:
        raise lib.Re2eError('x_is_relu_out needs stride 1')
    if pool and act != 'relu':
        raise lib.Re2eError('pool=True is conv -> ReLU -> maxpool2: act must be "relu"')
    return Conv2dFn.apply(x, W, b, stride, pad, ACT[act], relu_bwd_in_pool or relu_bwd_in_next, x_is_relu_out, pool, lims)


class ConvTranspose2dFn(torch.autograd.Function):
    """nn.ConvTranspose2d(C1, C2, k, stride 2, padding p) over NHWC (U-Net up-convolutions, enhance_model.py:266-283).
    x: (N,H,W,C1); Wt: (C1,C2,KH,KW) (ConvTranspose2d's own layout = a Conv2d weight with Cout=C1, Cin=C2); returns
    (N,2H,2W,C2).  A transposed convolution IS the data gradient of the convolution with that weight, so the three passes are
    the convolution's three kernels with their roles rotated: forward = stride-2 data gradient (re2e_conv_dgrad_s2),
    input gradient = stride-2 forward convolution of dy, weight gradient = re2e_conv_wgrad with (input, output-gradient) =
    (dy, x)."""

    @staticmethod
    def forward(ctx, x, Wt, b, stride, pad):
        _need_gpu(x)
        x = _f32(x)
        N, H, Wd, C1 = x.shape
        _, C2, KH, KW = Wt.shape
        if stride != 2 or KH % 2 or KW % 2 or KH != 2 * pad + 2:
            raise lib.Re2eError('conv_transpose2d is built for kernel 2p+2, stride 2 (output exactly 2x the input)')
        OH, OW = 2 * H, 2 * Wd
        y = conv_dgrad(x, Wt, (N, OH, OW, C2), stride, pad)
        if b is not None:
            ones, yb = torch.ones(C2, device=x.device), empty(y.shape, y)
            call('re2e_affine_cols', y.data_ptr(), b.data_ptr(), ones.data_ptr(), yb.data_ptr(), N * OH * OW, C2)      # (y + b) * 1
            y = yb
        ctx.Wt, ctx.b, ctx.cfg = Wt, b, (stride, pad)
        ctx.save_for_backward(x)
        return y

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        Wt, b = ctx.Wt, ctx.b
        stride, pad = ctx.cfg
        N, H, Wd, C1 = x.shape
        _, C2, KH, KW = Wt.shape
        OH, OW = 2 * H, 2 * Wd
        dy = _f32(dy).contiguous()
        dx = None
        if ctx.needs_input_grad[0]:
            wg = empty((C1, KH, KW, C2), dy)
            call('re2e_conv_weight_gather', Wt.data_ptr(), wg.data_ptr(), C1, C2, KH, KW, 0, KH, KW, 0, 0, 1)
            dx = empty((N, H, Wd, C1), dy)
            call('re2e_conv_igemm', dy.data_ptr(), N, OH, OW, C2, wg.data_ptr(), C1, KH, KW, H, Wd, stride, stride, 1, 1, -pad, -pad,
                 dx.data_ptr(), H, Wd, 1, 1, 0, 0, None, lib.ACT_NONE, 0.0)
        with param_grads(dy, x):
            if _wants(ctx, 1, Wt):
                wsb = query('re2e_conv_wgrad_workspace_bytes', N, H, Wd, C2, C1, KH, KW)
                ws = workspace(wsb, x.device, 'wgrad')
                with accumulate(Wt) as (gw, beta):
                    call('re2e_conv_wgrad', dy.data_ptr(), N, OH, OW, C2, x.data_ptr(), C1, KH, KW, H, Wd, stride, stride, -pad, -pad,
                         gw.data_ptr(), beta, ws.data_ptr(), wsb)
            if b is not None and _wants(ctx, 2, b):
                with accumulate(b) as (gb, beta):
                    colsum_into(dy.view(N * OH * OW, C2), N * OH * OW, C2, gb, beta)
        return dx, None, None, None, None


def conv_transpose2d(x, Wt, b=None, stride=2, pad=1):
    return ConvTranspose2dFn.apply(x, Wt, b, stride, pad)


class ActFn(torch.autograd.Function):
    """Stand-alone activation module (nn.LeakyReLU(0.2) / nn.ReLU / nn.Sigmoid of the U-Net blocks)."""

    @staticmethod
    def forward(ctx, x, act):
        _need_gpu(x)
        x = _f32(x).contiguous()
        y = empty(x.shape, x)
        call('re2e_act_fwd', x.data_ptr(), y.data_ptr(), x.numel(), act)
        ctx.act = act
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, dy):
        (y,) = ctx.saved_tensors
        dy = _f32(dy).contiguous()
        dz = empty(dy.shape, dy)
        call('re2e_act_bwd', dy.data_ptr(), y.data_ptr(), dz.data_ptr(), dy.numel(), ctx.act)
        return dz, None


def activation(x, act):
    return ActFn.apply(x, ACT[act])


class MaxPool2Fn(torch.autograd.Function):
    """2x2 / stride-2 ceil-mode max pool over NHWC.  ``relu_in``: x is the output of ``conv2d(..., act='relu',
    relu_bwd_in_pool=True)`` -- the pool's backward then returns the gradient of that ReLU's INPUT (windows with a maximum <= 0
    pass nothing back) and the convolution's backward skips its own activation pass."""

    @staticmethod
    def forward(ctx, x, relu_in):
        _need_gpu(x)
        x = _f32(x)
        N, H, W, C = x.shape
        y = empty((N, (H + 1) // 2, (W + 1) // 2, C), x)
        idx = torch.empty(y.shape, dtype=torch.uint8, device=x.device)
        call('re2e_maxpool2_fwd', x.data_ptr(), N, H, W, C, y.data_ptr(), idx.data_ptr(), 1 if relu_in else 0)
        ctx.save_for_backward(idx)
        ctx.xshape = x.shape
        return y

    @staticmethod
    def backward(ctx, dy):
        (idx,) = ctx.saved_tensors
        N, H, W, C = ctx.xshape
        dy = _f32(dy)
        dx = empty(ctx.xshape, dy)
        call('re2e_maxpool2_bwd', dy.data_ptr(), idx.data_ptr(), N, H, W, C, dx.data_ptr())
        return dx, None


def maxpool2(x, relu_in=False):
    return MaxPool2Fn.apply(x, relu_in)


class VggPackFn(torch.autograd.Function):
    """NHWC branches [(N_k,T,Fq,C)] -> ONE time-major (T, sum N_k, C*Fq) tensor, frames >= lens zeroed
    (e2e_encoder.py:272-278).  ``lens_list`` holds one int32 device tensor per branch."""

    @staticmethod
    def forward(ctx, lens_list, *xs):
        xs = [_f32(x) for x in xs]
        _, T, Fq, C = xs[0].shape
        Ntot = sum(x.shape[0] for x in xs)
        y = empty((T, Ntot, C * Fq), xs[0])
        off = 0
        for x, ld in zip(xs, lens_list):
            call('re2e_vgg_pack_fwd', x.data_ptr(), ld.data_ptr(), x.shape[0], T, Fq, C, y.data_ptr(), Ntot, off)
            off += x.shape[0]
        ctx.lens, ctx.shapes, ctx.Ntot = lens_list, [x.shape for x in xs], Ntot
        return y

    @staticmethod
    def backward(ctx, dy):
        dy = _f32(dy)
        outs, off = [], 0
        for k, (shp, ld) in enumerate(zip(ctx.shapes, ctx.lens)):
            N, T, Fq, C = shp
            if ctx.needs_input_grad[1 + k]:
                dx = empty(shp, dy)
                call('re2e_vgg_pack_bwd', dy.data_ptr(), ld.data_ptr(), N, T, Fq, C, dx.data_ptr(), ctx.Ntot, off)
                outs.append(dx)
            else:
                outs.append(None)
            off += N
        return (None,) + tuple(outs)


def vgg_pack(x, lens_dev):
    return VggPackFn.apply([lens_dev], x)


def vgg_pack_multi(xs, lens_list):
    return VggPackFn.apply(list(lens_list), *xs)


def _sync_world():
    from . import dist as rdist
    return rdist.world_size() if SYNC_BN else 1


_SYNC_BN_RAGGED = {}       # device -> fp64 scalar: BatchNorm calls of this process whose row count differed between the ranks


def _sync_rows_poison(Pn, dev):
    """Synchronised BatchNorm takes the global row count as rows x world (no per-step host sync for a count): true for equal shards only.
    Every call MAX-reduces the exact pair (rows, -rows) as fp64 -- the same tiny collective on every rank, whatever each rank has seen
    before -- and everything is decided from the REDUCED pair alone, on the device: max rows != min rows adds 1 to a per-device counter and
    returns NaN (0.0 otherwise), which the caller adds to the batch mean, so ragged shards can never produce an update: the NaN reaches the
    gradient norm of every replica through the gradient average, the NaN gate skips the step everywhere, and ``check_sync_bn`` -- called
    where the trainers read their meters back anyway (``JointTrainer.to_floats``) -- raises on EVERY rank in that same step.
    (Round 4 carried rows and rows^2 in the fp32 sums and let each rank decide from its own history: rows^2 is not exact in fp32 beyond
    4096 odd rows, and a rank that had seen its shape before skipped the check, so one rank raised alone and its peers hung.)"""
    from . import dist as rdist
    pair = rdist.allreduce_max_(torch.tensor([float(Pn), -float(Pn)], dtype=torch.float64).to(dev, non_blocking=True))
    ragged = (pair[0] + pair[1]) != 0
    cnt = _SYNC_BN_RAGGED.get(str(dev))
    if cnt is None:
        cnt = _SYNC_BN_RAGGED[str(dev)] = torch.zeros((), dtype=torch.float64, device=dev)
    cnt.add_(ragged.to(torch.float64))
    return torch.where(ragged, float('nan'), 0.0).to(torch.float32)


def sync_bn_flags():
    """The per-device counters of ``_sync_rows_poison`` (device scalars; empty when synchronised BatchNorm never ran)."""
    return list(_SYNC_BN_RAGGED.values())


def check_sync_bn(counts=None):
    """Raise if a synchronised BatchNorm met ragged shards.  ``counts``: the counters' values if the caller has read them back already
    (with its meters); otherwise they are read here (a host sync).  The counters hold REDUCED information, so all ranks raise together."""
    if counts is None:
        counts = [float(c.item()) for c in _SYNC_BN_RAGGED.values()]
    n = int(sum(counts))
    if n:
        for c in _SYNC_BN_RAGGED.values():
            c.zero_()
        raise lib.Re2eError('synchronised BatchNorm needs the same number of rows on every rank (%d BatchNorm calls saw ranks with different '
                            'row counts; their statistics were poisoned with NaN and no update was applied): shard equal utterance '
                            'counts of equal padded length, or leave opt.sync_bn off' % n)


class BnLreluFn(torch.autograd.Function):
    """BatchNorm2d (train-mode statistics, running-stat update) + LeakyReLU(0.2) over NHWC.  With ``SYNC_BN`` in a data-parallel run the
    statistics are those of the GLOBAL batch: three small all-reduces per layer and step (sum x; sum (x - mean)^2; sum dz | sum dz xhat),
    torch.nn.SyncBatchNorm's arithmetic -- the parameter gradients stay local sums (the gradient average makes them global)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, rm, rv, train, momentum, eps, slope=0.2):
        _need_gpu(x)
        x = _f32(x)
        C = x.shape[-1]
        Pn = x.numel() // C
        y = empty(x.shape, x)
        sm, si = empty((C,), x), empty((C,), x)
        wsb = query('re2e_bn_workspace_bytes', Pn, C)
        ws = workspace(wsb, x.device, 'bn')
        defer = BN_DEFER_RUNNING and BN_STATS_SINK is not None and train      # momentum 0 keeps running_mean / running_var bit-identical
        world = _sync_world() if train else 1
        ctx.ptot = Pn
        if world > 1:
            from . import dist as rdist
            acc = empty((C,), x)
            call('re2e_bn_sync_partial', x.data_ptr(), Pn, C, None, 0, acc.data_ptr(), ws.data_ptr(), wsb)
            rdist.allreduce_sum_(acc)
            ptot = Pn * world                                                 # equal shards (bench.py --scaling strong): no host sync for a count
            mean = (acc / float(ptot) + _sync_rows_poison(Pn, x.device)).contiguous()      # NaN when the shards are NOT equal (checked on the device)
            var = empty((C,), x)
            call('re2e_bn_sync_partial', x.data_ptr(), Pn, C, mean.data_ptr(), 1, var.data_ptr(), ws.data_ptr(), wsb)
            rdist.allreduce_sum_(var)
            var = (var / float(ptot)).contiguous()
            call('re2e_bn_sync_finalize', mean.data_ptr(), var.data_ptr(), ptot, C, 0.0 if defer else float(momentum), float(eps), rm.data_ptr(),
                 rv.data_ptr(), sm.data_ptr(), si.data_ptr())
            call('re2e_bn_apply', x.data_ptr(), Pn, C, sm.data_ptr(), si.data_ptr(), gamma.data_ptr(), beta.data_ptr(), float(slope), y.data_ptr())
            ctx.ptot = ptot
        else:
            call('re2e_bn_lrelu_fwd', x.data_ptr(), Pn, C, gamma.data_ptr(), beta.data_ptr(), rm.data_ptr(), rv.data_ptr(),
                 0.0 if defer else float(momentum), float(eps), int(train), float(slope), y.data_ptr(), sm.data_ptr(), si.data_ptr(), ws.data_ptr(), wsb)
        ctx.gamma, ctx.beta, ctx.train, ctx.slope = gamma, beta, train, float(slope)
        ctx.save_for_backward(x, sm, si)
        if BN_STATS_SINK is not None and train:
            BN_STATS_SINK.append((rm, rv, sm, si, ctx.ptot, float(momentum), float(eps)))
        return y

    @staticmethod
    def backward(ctx, dy):
        x, sm, si = ctx.saved_tensors
        if not ctx.train:
            raise lib.Re2eError('BatchNorm backward is implemented for train mode only')
        gamma, beta = ctx.gamma, ctx.beta
        C = x.shape[-1]
        Pn = x.numel() // C
        dy = _f32(dy)
        dx = empty(x.shape, x)
        wsb = query('re2e_bn_workspace_bytes', Pn, C)
        ws = workspace(wsb, x.device, 'bn')
        if ctx.ptot != Pn:                                                    # synchronised statistics
            from . import dist as rdist
            sums = empty((2 * C,), x)
            call('re2e_bn_sync_bwd_partial', dy.data_ptr(), x.data_ptr(), Pn, C, gamma.data_ptr(), beta.data_ptr(), sm.data_ptr(), si.data_ptr(),
                 ctx.slope, sums.data_ptr(), ws.data_ptr(), wsb)
            if _wants(ctx, 1, gamma):                                         # LOCAL sums: the gradient exchange averages them over the ranks
                with accumulate(gamma) as (dg, gbeta), accumulate(beta) as (db, _):
                    if gbeta:
                        dg.add_(sums[C:]); db.add_(sums[:C])
                    else:
                        dg.copy_(sums[C:]); db.copy_(sums[:C])
            tot = sums.clone()
            rdist.allreduce_sum_(tot)
            tot.mul_(float(Pn) / float(ctx.ptot))                             # the apply kernel divides by its own row count
            call('re2e_bn_sync_bwd_apply', dy.data_ptr(), x.data_ptr(), Pn, C, gamma.data_ptr(), beta.data_ptr(), sm.data_ptr(), si.data_ptr(),
                 ctx.slope, tot.data_ptr(), dx.data_ptr())
            return dx, None, None, None, None, None, None, None, None
        if _wants(ctx, 1, gamma):
            with accumulate(gamma) as (dg, gbeta), accumulate(beta) as (db, _):
                call('re2e_bn_lrelu_bwd', dy.data_ptr(), x.data_ptr(), Pn, C, gamma.data_ptr(), beta.data_ptr(), sm.data_ptr(), si.data_ptr(),
                     ctx.slope, dx.data_ptr(), dg.data_ptr(), db.data_ptr(), gbeta, ws.data_ptr(), wsb)
        else:
            call('re2e_bn_lrelu_bwd', dy.data_ptr(), x.data_ptr(), Pn, C, gamma.data_ptr(), beta.data_ptr(), sm.data_ptr(), si.data_ptr(),
                 ctx.slope, dx.data_ptr(), None, None, 0.0, ws.data_ptr(), wsb)
        return dx, None, None, None, None, None, None, None, None


class InstanceNormLreluFn(torch.autograd.Function):
    """InstanceNorm2d(affine=False, track_running_stats=False) + LeakyReLU(slope) over NHWC (--norm_D instance, gan_model.py:42-46):
    per-sample, per-channel statistics = the BatchNorm kernels on one sample at a time (N launches per pass: an option off the
    benchmarked path, kept simple)."""

    @staticmethod
    def forward(ctx, x, eps, slope):
        _need_gpu(x)
        x = _f32(x).contiguous()
        N, C = x.shape[0], x.shape[-1]
        Pn = x.numel() // (N * C)
        y = empty(x.shape, x)
        sm, si = empty((N, C), x), empty((N, C), x)
        ones, zero = torch.ones(C, device=x.device), torch.zeros(C, device=x.device)
        rm, rv = torch.zeros(C, device=x.device), torch.ones(C, device=x.device)          # momentum 0: never moved, never read back
        wsb = query('re2e_bn_workspace_bytes', Pn, C)
        ws = workspace(wsb, x.device, 'bn')
        st = Pn * C * 4
        for n in range(N):
            call('re2e_bn_lrelu_fwd', x.data_ptr() + n * st, Pn, C, ones.data_ptr(), zero.data_ptr(), rm.data_ptr(), rv.data_ptr(), 0.0, float(eps), 1,
                 float(slope), y.data_ptr() + n * st, sm[n].data_ptr(), si[n].data_ptr(), ws.data_ptr(), wsb)
        ctx.save_for_backward(x, sm, si, ones, zero)
        ctx.slope = float(slope)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, sm, si, ones, zero = ctx.saved_tensors
        N, C = x.shape[0], x.shape[-1]
        Pn = x.numel() // (N * C)
        dy = _f32(dy).contiguous()
        dx = empty(x.shape, x)
        wsb = query('re2e_bn_workspace_bytes', Pn, C)
        ws = workspace(wsb, x.device, 'bn')
        st = Pn * C * 4
        for n in range(N):
            call('re2e_bn_lrelu_bwd', dy.data_ptr() + n * st, x.data_ptr() + n * st, Pn, C, ones.data_ptr(), zero.data_ptr(), sm[n].data_ptr(),
                 si[n].data_ptr(), ctx.slope, dx.data_ptr() + n * st, None, None, 0.0, ws.data_ptr(), wsb)
        return dx, None, None


def instance_norm_lrelu(x, eps=1e-5, slope=0.2):
    return InstanceNormLreluFn.apply(x, eps, slope)


def bn_lrelu(x, gamma, beta, rm, rv, train=True, momentum=0.1, eps=1e-5, slope=0.2):
    """BatchNorm2d + LeakyReLU(slope); slope=1.0 is plain BatchNorm2d."""
    return BnLreluFn.apply(x, gamma, beta, rm, rv, train, momentum, eps, slope)


# ---------------------------------------------------------------------------------------------
# K4 bidirectional LSTM layer (time-major), packed-sequence semantics
# ---------------------------------------------------------------------------------------------
class BiLstmFn(torch.autograd.Function):
    """x (T,B,I) time-major -> y (T,B,2H).  ``w`` = [w_ih_f, w_hh_f, b_ih_f, b_hh_f, w_ih_r, w_hh_r, b_ih_r, b_hh_r]."""

    @staticmethod
    def forward(ctx, x, lens_dev, *w):
        _need_gpu(x)
        x = _f32(x)
        T, B, I = x.shape
        H = w[1].shape[1]
        x2 = x.view(T * B, I)
        # the input projections over the valid (t, b) rows only: the recurrence never uses a pre-activation with t >= len_b (it selects zeros
        # there), so those rows of xg stay unwritten
        maps = row_maps(lens_dev, T, B)
        ctx.maps = maps
        xg = [empty((T * B, 4 * H), x), empty((T * B, 4 * H), x)]
        Ip = (I + 3) & ~3
        if Ip != I and T * B >= 1024 and lib.exp_env('RE2E_NO_PAD_INPUT') != '1':
            # an input width that is not a multiple of 4 (the enhancer's 257 bins) would send three large GEMMs of this layer down
            # the scalar-load path of the engine: work on zero-padded copies of x and W_ih instead (exact: the extra products are 0)
            x2p = zeros((T * B, Ip), x)
            x2p[:, :I].copy_(x2)
            for d in range(2):
                wp = zeros((4 * H, Ip), x)
                wp[:, :I].copy_(w[4 * d].detach())
                if maps is not None:
                    gemm_rows(x2p, wp, xg[d], 4 * H, Ip, maps, bias=w[4 * d + 2], bias2=w[4 * d + 3])
                else:
                    gemm(x2p, wp, xg[d], T * B, 4 * H, Ip, transb=True, bias=w[4 * d + 2], bias2=w[4 * d + 3])
            x2 = x2p
        else:
            for d in range(2):
                if maps is not None and w[4 * d].is_contiguous():
                    gemm_rows(x2, w[4 * d], xg[d], 4 * H, I, maps, bias=w[4 * d + 2], bias2=w[4 * d + 3])
                else:
                    gemm(x2, w[4 * d], xg[d], T * B, 4 * H, I, transb=True, bias=w[4 * d + 2], bias2=w[4 * d + 3])
        # blocks 1..T are written in full by the recurrence (zeros beyond an utterance's length); only the two border blocks -- the
        # state before the first / after the last frame -- have to be cleared (2 x 64 KB instead of 2 x 52 MB for the enhancer)
        ybuf = empty((T + 2, B, 2 * H), x)
        cbuf = empty((T + 2, B, 2 * H), x)
        for buf in (ybuf, cbuf):
            buf[0].zero_()
            buf[T + 1].zero_()
        wsb = query('re2e_lstm_workspace_bytes', B, H)
        ws = workspace(wsb, x.device, 'lstm')
        call('re2e_lstm_seq_fwd', xg[0].data_ptr(), xg[1].data_ptr(), w[1].data_ptr(), w[5].data_ptr(), ybuf.data_ptr(), cbuf.data_ptr(),
             lens_dev.data_ptr(), T, B, H, ws.data_ptr(), wsb)
        ctx.w, ctx.lens = w, lens_dev
        ctx.save_for_backward(x2, xg[0], xg[1], ybuf, cbuf)
        ctx.dims = (T, B, I, H)
        return ybuf[1:T + 1]

    @staticmethod
    def backward(ctx, dy):
        x2, g_f, g_r, ybuf, cbuf = ctx.saved_tensors
        w = ctx.w
        T, B, I, H = ctx.dims
        dy = _f32(dy)
        # the saved gates are overwritten with d(pre-activation gates); a second backward is not supported
        dc = empty((B, 2 * H), dy)
        wsb = query('re2e_lstm_workspace_bytes', B, H)
        ws = workspace(wsb, dy.device, 'lstm')
        # dbias (2, 4H): the column sums of d(gates) = the gradient of b_ih and of b_hh per direction, accumulated beside the recurrence
        # (csrc/lstm.hip lstm_bwd3: no pass of its own over the (T B, 4H) tensors, 105 MB per direction and layer at config 4)
        dbias = empty((2, 4 * H), dy) if any(ctx.needs_input_grad[4 + 4 * d] or ctx.needs_input_grad[5 + 4 * d] for d in range(2)) else None
        if dbias is not None and not FUSED_DBIAS:      # (experiments: the column sums as a pass of their own on the weight-gradient stream, rounds 1-5)
            dbias = None
        call('re2e_lstm_seq_bwd', g_f.data_ptr(), g_r.data_ptr(), w[1].data_ptr(), w[5].data_ptr(), dy.data_ptr(), ybuf.data_ptr(),
             cbuf.data_ptr(), dc.data_ptr(), ctx.lens.data_ptr(), T, B, H, ptr(dbias), ws.data_ptr(), wsb)
        dG = (g_f, g_r)
        M = T * B
        dx = None
        if ctx.needs_input_grad[0]:
            dx = empty((M, I), dy)
            if ctx.maps is not None and I % 4 == 0:
                # d(gates) is zero in the padded rows (the recurrence wrote them): so is dx there -- written as zeros, not computed
                gemm_input_grad_rows(dG[0], w[0], dx, I, 4 * H, ctx.maps, fill=True)
                gemm_input_grad_rows(dG[1], w[4], dx, I, 4 * H, ctx.maps, beta=1.0)
            else:
                gemm_input_grad(dG[0], w[0], dx, M, I, 4 * H)
                gemm_input_grad(dG[1], w[4], dx, M, I, 4 * H, beta=1.0)
            dx = dx.view(T, B, I)
        # A layer whose input needs no gradient is the bottom of its network: its backward recurrence is the LAST link of the
        # chain on this stream (the enhancer's first layer ends the training step's backward), so nothing waits behind its weight
        # gradients here, while on the weight-gradient stream they would queue behind everything deferred so far: they stay on this
        # stream and run beside that backlog: step 72.26 -> 71.71 ms (3 + 3 runs, one GPU session; RE2E_NO_INLINE_LAST_WGRAD=1:
        # deferred like the others).  Handing the weight gradients of the layers ABOVE it to the trainer to run behind the chain as
        # well was measured and changes nothing (71.63 against 71.55; again in round 6, with the filler streams the last to finish: 46.21 against 46.13).
        yflat = ybuf.view((T + 2) * B, 2 * H)
        last = INLINE_LAST_WGRAD and not ctx.needs_input_grad[0]
        needs = ctx.needs_input_grad

        def weight_grads(inline):
            with param_grads(g_f, g_r, x2, ybuf, dbias, ctx.maps, inline=inline):
                for d in range(2):
                    w_ih, w_hh, b_ih, b_hh = w[4 * d:4 * d + 4]
                    n_ih, n_hh, n_bi, n_bh = needs[2 + 4 * d:6 + 4 * d]
                    mp = ctx.maps
                    if n_ih and x2.shape[1] != I:              # padded input copy (see forward): full-width product, then the real columns
                        tmp = empty((4 * H, x2.shape[1]), dy)
                        if mp is not None:
                            gemm_tn_rows(dG[d], x2, tmp, 4 * H, x2.shape[1], mp)
                        else:
                            gemm(dG[d], x2, tmp, 4 * H, x2.shape[1], M, transa=True)
                        with accumulate(w_ih) as (gw, beta):
                            if beta == 0.0:
                                gw.copy_(tmp[:, :I])
                            else:
                                gw.add_(tmp[:, :I])
                    elif n_ih:
                        with accumulate(w_ih) as (gw, beta):
                            if mp is not None:
                                gemm_tn_rows(dG[d], x2, gw, 4 * H, I, mp, beta=beta)       # d(gates) is zero in the padded rows: the valid ones are the sum
                            else:
                                gemm(dG[d], x2, gw, 4 * H, I, M, transa=True, beta=beta)
                    if n_hh:
                        # h_{t-1}: forward direction = ybuf block t (y[t-1]); reverse = ybuf block t+2 (y[t+1])
                        hprev = yflat[(0 if d == 0 else 2 * B):, d * H:]
                        with accumulate(w_hh) as (gw, beta):
                            if mp is not None:
                                gemm_tn_rows(dG[d], hprev.data_ptr(), gw, 4 * H, H, mp, beta=beta, lda=4 * H, ldb=2 * H)
                            else:
                                call_gemm_strided(dG[d], hprev, gw, 4 * H, H, M, lda=4 * H, ldb=2 * H, beta=beta)
                    csum = dbias[d] if dbias is not None else None
                    if csum is None and (n_bi or n_bh):
                        csum = empty((4 * H,), dy)
                        colsum_into(dG[d], M, 4 * H, csum, 0.0)
                    for b_, need in ((b_ih, n_bi), (b_hh, n_bh)):    # both biases get the column sums of d(gates): the recurrence left them in dbias
                        if need:
                            with accumulate(b_) as (gb, beta):
                                if beta == 0.0:
                                    gb.copy_(csum)
                                else:
                                    gb.add_(csum)

        weight_grads(last)
        return (dx, None) + (None,) * len(w)


def call_gemm_strided(A, Bview, C, M, N, K, lda, ldb, beta):
    """TN gemm where B is a strided view (pointer offset + leading dimension)."""
    wsb = query('re2e_gemm_workspace_bytes', 1, 0, M, N, K)
    ws = workspace(wsb, A.device, 'gemm') if wsb else None
    call('re2e_gemm', 1, 0, M, N, K, A.data_ptr(), lda, Bview.data_ptr(), ldb, C.data_ptr(), N, None, None, lib.ACT_NONE, float(beta),
         None, None, None, 0, ptr(ws), wsb)


def bilstm(x_tm, lens_dev, weights):
    return BiLstmFn.apply(x_tm, lens_dev, *weights)


# ---------------------------------------------------------------------------------------------
# K6 CTC
# ---------------------------------------------------------------------------------------------
class CtcFn(torch.autograd.Function):
    """logits (T,B,V) time-major raw activations -> loss (sum_b nll_b / B), shape (1,)."""

    @staticmethod
    def forward(ctx, logits, hlens_dev, labels_dev, loff_dev, llen_dev, Lmax):
        _need_gpu(logits)
        logits = _f32(logits)
        T, B, V = logits.shape
        wsb = query('re2e_ctc_workspace_bytes', T, B, Lmax)
        ws = torch.empty(wsb // 4 + 4, dtype=torch.float32, device=logits.device)   # kept for the backward
        loss = empty((1,), logits)
        nll = empty((B,), logits)
        call('re2e_ctc_fwd', logits.data_ptr(), T, B, V, hlens_dev.data_ptr(), labels_dev.data_ptr(), loff_dev.data_ptr(), llen_dev.data_ptr(),
             Lmax, loss.data_ptr(), nll.data_ptr(), ws.data_ptr(), wsb)
        ctx.save_for_backward(logits, ws, nll)
        ctx.meta = (hlens_dev, labels_dev, loff_dev, llen_dev, Lmax)
        return loss

    @staticmethod
    def backward(ctx, g):
        logits, ws, nll = ctx.saved_tensors
        hlens_dev, labels_dev, loff_dev, llen_dev, Lmax = ctx.meta
        T, B, V = logits.shape
        g = _f32(g).reshape(1)
        # An odd vocabulary (V = 4233): the gradient is written with rows padded to a multiple of 16 (zeros) and handed on as a VIEW of
        # its first V columns; LinearFn.backward recognises the buffer (_PADDED_GRADS) and runs the projection's input- and
        # weight-gradient products over the padded width on the engine's 16-byte-load path (54 -> ~110 TFLOP/s for the input gradient)
        Vp = (V + 15) // 16 * 16 if (V % 4 and T * B >= 2048) else V
        d = empty((T, B, Vp), logits)
        call('re2e_ctc_bwd', logits.data_ptr(), T, B, V, hlens_dev.data_ptr(), labels_dev.data_ptr(), loff_dev.data_ptr(), llen_dev.data_ptr(),
             Lmax, nll.data_ptr(), g.data_ptr(), d.data_ptr(), Vp, ws.data_ptr())
        if Vp != V:
            if len(_PADDED_GRADS) > 8:                       # entries nobody claimed (the logits were a leaf)
                _PADDED_GRADS.clear()
            _PADDED_GRADS[d.data_ptr()] = (T * B, V, Vp)
            d = d[..., :V]
        return d, None, None, None, None, None


ctc_loss = CtcFn.apply


# ---------------------------------------------------------------------------------------------
# row gather (valid frames) and Deep-CORAL
# ---------------------------------------------------------------------------------------------
class GatherRowsFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, src2d, idx_dev):
        src2d = _f32(src2d)
        n, W = idx_dev.numel(), src2d.shape[1]
        out = empty((n, W), src2d)
        call('re2e_gather_rows', src2d.data_ptr(), idx_dev.data_ptr(), out.data_ptr(), n, W)
        ctx.idx, ctx.shape = idx_dev, src2d.shape
        return out

    @staticmethod
    def backward(ctx, dy):
        dy = _f32(dy)
        dx = zeros(ctx.shape, dy)
        call('re2e_scatter_rows', dy.data_ptr(), ctx.idx.data_ptr(), dx.data_ptr(), ctx.idx.numel(), ctx.shape[1])
        return dx, None


gather_rows = GatherRowsFn.apply


class CoralFn(torch.autograd.Function):
    """Deep-CORAL ||Cov(src)-Cov(tgt)||_F^2 / (4 d^2) (S2, build-defined)."""

    @staticmethod
    def forward(ctx, src, tgt):
        _need_gpu(src)
        src, tgt = _f32(src), _f32(tgt)
        d = src.shape[1]
        covs, mus = [], []
        for x in (src, tgt):
            n = x.shape[0]
            mu = empty((d,), x)
            colsum_into(x, n, d, mu, 0.0)
            call('re2e_axpby', 1.0 / n, mu.data_ptr(), 0.0, mu.data_ptr(), d)
            G = empty((d, d), x)
            call_gemm_strided(x, x, G, d, d, n, lda=d, ldb=d, beta=0.0)            # X^T X
            M2 = empty((d, d), x)
            gemm(mu, mu, M2, d, d, 1, lda=1, ldb=d)                                # mu mu^T
            call('re2e_axpby', -float(n) / (n - 1), M2.data_ptr(), 1.0 / (n - 1), G.data_ptr(), d * d)   # cov
            covs.append(G)
            mus.append(mu)
        D = covs[0]
        call('re2e_axpby', -1.0, covs[1].data_ptr(), 1.0, D.data_ptr(), d * d)      # D = Cs - Ct
        out = empty((1,), src)
        wsb = query('re2e_reduce_workspace_bytes', d * d)
        ws = workspace(wsb, src.device, 'reduce')
        call('re2e_sumsq', D.data_ptr(), d * d, out.data_ptr(), ws.data_ptr(), wsb)
        call('re2e_axpby', 1.0 / (4.0 * d * d), out.data_ptr(), 0.0, out.data_ptr(), 1)
        ctx.save_for_backward(src, tgt, D, mus[0], mus[1])
        return out.view(())

    @staticmethod
    def backward(ctx, g):
        src, tgt, D, mu_s, mu_t = ctx.saved_tensors
        d = src.shape[1]
        g = _f32(g).reshape(1)
        outs = []
        for x, mu, sign, need in ((src, mu_s, 1.0, ctx.needs_input_grad[0]), (tgt, mu_t, -1.0, ctx.needs_input_grad[1])):
            if not need:
                outs.append(None)
                continue
            n = x.shape[0]
            # dX = g * sign * (4/(n-1)) / (4 d^2) * (X - 1 mu^T) D
            Ds = empty((d, d), x)
            call('re2e_loss_bwd', D.data_ptr(), None, 0.0, d * d, lib.LOSS_L2, g.data_ptr(), sign * 0.5 * d * d * (4.0 / (n - 1)) / (4.0 * d * d),
                 Ds.data_ptr(), 0.0)   # Ds = g * coef * D  (L2 grad = 2*D/n_elems; scale cancels it)
            v = empty((1, d), x)
            gemm(mu, Ds, v, 1, d, d)                                               # mu^T Ds
            call('re2e_axpby', -1.0, v.data_ptr(), 0.0, v.data_ptr(), d)
            dx = empty((n, d), x)
            gemm(x, Ds, dx, n, d, d, bias=v.view(d))
            outs.append(dx)
        return tuple(outs)


coral = CoralFn.apply


# ---------------------------------------------------------------------------------------------
# K8 cross-entropy (ignore -1) + accuracy counters
# ---------------------------------------------------------------------------------------------
class CeFn(torch.autograd.Function):
    """logits (R,V), targets int32 (R) with -1 = ignore -> (loss*scale (1,), stats (3,) = [loss, #correct, #valid])."""

    @staticmethod
    def forward(ctx, logits, targets_dev, scale):
        _need_gpu(logits)
        logits = _f32(logits)
        R, V = logits.shape
        out = empty((3,), logits)
        lse = empty((R,), logits)
        ws = workspace(3 * R * 4, logits.device, 'ce')
        call('re2e_ce_fwd', logits.data_ptr(), targets_dev.data_ptr(), R, V, float(scale), out.data_ptr(), lse.data_ptr(), ws.data_ptr(), 3 * R * 4)
        ctx.save_for_backward(logits, lse, out)
        ctx.targets, ctx.scale = targets_dev, float(scale)
        ctx.mark_non_differentiable(out)
        return out[0:1].clone(), out

    @staticmethod
    def backward(ctx, g, _unused):
        logits, lse, out = ctx.saved_tensors
        R, V = logits.shape
        g = _f32(g).reshape(1)
        d = empty((R, V), logits)
        call('re2e_ce_bwd', logits.data_ptr(), ctx.targets.data_ptr(), lse.data_ptr(), out.data_ptr(), R, V, ctx.scale, g.data_ptr(), d.data_ptr())
        return d, None, None


def cross_entropy(logits, targets_dev, scale=1.0):
    return CeFn.apply(logits, targets_dev, scale)


class LabelSmoothFn(torch.autograd.Function):
    """-(1/nutt) * sum(log_softmax(logits) * dist) over all rows  (label-smoothing term, e2e_decoder.py:162-166)."""

    @staticmethod
    def forward(ctx, logits, dist, nutt):
        _need_gpu(logits)
        logits = _f32(logits)
        R, V = logits.shape
        out = empty((1,), logits)
        ws = workspace(R * 4, logits.device, 'lsm')
        call('re2e_lsm_fwd', logits.data_ptr(), dist.data_ptr(), R, V, int(nutt), out.data_ptr(), ws.data_ptr(), R * 4)
        ctx.save_for_backward(logits, dist)
        ctx.nutt = int(nutt)
        return out

    @staticmethod
    def backward(ctx, g):
        logits, dist = ctx.saved_tensors
        R, V = logits.shape
        g = _f32(g).reshape(1)
        d = empty((R, V), logits)
        call('re2e_lsm_bwd', logits.data_ptr(), dist.data_ptr(), R, V, ctx.nutt, g.data_ptr(), d.data_ptr())
        return d, None, None


label_smoothing = LabelSmoothFn.apply


# ---------------------------------------------------------------------------------------------
# K7 + K8 decoder loop: AttLoc step -> LSTMCell, teacher forced  (e2e_decoder.py:121-152)
# ---------------------------------------------------------------------------------------------
class CmvnPairFn(torch.autograd.Function):
    """(a, b) (B,T,N) each -> (2B,T,N) = [(a+c0)*c1 ; (b+c0)*c1]  (S1: both ASR branches, one batch)."""

    @staticmethod
    def forward(ctx, a, b, cmvn):
        _need_gpu(a)
        a = _f32(a)
        B, T, N = a.shape
        rows = B * T
        out = empty(((2 if b is not None else 1) * B, T, N), a)
        call('re2e_affine_cols', a.data_ptr(), cmvn[0].data_ptr(), cmvn[1].data_ptr(), out.data_ptr(), rows, N)
        if b is not None:
            b = _f32(b)
            call('re2e_affine_cols', b.data_ptr(), cmvn[0].data_ptr(), cmvn[1].data_ptr(), out.data_ptr() + 4 * rows * N, rows, N)
        ctx.cmvn, ctx.dims, ctx.two = cmvn, (B, T, N), b is not None
        return out

    @staticmethod
    def backward(ctx, dy):
        B, T, N = ctx.dims
        rows = B * T
        dy = _f32(dy)
        outs = []
        for k, need in enumerate(ctx.needs_input_grad[:2]):
            if not need or (k == 1 and not ctx.two):
                outs.append(None)
                continue
            d = empty((B, T, N), dy)
            call('re2e_affine_cols', dy.data_ptr() + 4 * k * rows * N, None, ctx.cmvn[1].data_ptr(), d.data_ptr(), rows, N)
            outs.append(d)
        return outs[0], outs[1], None


def cmvn_pair(a, b, cmvn):
    return CmvnPairFn.apply(a, b, cmvn)


class MulConstFn(torch.autograd.Function):
    """a * b where b carries no gradient (clean * cos_angle)."""

    @staticmethod
    def forward(ctx, a, b):
        a, b = _f32(a), _f32(b)
        out = empty(a.shape, a)
        call('re2e_mul', a.data_ptr(), b.data_ptr(), out.data_ptr(), a.numel())
        ctx.save_for_backward(b)
        return out

    @staticmethod
    def backward(ctx, dy):
        (b,) = ctx.saved_tensors
        dy = _f32(dy)
        d = empty(dy.shape, dy)
        call('re2e_mul', dy.data_ptr(), b.data_ptr(), d.data_ptr(), dy.numel())
        return d, None


mul_const = MulConstFn.apply


# ---------------------------------------------------------------------------------------------
# Dropout (F.dropout e2e_ctc.py:51, nn.LSTM(dropout=) e2e_encoder.py:156-157, nn.Dropout enhance_model.py:298)
# ---------------------------------------------------------------------------------------------
_DROPOUT = {'seed': 0x5EED5EED5EED, 'call': 0}


def dropout_seed(seed, call=0):
    """Seed of the counter-based dropout masks (re2e_dropout) and the index of the next mask: every ``dropout`` call draws
    mask number ``call`` of the stream ``seed`` and advances it, so a run is reproducible from (seed, call)."""
    _DROPOUT['seed'], _DROPOUT['call'] = int(seed) & 0xFFFFFFFFFFFFFFFF, int(call)


def dropout_state():
    return _DROPOUT['seed'], _DROPOUT['call']


class DropoutFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, p, seed, call_idx):
        _need_gpu(x)
        x = _f32(x)
        y = empty(x.shape, x)
        call('re2e_dropout', x.data_ptr(), y.data_ptr(), x.numel(), float(p), seed, call_idx)
        ctx.cfg = (float(p), seed, call_idx)
        return y

    @staticmethod
    def backward(ctx, dy):
        p, seed, call_idx = ctx.cfg
        dy = _f32(dy)
        dx = empty(dy.shape, dy)
        call('re2e_dropout', dy.data_ptr(), dx.data_ptr(), dy.numel(), p, seed, call_idx)      # same mask, regenerated
        return dx, None, None, None


def dropout(x, p):
    """x * mask / (1 - p) with the next mask of the module-level stream (``dropout_seed``); p == 0 returns x."""
    if not p:
        return x
    seed, idx = _DROPOUT['seed'], _DROPOUT['call']
    _DROPOUT['call'] = (idx + 1) & 0xFFFFFFFF
    return DropoutFn.apply(x, p, seed, idx)


DECODER_FUSED = lib.exp_env('RE2E_NO_DECODER_FUSION', '0') != '1'     # fused LSTMCell step kernels (A/B switch)
_DECLOOP_BWS_READ = {}     # workspace address -> event behind the last re2e_dec_loop_dwconv that read it (DecoderLoopFn.backward)
DECODER_PERSIST = os.environ.get('RE2E_DEC_PERSIST', '1') != '0'     # the teacher-forced loop as one persistent launch (csrc/decloop.hip)


class DecoderLoopFn(torch.autograd.Function):
    """hmask (B,T,E) masked encoder states, pre (B,T,A) = mlp_enc(hmask) -> decoder states (L1,B,D) step-major.

    ``P`` is a dict of the decoder / attention Parameters (reference names):
      embed, w_ih, w_hh, b_ih, b_hh, mlp_dec, mlp_att, loc_conv, gvec_w, gvec_b
    Teacher forcing makes the embedding half of the LSTMCell input projection independent of the
    recurrence, so it is ONE batched GEMM over all (L+1)*B tokens before the loop; inside the loop only
    the context (K=eprojs) and recurrent (K=dunits) halves remain (skinny split-K GEMMs).

    ``sample_steps`` (optional tuple of L1 bools): step i feeds back the arg-max of step i-1's output layer instead
    of the reference label (scheduled sampling e2e_decoder.py:123-127, and the greedy pass of
    calculate_all_attentions :408-412); needs ``out_w`` / ``out_b`` in ``P``.  The fed-back token is an integer, so
    no gradient flows through the arg-max -- only into the embedding row that was used."""

    @staticmethod
    def forward(ctx, hmask, pre, ids_tm, hlens_dev, L1, Pm, sample_steps=None):
        _need_gpu(hmask)
        hmask, pre = _f32(hmask), _f32(pre)
        dev = hmask.device
        B, T, E = hmask.shape
        A = pre.shape[2]
        Dd = Pm['embed'].shape[1]
        D = Pm['w_hh'].shape[1]
        C, Kf = Pm['loc_conv'].shape[0], Pm['loc_conv'].shape[3]
        Fh = (Kf - 1) // 2
        M = L1 * B
        w_ih = Pm['w_ih']
        ldw = Dd + E
        emb = empty((L1, B, Dd), hmask)
        call('re2e_embedding_fwd', Pm['embed'].data_ptr(), ids_tm.data_ptr(), M, Dd, emb.data_ptr(), Dd)
        gates = empty((L1, B, 4 * D), hmask)
        gemm(emb, w_ih, gates, M, 4 * D, Dd, transb=True, ldb=ldw, bias=Pm['b_ih'], bias2=Pm['b_hh'])
        cx = empty((L1, B, E), hmask)
        z = zeros((L1 + 1, B, D), hmask)
        c = zeros((L1 + 1, B, D), hmask)
        w = empty((L1, B, T), hmask)
        w_ctx = w_ih.data_ptr() + 4 * Dd            # W_ih[:, Dd:]  (4D, E) view with leading dimension Dd+E
        w_decT = empty((D, A), hmask)               # mlp_dec.weight transposed once: coalesced reads in the step kernel
        call('re2e_transpose01', Pm['mlp_dec'].data_ptr(), w_decT.data_ptr(), A, D, 1)
        conv = empty((L1, B, T, C), hmask)          # saved for the backward (no recomputation of the location conv)
        dpj = empty((L1, B, A), hmask)
        e_scr = empty((B, T), hmask)
        fused = DECODER_FUSED and E % 4 == 0 and D % 4 == 0 and ldw % 4 == 0
        sampled = sample_steps is not None and any(sample_steps[1:])
        if sampled:
            ids_tm = ids_tm.clone()                 # becomes the list of tokens actually fed
            V = Pm['out_w'].shape[0]
            logits = empty((B, V), hmask)
        # the whole loop as ONE persistent launch (csrc/decloop.hip) when no step's token depends on the previous step's output
        # and the shape is inside the resident form's limits (0 bytes = not); the launch-per-step sequence below otherwise
        lwsb = 0 if (sampled or not fused or not DECODER_PERSIST) else query('re2e_dec_loop_workspace_bytes', L1, B, T, E, D, A, C, Fh)
        if lwsb:
            lws = workspace(lwsb, dev, 'decloop')
            call('re2e_dec_loop_fwd', pre.data_ptr(), hmask.data_ptr(), hlens_dev.data_ptr(), w_decT.data_ptr(), Pm['mlp_att'].data_ptr(),
                 Pm['loc_conv'].data_ptr(), Pm['gvec_w'].data_ptr(), Pm['gvec_b'].data_ptr(), w_ctx, ldw, Pm['w_hh'].data_ptr(), gates.data_ptr(),
                 z.data_ptr(), c.data_ptr(), w.data_ptr(), cx.data_ptr(), conv.data_ptr(), dpj.data_ptr(), L1, B, T, E, D, A, C, Fh,
                 lws.data_ptr(), lwsb)
        for i in range(0 if not lwsb else L1, L1):
            if sampled and i > 0 and sample_steps[i]:
                gemm(z[i], Pm['out_w'], logits, B, V, D, transb=True, bias=Pm['out_b'])          # y_{i-1} = output(z_{i-1})
                ids_i = ids_tm.data_ptr() + 4 * i * B
                call('re2e_argmax_rows', logits.data_ptr(), B, V, V, ids_i)
                call('re2e_embedding_fwd', Pm['embed'].data_ptr(), ids_i, B, Dd, emb[i].data_ptr(), Dd)
                gemm(emb[i], w_ih, gates[i], B, 4 * D, Dd, transb=True, ldb=ldw, bias=Pm['b_ih'], bias2=Pm['b_hh'])
            call('re2e_attloc_fwd', pre.data_ptr(), hmask.data_ptr(), z[i].data_ptr(), w[i - 1].data_ptr() if i > 0 else None,
                 hlens_dev.data_ptr(), w_decT.data_ptr(), Pm['mlp_att'].data_ptr(), Pm['loc_conv'].data_ptr(),
                 Pm['gvec_w'].data_ptr(), Pm['gvec_b'].data_ptr(), B, T, E, D, A, C, Fh, w[i].data_ptr(), cx[i].data_ptr(), E,
                 conv[i].data_ptr(), dpj[i].data_ptr(), e_scr.data_ptr())
            if fused:
                # gates += [ctx | z] [W_ih[:, Dd:] | W_hh]^T and the cell in ONE launch (two skinny GEMMs + the cell kernel before)
                call('re2e_dec_gates_cell_fwd', cx[i].data_ptr(), z[i].data_ptr(), w_ctx, ldw, Pm['w_hh'].data_ptr(), gates[i].data_ptr(),
                     c[i].data_ptr(), c[i + 1].data_ptr(), z[i + 1].data_ptr(), B, E, D)
            else:
                gemm(cx[i], w_ctx, gates[i], B, 4 * D, E, transb=True, ldb=ldw, beta=1.0, dev=dev)
                gemm(z[i], Pm['w_hh'], gates[i], B, 4 * D, D, transb=True, beta=1.0)
                call('re2e_lstm_cell_fwd', gates[i].data_ptr(), c[i].data_ptr(), c[i + 1].data_ptr(), z[i + 1].data_ptr(), B, D)
        ctx.Pm, ctx.ids, ctx.hlens = Pm, ids_tm, hlens_dev
        ctx.persist = bool(lwsb)
        ctx.dims = (B, T, E, A, Dd, D, C, Fh, L1)
        ctx.save_for_backward(hmask, pre, emb, cx, z, c, w, gates, conv, dpj)
        ctx.mark_non_differentiable(w)
        return z[1:], w

    @staticmethod
    def backward(ctx, dZ, _dw_unused):
        hmask, pre, emb, cx, z, c, w, gates, conv, dpj = ctx.saved_tensors
        Pm = ctx.Pm
        B, T, E, A, Dd, D, C, Fh, L1 = ctx.dims
        dev = hmask.device
        dZ = _f32(dZ)
        w_ih = Pm['w_ih']
        ldw = Dd + E
        w_ctx = w_ih.data_ptr() + 4 * Dd
        d_pre = empty((B, T, A), hmask)
        de_all = empty((L1, B, T), hmask)
        d_enc = empty((B, T, E), hmask)
        npart = query('re2e_attloc_partial_floats', A, C, Fh)
        partials = zeros((B, npart), hmask)
        ddp = empty((L1, B, A), hmask)
        d_cx_all = empty((L1, B, E), hmask)
        awsb = query('re2e_attloc_workspace_bytes', B, T, A, C)
        aws = workspace(awsb, dev, 'attloc')
        dz_carry = zeros((B, D), hmask)
        dc_a, dc_b = zeros((B, D), hmask), empty((B, D), hmask)
        dw_a, dw_b = empty((B, T), hmask), empty((B, T), hmask)
        have_dw = False
        fused = DECODER_FUSED and B <= 32
        # the whole reverse loop as ONE persistent launch (csrc/decloop.hip) when the forward took that form too (no sampled tokens)
        bwsb = query('re2e_dec_loop_bwd_workspace_bytes', L1, B, T, E, D, A, C, Fh) if (ctx.persist and DECODER_PERSIST and fused) else 0
        if bwsb:
            bws = workspace(bwsb, dev, 'decloop_bwd')
            # the grow-only workspace is read once more by re2e_dec_loop_dwconv on the weight-gradient stream (below): a second backward on this
            # stream before the trainer has joined that stream (two decoder passes in one graph, gradient accumulation) must not memset / rewrite
            # the d conv rows under it (record_stream does nothing for a buffer that is never freed)
            ev_prev = _DECLOOP_BWS_READ.pop(bws.data_ptr(), None)
            if ev_prev is not None:
                torch.cuda.current_stream().wait_event(ev_prev)
            call('re2e_dec_loop_bwd', pre.data_ptr(), hmask.data_ptr(), cx.data_ptr(), z.data_ptr(), c.data_ptr(), w.data_ptr(), conv.data_ptr(), dpj.data_ptr(),
                 dZ.data_ptr(), ctx.hlens.data_ptr(), w_ctx, ldw, Pm['w_hh'].data_ptr(), Pm['mlp_dec'].data_ptr(), Pm['mlp_att'].data_ptr(),
                 Pm['loc_conv'].data_ptr(), Pm['gvec_w'].data_ptr(), gates.data_ptr(), d_cx_all.data_ptr(), de_all.data_ptr(), ddp.data_ptr(),
                 d_pre.data_ptr(), L1, B, T, E, D, A, C, Fh, bws.data_ptr(), bwsb)
        for i in range(L1 - 1 if not bwsb else -1, -1, -1):
            d_cx = d_cx_all[i]
            if fused:
                # dh = carried dz + dZ[i] inside the cell kernel; d ctx and d z_{i-1} from the same dgates in one launch
                call('re2e_lstm_cell_bwd', gates[i].data_ptr(), c[i].data_ptr(), c[i + 1].data_ptr(), dz_carry.data_ptr(), dZ[i].data_ptr(),
                     dc_a.data_ptr(), dc_b.data_ptr(), B, D)
                dc_a, dc_b = dc_b, dc_a
                call('re2e_gemm_skinny2', B, 4 * D, gates[i].data_ptr(), 4 * D, w_ctx, ldw, E, d_cx.data_ptr(), E, Pm['w_hh'].data_ptr(), D, D,
                     dz_carry.data_ptr(), D)
            else:
                call('re2e_axpby', 1.0, dZ[i].data_ptr(), 1.0, dz_carry.data_ptr(), B * D)
                call('re2e_lstm_cell_bwd', gates[i].data_ptr(), c[i].data_ptr(), c[i + 1].data_ptr(), dz_carry.data_ptr(), None, dc_a.data_ptr(),
                     dc_b.data_ptr(), B, D)
                dc_a, dc_b = dc_b, dc_a
                gemm(gates[i], w_ctx, d_cx, B, E, 4 * D, ldb=ldw, dev=dev)                     # d ctx = dgates W_ih[:, Dd:]
                gemm(gates[i], Pm['w_hh'], dz_carry, B, D, 4 * D)                                # d z_{i-1} (recurrent path)
            call('re2e_attloc_bwd', pre.data_ptr(), hmask.data_ptr(), w[i - 1].data_ptr() if i > 0 else None, w[i].data_ptr(),
                 ctx.hlens.data_ptr(), Pm['mlp_att'].data_ptr(), Pm['loc_conv'].data_ptr(), Pm['gvec_w'].data_ptr(), conv[i].data_ptr(),
                 dpj[i].data_ptr(), cx[i].data_ptr(), d_cx.data_ptr(), E, dw_a.data_ptr() if have_dw else None, B, T, E, A, C, Fh,
                 de_all[i].data_ptr(), dw_b.data_ptr() if i > 0 else None, ddp[i].data_ptr(), partials.data_ptr(), aws.data_ptr(), awsb)
            dw_a, dw_b = dw_b, dw_a
            have_dw = True
            gemm(ddp[i], Pm['mlp_dec'], dz_carry, B, D, A, beta=1.0)
        call('re2e_attloc_denc', w.data_ptr(), d_cx_all.data_ptr(), L1, B, T, E, d_enc.data_ptr(), 0.0)
        def dpre_into(d_pre_out):
            aw = workspace(awsb, dev, 'attloc')
            call('re2e_attloc_dpre', pre.data_ptr(), conv.data_ptr(), dpj.data_ptr(), de_all.data_ptr(), Pm['mlp_att'].data_ptr(), Pm['gvec_w'].data_ptr(),
                 L1, B, T, A, C, Fh, d_pre_out.data_ptr(), partials.data_ptr(), aw.data_ptr(), awsb)
        if not bwsb:
            dpre_into(d_pre)
        M = L1 * B
        G2, zp2 = gates.view(M, 4 * D), z[:L1].reshape(M, D)
        # nothing downstream waits for the decoder's weight gradients (~0.4 ms of small GEMMs and reductions): weight-gradient stream
        if w_ih.requires_grad:
            with param_grads(gates, z, emb, cx, ddp, partials, *((bws, de_all, conv, dpj, pre, w) if bwsb else ())):
                if bwsb:
                    # the persistent loop has written d_pre itself: the weight-gradient sums of the energy backward (d gvec, d W_att, recomputed
                    # from the saved rows) and d W_conv (from the d conv rows it left in its workspace) leave the critical stream
                    dpre_into(empty((B, T, A), hmask))
                    call('re2e_dec_loop_dwconv', w.data_ptr(), ctx.hlens.data_ptr(), bws.data_ptr(), bwsb, partials.data_ptr(), npart, A + 1 + A * C,
                         L1, B, T, E, D, A, C, Fh)
                    ev_read = torch.cuda.Event()
                    ev_read.record()                         # (on the stream the call above ran on)
                    _DECLOOP_BWS_READ[bws.data_ptr()] = ev_read
                with accumulate(w_ih) as (gw, beta):
                    gemm(G2, emb.view(M, Dd), gw, 4 * D, Dd, M, transa=True, ldc=ldw, beta=beta)                       # dW_ih[:, :Dd]
                    gemm(G2, cx.view(M, E), gw.data_ptr() + 4 * Dd, 4 * D, E, M, transa=True, ldc=ldw, beta=beta)      # dW_ih[:, Dd:]
                with accumulate(Pm['w_hh']) as (gw, beta):
                    gemm(G2, zp2, gw, 4 * D, D, M, transa=True, beta=beta)
                for k in ('b_ih', 'b_hh'):
                    with accumulate(Pm[k]) as (gb, beta):
                        colsum_into(G2, M, 4 * D, gb, beta)
                with accumulate(Pm['mlp_dec']) as (gw, beta):
                    gemm(ddp.view(M, A), zp2, gw, A, D, M, transa=True, beta=beta)
                d_emb = empty((M, Dd), hmask)
                gemm(G2, w_ih, d_emb, M, Dd, 4 * D, ldb=ldw)                                                        # dgates W_ih[:, :Dd]
                V = Pm['embed'].shape[0]
                with accumulate(Pm['embed']) as (gw, beta):
                    call('re2e_embedding_bwd', d_emb.data_ptr(), Dd, ctx.ids.data_ptr(), M, Dd, V, gw.data_ptr(), beta)
                tot = empty((npart,), hmask)
                colsum_into(partials, B, npart, tot, 0.0)
                off = 0
                for k, n in (('gvec_w', A), ('gvec_b', 1), ('mlp_att', A * C), ('loc_conv', C * (2 * Fh + 1))):
                    with accumulate(Pm[k]) as (gt, beta):
                        call('re2e_axpby', 1.0, tot.data_ptr() + 4 * off, beta, gt.data_ptr(), n)
                    off += n
        return d_enc, d_pre, None, None, None, None, None


decoder_loop = DecoderLoopFn.apply


class MaskRowsFn(torch.autograd.Function):
    """mask_by_length(x, lens, 0)  (e2e_common.py:190-195)."""

    @staticmethod
    def forward(ctx, x, lens_dev):
        x = _f32(x)
        B, T, W = x.shape
        y = empty(x.shape, x)
        call('re2e_mask_rows', x.data_ptr(), y.data_ptr(), lens_dev.data_ptr(), B, T, W)
        ctx.lens = lens_dev
        return y

    @staticmethod
    def backward(ctx, dy):
        dy = _f32(dy)
        B, T, W = dy.shape
        dx = empty(dy.shape, dy)
        call('re2e_mask_rows', dy.data_ptr(), dx.data_ptr(), ctx.lens.data_ptr(), B, T, W)
        return dx, None


mask_rows = MaskRowsFn.apply
