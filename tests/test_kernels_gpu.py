"""Parity of every HIP kernel family (through the C ABI + autograd wrappers) against plain
PyTorch fp32 ops on the CPU.  GPU only.  Tolerance: 1e-3 relative to the tensor scale
(north_star: "within 1e-3 fp32"); most kernels land at 1e-5..1e-6."""
import math
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

DEV = 'cuda:0'
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _ops():
    from robust_e2e_gan_amd import ops, lib
    assert lib.query('re2e_device_ok') == 1, 'not a gfx950 device'
    return ops, lib


def close(name, got, ref, tol=1e-3, atol=1e-6):
    got = got.detach().float().cpu()
    ref = ref.detach().float().cpu()
    assert got.shape == ref.shape, (name, got.shape, ref.shape)
    err = (got - ref).abs().max().item()
    scale = ref.abs().max().item()
    assert math.isfinite(err), name
    assert err <= tol * scale + atol, '%s: max err %.3e vs scale %.3e' % (name, err, scale)
    return err


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed + sum(shape))
    return torch.randn(*shape, generator=g) * scale


# ---------------------------------------------------------------------------------------------
# the last two: 800 / 801 x 16 tiles of 256x128 on 256 workgroup slots -> the rows of the mostly empty 4th round run as 64x64 tiles
@pytest.mark.parametrize('M,N,K', [(77, 257, 130), (256, 128, 64), (5, 3, 7), (300, 4233, 512), (130, 64, 257), (12800, 2048, 48),
                                   (12803, 2040, 36)])
def test_gemm_nt_nn(M, N, K):
    ops, lib = _ops()
    A, Bm, bias, C0 = rnd(M, K), rnd(N, K, seed=1), rnd(N, seed=2), rnd(M, N, seed=3)
    a, b, bi = A.to(DEV), Bm.to(DEV), bias.to(DEV)
    c = torch.empty(M, N, device=DEV)
    ops.gemm(a, b, c, M, N, K, transb=True, bias=bi, act=lib.ACT_TANH)
    close('nt+bias+tanh', c, torch.tanh(A @ Bm.t() + bias))
    c = C0.to(DEV).clone()
    ops.gemm(a, b, c, M, N, K, transb=True, beta=1.0)
    close('nt beta', c, A @ Bm.t() + C0)
    bt = Bm.t().contiguous().to(DEV)        # stored [K,N]
    c = torch.empty(M, N, device=DEV)
    ops.gemm(a, bt, c, M, N, K, act=lib.ACT_RELU)
    close('nn relu', c, torch.relu(A @ Bm.t()))


# x W^T on csrc/gemm_nt.hip (LDS-DMA pipeline + stream-K tail).  The shapes pick, on a 256-CU chip: whole rounds only (768 tiles of 256x128);
# a tail behind whole tiles (800 tiles of 128x64: 32 tiles cut into 256 unit ranges; 12800 x 512: 144 tiles behind 256); a product that is ALL tail
# (100 tiles) with a long K; K tails (260 = 16 k-tiles + 4, 36 = 2 + 4, 20); ragged M and N edges; 4-column N; one k-tile.
@pytest.mark.parametrize('M,N,K', [(12288, 2048, 64), (12800, 512, 260), (6400, 512, 4240), (12800, 1024, 512), (2049, 260, 36), (7777, 1028, 1000),
                                   (300, 516, 200), (4100, 4, 20), (256, 64, 16), (25600, 256, 260)])
def test_gemm_nt_pipeline_and_stream_k(M, N, K):
    ops, lib = _ops()
    assert lib.query('re2e_gemm_workspace_bytes', 0, 1, M, N, K) >= 0
    A, Bm, b1, b2, C0 = rnd(M, K + 4), rnd(N, K + 8, seed=1), rnd(N, seed=2), rnd(N, seed=3), rnd(M, N + 12, seed=4)
    a, b = A.to(DEV), Bm.to(DEV)
    want = A[:, :K].double() @ Bm[:, :K].double().t()                      # leading dimensions larger than the widths, fp64 truth
    c = C0.to(DEV).clone()
    ops.gemm(a, b, c, M, N, K, transb=True, lda=K + 4, ldb=K + 8, ldc=N + 12, bias=b1.to(DEV), bias2=b2.to(DEV), act=lib.ACT_TANH)
    # tanh's slope is <= 1: the output may be off by what the fp32 sum is allowed below (2e-5 of the largest pre-activation), not by 2e-5 of tanh's own range
    close('nt2 tanh', c[:, :N], torch.tanh(want + b1.double() + b2.double()).float(), tol=0.0, atol=2e-5 * want.abs().max().item())
    assert torch.equal(c[:, N:].cpu(), C0[:, N:]), 'wrote beyond the N columns'
    c2 = C0.to(DEV).clone()
    ops.gemm(a, b, c2, M, N, K, transb=True, lda=K + 4, ldb=K + 8, ldc=N + 12, bias=b1.to(DEV), bias2=b2.to(DEV), act=lib.ACT_TANH)
    assert torch.equal(c, c2), 'the stream-K finisher is whoever arrives last: the sum must not depend on it'
    for act, fn in ((lib.ACT_NONE, lambda v: v), (lib.ACT_RELU, torch.relu), (lib.ACT_LRELU, lambda v: F.leaky_relu(v, 0.2)), (lib.ACT_SIGMOID, torch.sigmoid)):
        c = C0.to(DEV).clone()
        ops.gemm(a, b, c, M, N, K, transb=True, lda=K + 4, ldb=K + 8, ldc=N + 12, act=act, beta=1.0)
        close('nt2 act %d beta' % act, c[:, :N], (fn(want) + C0[:, :N].double()).float(), tol=2e-5)
    # the tiles a FILLER stream gets (4-wave only) give the same numbers up to the summation order of a cut tile
    st = torch.cuda.Stream()
    lib.set_stream_role(st, True)
    try:
        st.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(st):
            c = torch.empty(M, N, device=DEV)
            ops.gemm(a, b, c, M, N, K, transb=True, lda=K + 4, ldb=K + 8)
        st.synchronize()
    finally:
        lib.set_stream_role(st, False)
    close('nt2 filler stream', c, want.float(), tol=2e-5)


@pytest.mark.parametrize('M,N,K', [(512, 1024, 12800), (300, 1312, 4232), (2048, 512, 12800), (6400, 512, 4240), (1024, 256, 25600)])
def test_gemm_nt_stream_k_tail_under_load(M, N, K):
    """The stream-K tail of csrc/gemm_nt.hip on products whose tiles are cut into MANY parts (few tiles, long K: the whole product is tail): the
    finisher of a tile is whoever arrives last, and beside a second stream that keeps the chip unevenly busy that changes from launch to launch --
    every launch must give the fp64 product and the very same bits.  Also through a row map (re2e_gemm_nt_rows: gathered / scattered rows)."""
    ops, lib = _ops()
    A, Bm = rnd(M, K).to(DEV), rnd(N, K, seed=1).to(DEV)
    want = (A.double() @ Bm.double().t())
    scale = want.abs().max().item()
    side = torch.cuda.Stream()
    X = torch.randn(3072, 3072, device=DEV)
    first = None
    for rep in range(8):
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(rep % 3):
                ops.gemm(X, X, torch.empty_like(X), 3072, 3072, 3072, transb=True)
        C = torch.full((M, N), float('nan'), device=DEV)
        ops.gemm(A, Bm, C, M, N, K, transb=True)
        torch.cuda.synchronize()
        assert (C.double() - want).abs().max().item() <= 2e-5 * scale, rep
        if first is None:
            first = C.clone()
        else:
            assert torch.equal(first, C), 'launch %d differs from the first one' % rep
    # mapped rows: every third physical row is padding
    if M >= 256:
        phys = M + M // 2
        valid = torch.tensor([r for r in range(phys) if r % 3 != 2][:M], dtype=torch.int32, device=DEV)
        Ap = torch.full((phys, K), float('nan'), device=DEV)
        Ap[valid.long()] = A
        wsb = lib.query('re2e_gemm_workspace_bytes', 0, 1, M, N, K)
        ws = ops.workspace(wsb, A.device, 'gemm') if wsb else None
        outs = []
        for rep in range(4):
            with torch.cuda.stream(side):
                for _ in range(rep % 3):
                    ops.gemm(X, X, torch.empty_like(X), 3072, 3072, 3072, transb=True)
            Cp = torch.full((phys, N), 7.0, device=DEV)
            if not lib.call_supported('re2e_gemm_nt_rows', M, N, K, Ap.data_ptr(), K, Bm.data_ptr(), K, Cp.data_ptr(), N, None, None, lib.ACT_NONE, 0.0,
                                      valid.data_ptr(), 0, phys, ws.data_ptr() if ws is not None else None, wsb):
                pytest.skip('the pipeline declines this shape')
            torch.cuda.synchronize()
            outs.append(Cp)
        assert (outs[0][valid.long()].double() - want).abs().max().item() <= 2e-5 * scale
        untouched = torch.ones(phys, dtype=torch.bool, device=DEV)
        untouched[valid.long()] = False
        assert bool((outs[0][untouched] == 7.0).all())
        for o in outs[1:]:
            assert torch.equal(o, outs[0])


@pytest.mark.parametrize('M,N,K', [(64, 96, 5000), (257, 130, 77), (1200, 812, 1312), (8, 4, 40000)])
def test_gemm_tn_splitk(M, N, K):
    ops, lib = _ops()
    A, Bm, C0 = rnd(K, M), rnd(K, N, seed=1), rnd(M, N, seed=2)
    c = C0.to(DEV).clone()
    ops.gemm(A.to(DEV), Bm.to(DEV), c, M, N, K, transa=True, beta=1.0)
    close('tn', c, A.t() @ Bm + C0, tol=2e-4)
    c2 = C0.to(DEV).clone()
    ops.gemm(A.to(DEV), Bm.to(DEV), c2, M, N, K, transa=True, beta=1.0)
    assert torch.equal(c, c2), 'split-K reduction must be bitwise reproducible'


def test_linear_autograd():
    ops, lib = _ops()
    x, W, b = rnd(4, 9, 37), rnd(21, 37, seed=1), rnd(21, seed=2)
    xr, Wr, br = [t.clone().requires_grad_(True) for t in (x, W, b)]
    yr = torch.tanh(F.linear(xr, Wr, br))
    (yr * rnd(4, 9, 21, seed=5)).sum().backward()
    xg = x.to(DEV).requires_grad_(True)
    Wg, bg = torch.nn.Parameter(W.to(DEV)), torch.nn.Parameter(b.to(DEV))
    y = ops.linear(xg, Wg, bg, 'tanh')
    (y * rnd(4, 9, 21, seed=5).to(DEV)).sum().backward()
    close('y', y, yr)
    close('dx', xg.grad, xr.grad)
    close('dW', Wg.grad, Wr.grad)
    close('db', bg.grad, br.grad)
    # second backward accumulates into .grad (beta=1 path)
    y = ops.linear(xg, Wg, bg, 'tanh')
    (y * rnd(4, 9, 21, seed=5).to(DEV)).sum().backward()
    close('dW x2', Wg.grad, 2 * Wr.grad)


def test_stream_role_changes_tiles_not_results():
    """re2e_stream_role: a FILLER stream gets the 4-wave engine tiles; the product is the same"""
    ops, lib = _ops()
    M, N, K = 4100, 300, 200
    A, Bm = rnd(M, K).to(DEV), rnd(N, K, seed=1).to(DEV)
    ref = torch.empty(M, N, device=DEV)
    ops.gemm(A, Bm, ref, M, N, K, transb=True)
    st = torch.cuda.Stream()
    lib.set_stream_role(st, True)
    lib.set_stream_role(st, True)                       # idempotent
    got = torch.empty(M, N, device=DEV)
    st.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(st):
        ops.gemm(A, Bm, got, M, N, K, transb=True)
    st.synchronize()
    lib.set_stream_role(st, False)
    close('filler-stream gemm', got, ref, tol=1e-6)
    with pytest.raises(lib.Re2eError):
        if lib.load().re2e_stream_role(st.cuda_stream, 7) != 0:
            raise lib.Re2eError(lib.load().re2e_last_error().decode())


def test_linear_input_grad_many_rows():
    """M >= 2048 rows: the input gradient runs as dz (W^T)^T with a transposed copy of the weight (ops.gemm_input_grad)"""
    ops, lib = _ops()
    x, W, b = rnd(9, 260, 36), rnd(44, 36, seed=1), rnd(44, seed=2)
    g = rnd(9, 260, 44, seed=5)
    xr, Wr, br = [t.clone().requires_grad_(True) for t in (x, W, b)]
    (F.linear(xr, Wr, br) * g).sum().backward()
    xg = x.to(DEV).requires_grad_(True)
    Wg, bg = torch.nn.Parameter(W.to(DEV)), torch.nn.Parameter(b.to(DEV))
    (ops.linear(xg, Wg, bg, None) * g.to(DEV)).sum().backward()
    close('dx', xg.grad, xr.grad)
    close('dW', Wg.grad, Wr.grad, tol=2e-4)
    dz = g.view(-1, 44).to(DEV).contiguous()
    dx0 = rnd(9 * 260, 36, seed=7)
    dx = dx0.to(DEV).clone()
    ops.gemm_input_grad(dz, Wg.data, dx, 9 * 260, 36, 44, beta=1.0)
    close('dx beta', dx, dx0 + g.view(-1, 44) @ W)


@pytest.mark.parametrize('shape', [(3, 11, 16, 257, [11, 7, 4]), (3, 700, 16, 37, [700, 512, 33])])
def test_mask_fc(shape):
    """Second shape: >= 2048 rows and an odd width -- the backward writes d(linear) with padded rows (re2e_mask_mul_bwd_ld) and runs
    both products over the padded width."""
    ops, lib = _ops()
    B, T, K, N, lens = shape
    proj, W, mix = rnd(B, T, K), rnd(N, K, seed=1), rnd(B, T, N, seed=2).abs()
    pr, Wr = proj.clone().requires_grad_(True), W.clone().requires_grad_(True)
    valid = (torch.arange(T).unsqueeze(0) < torch.tensor(lens).view(-1, 1)).unsqueeze(-1).float()
    ref = torch.sigmoid(F.linear(pr, Wr)) * valid * mix
    (ref * rnd(B, T, N, seed=3)).sum().backward()
    pg = proj.to(DEV).requires_grad_(True)
    Wg = torch.nn.Parameter(W.to(DEV))
    out, mask = ops.mask_fc(pg, Wg, mix.to(DEV), torch.tensor(lens, dtype=torch.int32, device=DEV), T)
    (out * rnd(B, T, N, seed=3).to(DEV)).sum().backward()
    close('out', out, ref)
    assert (out[1, lens[1]:] == 0).all() and (out[2, lens[2]:] == 0).all()
    close('dproj', pg.grad, pr.grad, tol=2e-4 if T > 100 else 1e-4)
    close('dW', Wg.grad, Wr.grad, tol=2e-4 if T > 100 else 1e-4)


def test_fbank():
    ops, lib = _ops()
    from oracle import nets
    from robust_e2e_gan_amd.model.feat_model import band_from_matrix, mel_matrix
    W = torch.from_numpy(mel_matrix())
    x = rnd(3, 13, 257, scale=20.0).abs()
    x[1, 9:] = 0
    cm = torch.stack([torch.linspace(10, 14, 80), torch.linspace(0.3, 0.6, 80)])
    xr = x.clone().requires_grad_(True)
    y0, y1 = nets.fbank_forward(xr, W), nets.fbank_forward(xr, W, cm)
    g0, g1 = rnd(3, 13, 80, seed=1), rnd(3, 13, 80, seed=2)
    ((y0 * g0).sum() + (y1 * g1).sum()).backward()
    band = band_from_matrix(W, DEV)
    xg = x.to(DEV).requires_grad_(True)
    raw, nrm = ops.fbank(xg, band, cm.to(DEV), True, True)
    ((raw * g0.to(DEV)).sum() + (nrm * g1.to(DEV)).sum()).backward()
    close('raw', raw, y0, tol=1e-5)
    close('norm', nrm, y1, tol=1e-5)
    assert torch.allclose(raw[1, 9:].cpu(), torch.full((4, 80), math.log(1e-7)))
    close('dx', xg.grad, xr.grad, tol=1e-4)


def test_fbank_keeps_a_nan_a_nan():
    """feat_model.py:130 is torch.clamp(min=1e-7): a NaN bin stays a NaN in every filter that covers it (and the joint step's NaN gate,
    joint_train.py:189-193, then sees it).  `x > c ? x : c` or fmaxf would turn it into log(1e-7) -- round 4 found the enhancer being
    updated with NaN gradients behind exactly that."""
    ops, lib = _ops()
    from oracle import nets
    from robust_e2e_gan_amd.model.feat_model import band_from_matrix, mel_matrix
    W = torch.from_numpy(mel_matrix())
    x = rnd(2, 40, 257, scale=20.0).abs()
    x[1, 7, 100] = float('nan')
    x[0, 33, :] = float('nan')
    cm = torch.stack([torch.linspace(10, 14, 80), torch.linspace(0.3, 0.6, 80)])
    ref = nets.fbank_forward(x, W, cm)                    # dense x^2 . W: NaN * 0 = NaN poisons ALL 80 filters of a frame with a NaN bin
    band = band_from_matrix(W, DEV)
    _, nrm = ops.fbank(x.to(DEV), band, cm.to(DEV), False, True)
    got = torch.isnan(nrm.cpu())
    # the banded kernel multiplies a bin only with the filters of the 32-filter tiles whose bin range holds it: every filter that COVERS a
    # NaN bin is NaN (and maybe its tile neighbours), never a filter of a frame without one -- between the band-exact and the dense
    # product's pattern, and never empty where the reference's is not, which is what the NaN gate needs
    covers = (torch.isnan(x).float() @ (W != 0).float()) > 0
    assert (covers <= got).all() and (got <= torch.isnan(ref)).all() and got[0, 33].all() and int(got[1, 7].sum()) > 0
    assert torch.equal(got.any(-1), torch.isnan(ref).any(-1))
    ok = ~torch.isnan(ref)
    close('finite part', nrm.cpu()[ok], ref[ok], tol=1e-5)
    z = (x * x) @ W
    y = torch.empty(2, 40, 80, device=DEV)
    lib.call('re2e_logclamp_fwd', z.to(DEV).contiguous().data_ptr(), cm.to(DEV).data_ptr(), 80, 80, y.data_ptr())      # the trainable-matrix path's kernel
    assert torch.equal(torch.isnan(y.cpu()), torch.isnan(ref))


@pytest.mark.parametrize('kind', [0, 1, 2])
def test_mean_loss(kind):
    ops, lib = _ops()
    a, b = rnd(5, 7, 80, scale=2.0), rnd(5, 7, 80, seed=1, scale=2.0)
    fn = [F.mse_loss, F.l1_loss, F.smooth_l1_loss][kind]
    ar = a.clone().requires_grad_(True)
    lr = fn(ar, b)
    (lr * 1.7).backward()
    ag = a.to(DEV).requires_grad_(True)
    l = ops.mean_loss(ag, b.to(DEV), 0.0, kind)
    (l * 1.7).backward()
    close('loss', l.view(1), lr.view(1), tol=1e-5)
    close('da', ag.grad, ar.grad, tol=1e-5)
    if kind == 0:
        l1 = ops.mean_loss(ag, None, 1.0, 0)
        close('lsgan', l1.view(1), ((a - 1.0) ** 2).mean().view(1), tol=1e-5)


CONVS = [  # N, H, W, Cin, Cout, k, stride, pad, act, bias
    (3, 37, 20, 1, 8, 3, 1, 1, 'relu', True),
    (2, 16, 12, 64, 64, 3, 1, 1, 'relu', True),
    (2, 9, 10, 64, 128, 3, 1, 1, None, True),
    (3, 37, 80, 1, 8, 4, 2, 1, 'lrelu', True),
    (3, 18, 40, 8, 16, 4, 2, 1, None, False),
    (2, 18, 40, 64, 128, 4, 2, 1, None, False),
    (3, 9, 9, 16, 32, 4, 1, 1, None, False),
    (3, 8, 8, 32, 1, 4, 1, 1, None, True),
    (2, 7, 5, 6, 10, 3, 1, 1, 'relu', True),
    # thin-channel direct kernels (thinconv.hip): Cin == 1 / Cout == 1 at the VGG / discriminator widths,
    # plus widths the thin weight-gradient kernels do not cover (fall back to the implicit GEMM)
    (2, 21, 16, 1, 64, 3, 1, 1, 'relu', True),
    (2, 37, 20, 1, 64, 4, 2, 1, 'lrelu', True),
    (2, 12, 9, 512, 1, 4, 1, 1, None, True),
    (2, 11, 7, 64, 1, 3, 1, 1, None, False),
    (2, 9, 8, 1, 12, 3, 1, 1, None, True),
    (2, 9, 8, 12, 1, 3, 1, 1, 'lrelu', True),
    # the row-tile Cout == 1 kernel (16 lanes per pixel): full-width rows, a width that is not a multiple of 16, ragged last row tile
    (2, 19, 80, 1, 64, 3, 1, 1, 'relu', True),
    (1, 13, 23, 64, 1, 3, 1, 1, None, True),
    # halo-patch 3x3 kernels (conv3x3.hip): C % 16 == 0 and Cout % 64 == 0; 16x16 patches (ragged in both directions) and
    # the 32x8 patches chosen for narrow images; one / two 64-channel groups; 1, 4 and 8 channel chunks
    (1, 35, 80, 64, 64, 3, 1, 1, 'relu', True),
    (1, 100, 40, 128, 64, 3, 1, 1, None, False),
    (2, 33, 8, 16, 64, 3, 1, 1, 'relu', True),
    (1, 67, 40, 64, 128, 3, 1, 1, 'relu', True),
    (3, 5, 3, 32, 128, 3, 1, 1, None, True),
    # Winograd F(2x2,4x4) (wino44.hip): 4x4 / stride 1 / pad 1, >= 64 channels both ways; odd output sizes (ragged tiles)
    (2, 50, 48, 64, 128, 4, 1, 1, None, False),
    (1, 67, 65, 128, 64, 4, 1, 1, None, False),
]


@pytest.mark.parametrize('cfg', CONVS)
def test_conv2d(cfg):
    ops, lib = _ops()
    N, H, W, Cin, Cout, k, s, p, act, has_b = cfg
    x, Wt = rnd(N, Cin, H, W), rnd(Cout, Cin, k, k, seed=1, scale=1.0 / math.sqrt(Cin * k * k))
    b = rnd(Cout, seed=2) if has_b else None
    xr, Wr = x.clone().requires_grad_(True), Wt.clone().requires_grad_(True)
    br = b.clone().requires_grad_(True) if has_b else None
    yr = F.conv2d(xr, Wr, br, stride=s, padding=p)
    yr = {'relu': F.relu, 'lrelu': lambda t: F.leaky_relu(t, 0.2), None: lambda t: t}[act](yr)
    go = rnd(*yr.shape, seed=3)
    (yr * go).sum().backward()
    xg = x.permute(0, 2, 3, 1).contiguous().to(DEV).requires_grad_(True)
    Wg = torch.nn.Parameter(Wt.to(DEV))
    bg = torch.nn.Parameter(b.to(DEV)) if has_b else None
    y = ops.conv2d(xg, Wg, bg, s, p, act)
    (y * go.permute(0, 2, 3, 1).contiguous().to(DEV)).sum().backward()
    close('y', y.permute(0, 3, 1, 2), yr, tol=2e-4)
    close('dx', xg.grad.permute(0, 3, 1, 2), xr.grad, tol=2e-4)
    close('dW', Wg.grad, Wr.grad, tol=2e-4)
    if has_b:
        close('db', bg.grad, br.grad, tol=2e-4)


def test_conv3x3_halo_equals_general_engine_and_accumulates():
    """The halo-patch kernel against the general implicit-GEMM engine on the same call (RE2E_NO_HALO is read once per
    process, so the general engine is reached through a geometry the halo kernel declines: the same convolution with the
    input channels padded to C+4 ... is not the same call; instead compare with F.conv2d) and the beta = 1 epilogue."""
    ops, lib = _ops()
    N, H, W, C, K = 2, 21, 24, 32, 64
    x, Wt, b = rnd(N, H, W, C), rnd(K, C, 3, 3, seed=1, scale=0.1), rnd(K, seed=2)
    ref = F.conv2d(x.permute(0, 3, 1, 2), Wt, b, padding=1).permute(0, 2, 3, 1)
    xg, Wg, bg = x.to(DEV), Wt.to(DEV), b.to(DEV)
    wg = torch.empty(K, 3, 3, C, device=DEV)
    lib.call('re2e_conv_weight_gather', Wg.data_ptr(), wg.data_ptr(), K, C, 3, 3, 0, 3, 3, 0, 0, 1)
    y0 = rnd(N, H, W, K, seed=5).to(DEV)
    y = y0.clone()
    lib.call('re2e_conv_igemm', xg.data_ptr(), N, H, W, C, wg.data_ptr(), K, 3, 3, H, W, 1, 1, 1, 1, -1, -1, y.data_ptr(), H, W, 1, 1, 0, 0,
             bg.data_ptr(), lib.ACT_NONE, 1.0)
    close('beta=1', y, ref + y0.cpu(), tol=2e-4)
    lib.call('re2e_conv_igemm', xg.data_ptr(), N, H, W, C, wg.data_ptr(), K, 3, 3, H, W, 1, 1, 1, 1, -1, -1, y.data_ptr(), H, W, 1, 1, 0, 0,
             bg.data_ptr(), lib.ACT_RELU, 0.0)
    close('relu', y, F.relu(ref), tol=2e-4)


@pytest.mark.parametrize('C', [6, 8, 64])        # scalar path (C % 4 != 0) and the four-channels-per-thread path
def test_pool_and_pack(C):
    ops, lib = _ops()
    x = rnd(3, C, 37, 21)                     # NCHW, odd sizes: ceil-mode windows at both edges
    x = torch.round(x * 2) / 2                # many ties inside a window: the first maximum in row-major order must win
    xr = x.clone().requires_grad_(True)
    pr = F.max_pool2d(xr, 2, stride=2, ceil_mode=True)
    lens = [19, 12, 5]
    T2 = pr.shape[2]
    ref = pr.transpose(1, 2).contiguous().view(3, T2, -1)
    ref = torch.stack([torch.cat([ref[i, :lens[i]], torch.zeros(T2 - lens[i], ref.shape[2])]) for i in range(3)])
    go = rnd(3, T2, ref.shape[2], seed=4)
    (ref * go).sum().backward()
    xg = x.permute(0, 2, 3, 1).contiguous().to(DEV).requires_grad_(True)
    pg = ops.maxpool2(xg)
    out = ops.vgg_pack(pg, torch.tensor(lens, dtype=torch.int32, device=DEV))     # (T,N,C*F)
    (out * go.transpose(0, 1).contiguous().to(DEV)).sum().backward()
    close('pool', pg.permute(0, 3, 1, 2), pr, tol=0, atol=0)
    close('pack', out.transpose(0, 1), ref, tol=0, atol=0)
    close('dx', xg.grad.permute(0, 3, 1, 2), xr.grad, tol=1e-6)


@pytest.mark.parametrize('Cin,Cout', [(16, 64), (3, 6)])      # halo-patch conv + float4 pool / general engine + scalar pool
def test_conv_relu_pool_backward_in_pool(Cin, Cout):
    """conv2d(act=relu, relu_bwd_in_pool) -> maxpool2(relu_in): the pool's index byte carries the ReLU mask, the convolution's
    backward skips its activation pass; every gradient equals conv -> relu -> max_pool2d(ceil_mode) of torch"""
    ops, lib = _ops()
    x, W, b = rnd(2, Cin, 21, 13), rnd(Cout, Cin, 3, 3, seed=1, scale=0.2), rnd(Cout, seed=2)
    xr, Wr, br = [t.clone().requires_grad_(True) for t in (x, W, b)]
    yr = F.max_pool2d(F.relu(F.conv2d(xr, Wr, br, padding=1)), 2, stride=2, ceil_mode=True)
    go = rnd(*yr.shape, seed=3)
    (yr * go).sum().backward()
    xg = x.permute(0, 2, 3, 1).contiguous().to(DEV).requires_grad_(True)
    Wg, bg = torch.nn.Parameter(W.to(DEV)), torch.nn.Parameter(b.to(DEV))
    y = ops.maxpool2(ops.conv2d(xg, Wg, bg, 1, 1, 'relu', relu_bwd_in_pool=True), relu_in=True)
    (y * go.permute(0, 2, 3, 1).contiguous().to(DEV)).sum().backward()
    close('y', y.permute(0, 3, 1, 2), yr, tol=2e-5)
    close('dx', xg.grad.permute(0, 3, 1, 2), xr.grad, tol=2e-5)
    close('dW', Wg.grad, Wr.grad, tol=2e-5)
    close('db', bg.grad, br.grad, tol=2e-5)
    # the unfused composition gives the same numbers
    xg2 = x.permute(0, 2, 3, 1).contiguous().to(DEV).requires_grad_(True)
    W2, b2 = torch.nn.Parameter(W.to(DEV)), torch.nn.Parameter(b.to(DEV))
    y2 = ops.maxpool2(ops.conv2d(xg2, W2, b2, 1, 1, 'relu'))
    (y2 * go.permute(0, 2, 3, 1).contiguous().to(DEV)).sum().backward()
    assert torch.equal(y, y2) and torch.equal(xg.grad, xg2.grad) and torch.equal(Wg.grad, W2.grad)
    close('db vs unfused', bg.grad, b2.grad, tol=1e-6)


@pytest.mark.parametrize('N,H,W,Cin,Cout', [(2, 37, 19, 16, 64), (1, 40, 83, 64, 64), (2, 33, 8, 128, 128), (2, 9, 7, 6, 10)])
def test_conv_relu_pool_one_launch(N, H, W, Cin, Cout):
    """conv2d(..., act=relu, pool=True): re2e_conv3x3_relu_pool (16x16 and 32x8 patches, ragged / odd edges, two channel groups) or,
    for the last shape, the unfused fallback -- output and index bytes identical to conv -> maxpool2(relu_in), gradients as torch's"""
    ops, lib = _ops()
    x, Wt, b = rnd(N, Cin, H, W), rnd(Cout, Cin, 3, 3, seed=1, scale=0.15), rnd(Cout, seed=2, scale=0.3)
    xr, Wr, br = [t.clone().requires_grad_(True) for t in (x, Wt, b)]
    yr = F.max_pool2d(F.relu(F.conv2d(xr, Wr, br, padding=1)), 2, stride=2, ceil_mode=True)
    go = rnd(*yr.shape, seed=3)
    (yr * go).sum().backward()
    outs = []
    for fused in (True, False):
        xg = x.permute(0, 2, 3, 1).contiguous().to(DEV).requires_grad_(True)
        Wg, bg = torch.nn.Parameter(Wt.to(DEV)), torch.nn.Parameter(b.to(DEV))
        if fused:
            y = ops.conv2d(xg, Wg, bg, 1, 1, 'relu', pool=True)
        else:
            y = ops.maxpool2(ops.conv2d(xg, Wg, bg, 1, 1, 'relu', relu_bwd_in_pool=True), relu_in=True)
        (y * go.permute(0, 2, 3, 1).contiguous().to(DEV)).sum().backward()
        outs.append((y, xg.grad, Wg.grad, bg.grad))
    y, dx, dW, db = outs[0]
    close('y', y.permute(0, 3, 1, 2), yr, tol=3e-5)
    close('dx', dx.permute(0, 3, 1, 2), xr.grad, tol=3e-5)
    close('dW', dW, Wr.grad, tol=3e-5)
    close('db', db, br.grad, tol=3e-5)
    for a, bb in zip(outs[0], outs[1]):
        assert torch.equal(a, bb), 'fused conv+pool must reproduce conv -> pool bit for bit'


@pytest.mark.parametrize('C0,C1,C2', [(1, 64, 64), (16, 64, 128), (3, 6, 10)])   # VGG's first pair; halo epilogue mask; second-pass mask
def test_conv_relu_conv_backward_through_relu(C0, C1, C2):
    """conv2d(relu, relu_bwd_in_next) -> conv2d(relu, x_is_relu_out, relu_bwd_in_pool) -> maxpool2(relu_in): no activation-backward
    pass anywhere; every gradient equals torch's conv -> relu -> conv -> relu -> max_pool2d"""
    ops, lib = _ops()
    x = rnd(2, C0, 37, 19)
    W1, b1, W2, b2 = rnd(C1, C0, 3, 3, seed=1, scale=0.3), rnd(C1, seed=2, scale=0.3), rnd(C2, C1, 3, 3, seed=3, scale=0.1), rnd(C2, seed=4, scale=0.3)
    ref = [t.clone().requires_grad_(True) for t in (x, W1, b1, W2, b2)]
    yr = F.max_pool2d(F.relu(F.conv2d(F.relu(F.conv2d(ref[0], ref[1], ref[2], padding=1)), ref[3], ref[4], padding=1)), 2, stride=2, ceil_mode=True)
    go = rnd(*yr.shape, seed=5)
    (yr * go).sum().backward()
    xg = x.permute(0, 2, 3, 1).contiguous().to(DEV).requires_grad_(True)
    P = [torch.nn.Parameter(t.to(DEV)) for t in (W1, b1, W2, b2)]
    h = ops.conv2d(xg, P[0], P[1], 1, 1, 'relu', relu_bwd_in_next=True)
    h = ops.conv2d(h, P[2], P[3], 1, 1, 'relu', relu_bwd_in_pool=True, x_is_relu_out=True)
    y = ops.maxpool2(h, relu_in=True)
    (y * go.permute(0, 2, 3, 1).contiguous().to(DEV)).sum().backward()
    close('y', y.permute(0, 3, 1, 2), yr, tol=3e-5)
    close('dx', xg.grad.permute(0, 3, 1, 2), ref[0].grad, tol=3e-5)
    for name, g, r in zip(('dW1', 'db1', 'dW2', 'db2'), P, ref[1:]):
        close(name, g.grad, r.grad, tol=3e-5)


@pytest.mark.parametrize('N,C,H,W', [(3, 16, 9, 5), (3, 6, 9, 5), (5, 128, 67, 31)])     # float4 path, scalar path, many rows per chunk
def test_bn_lrelu(N, C, H, W):
    ops, lib = _ops()
    x = rnd(N, C, H, W, scale=2.0) + 0.5
    bn = torch.nn.BatchNorm2d(C)
    bn.weight.data.normal_(1, 0.02)
    bn.bias.data.normal_(0, 0.1)
    xr = x.clone().requires_grad_(True)
    yr = F.leaky_relu(bn(xr), 0.2)
    go = rnd(N, C, H, W, seed=3)
    (yr * go).sum().backward()
    xg = x.permute(0, 2, 3, 1).contiguous().to(DEV).requires_grad_(True)
    gam, bet = torch.nn.Parameter(bn.weight.data.clone().to(DEV)), torch.nn.Parameter(bn.bias.data.clone().to(DEV))
    rm, rv = torch.zeros(C, device=DEV), torch.ones(C, device=DEV)
    y = ops.bn_lrelu(xg, gam, bet, rm, rv, True)
    (y * go.permute(0, 2, 3, 1).contiguous().to(DEV)).sum().backward()
    close('y', y.permute(0, 3, 1, 2), yr, tol=1e-5)
    close('dx', xg.grad.permute(0, 3, 1, 2), xr.grad, tol=1e-4)
    close('dgamma', gam.grad, bn.weight.grad, tol=1e-4)
    close('dbeta', bet.grad, bn.bias.grad, tol=1e-4)
    close('running_mean', rm, bn.running_mean, tol=1e-5)
    close('running_var', rv, bn.running_var, tol=1e-5)


@pytest.mark.parametrize('B,T,I,H,lens', [(3, 11, 20, 16, [11, 7, 4]), (40, 9, 33, 64, None), (5, 6, 257, 24, [6, 6, 5, 2, 1]),
                                          # shapes the persistent (one launch per sequence) forward covers: (waves, k-groups per wave)
                                          (40, 23, 17, 256, None), (33, 14, 9, 320, None), (64, 12, 12, 512, None), (5, 19, 8, 32, [19, 19, 7, 2, 1]),
                                          (7, 13, 8, 128, None), (2, 10, 8, 192, [10, 4]),
                                          (40, 30, 257, 32, None)])       # >= 1024 rows with an odd input width: zero-padded copies of x / W_ih
def test_bilstm(B, T, I, H, lens):
    ops, lib = _ops()
    if lens is None:
        lens = sorted([1 + (i * 7) % T for i in range(B)], reverse=True)
        lens[0] = T
    lstm = torch.nn.LSTM(I, H, 1, batch_first=True, bidirectional=True)
    x = rnd(B, T, I)
    for b, l in enumerate(lens):
        x[b, l:] = 0
    xr = x.clone().requires_grad_(True)
    pk = torch.nn.utils.rnn.pack_padded_sequence(xr, torch.tensor(lens), batch_first=True)
    yr, _ = torch.nn.utils.rnn.pad_packed_sequence(lstm(pk)[0], batch_first=True, total_length=T)
    go = rnd(B, T, 2 * H, seed=9)
    (yr * go).sum().backward()
    names = ['weight_ih_l0', 'weight_hh_l0', 'bias_ih_l0', 'bias_hh_l0']
    ws = [torch.nn.Parameter(getattr(lstm, n + sfx).data.clone().to(DEV)) for sfx in ('', '_reverse') for n in names]
    xg = x.to(DEV).requires_grad_(True)
    y = ops.bilstm(ops.transpose01(xg), torch.tensor(lens, dtype=torch.int32, device=DEV), ws)
    (y * go.transpose(0, 1).contiguous().to(DEV)).sum().backward()
    close('y', y.transpose(0, 1), yr, tol=1e-4)
    close('dx', xg.grad, xr.grad, tol=2e-4)
    i = 0
    for sfx in ('', '_reverse'):
        for n in names:
            close(n + sfx, ws[i].grad, getattr(lstm, n + sfx).grad, tol=3e-4)
            i += 1


def _poison_free_blocks(nbytes):
    """NaN in the caching allocator's free blocks: what the next ``torch.empty`` of that size class hands out."""
    junk = [torch.full((nbytes // 4,), float('nan'), device=DEV) for _ in range(3)]
    torch.cuda.synchronize()
    del junk


# x W^T over the valid rows of a ragged time-major batch (re2e_gemm_nt_rows + re2e_fill_rows; ops.row_maps / gemm_rows): rows outside the map are
# not touched (or zero-filled), rows inside equal the product over all rows.  Shapes: the enhancer's and the BLSTMP's row spaces, a K tail, ragged
# N edge, a stream-K tail, beta = 1.
@pytest.mark.parametrize('T,B,N,K,act', [(800, 32, 1024, 260, 'none'), (200, 64, 512, 1024, 'tanh'), (97, 24, 260, 36, 'relu'), (400, 16, 2048, 512, 'none')])
def test_gemm_nt_rows(T, B, N, K, act, monkeypatch):
    ops, lib = _ops()
    from robust_e2e_gan_amd.model.e2e_common import lens_dev
    if K < ops.ROW_MAPS_MIN_K:
        # short contractions stay on all rows in the step (ops.ROW_MAPS_MIN_K: the per-tile row lookups do not pay); the kernel takes them all the same
        a0, w0, c0 = torch.zeros(T * B, K, device=DEV), torch.zeros(N, K, device=DEV), torch.ones(T * B, N, device=DEV)
        ops.gemm_rows(a0, w0, c0, N, K, ops.row_maps(lens_dev([max(1, int(round(T * (1 - 0.3 * i / (B - 1))))) for i in range(B)], DEV), T, B))
        assert float(c0.abs().max()) == 0.0, 'below ROW_MAPS_MIN_K the product runs over all rows'
        monkeypatch.setattr(ops, 'ROW_MAPS_MIN_K', 0)
    lens = [max(1, int(round(T * (1 - 0.3 * i / (B - 1))))) for i in range(B)]
    maps = ops.row_maps(lens_dev(lens, DEV), T, B)
    assert maps is not None and maps.nv == sum(lens) and maps.nv + maps.ni == T * B
    ok = (torch.arange(T)[:, None] < torch.tensor(lens)[None, :]).reshape(-1)
    assert torch.equal(maps.valid.cpu().long(), torch.nonzero(ok).flatten()) and torch.equal(maps.invalid.cpu().long(), torch.nonzero(~ok).flatten())
    A, W, b1, b2, C0 = rnd(T * B, K), rnd(N, K, seed=1), rnd(N, seed=2), rnd(N, seed=3), rnd(T * B, N, seed=4)
    a, w = A.to(DEV), W.to(DEV)
    fn = {'none': lambda v: v, 'tanh': torch.tanh, 'relu': torch.relu}[act]
    want = fn(A.double() @ W.double().t() + b1.double() + b2.double())
    c = C0.to(DEV).clone()
    ops.gemm_rows(a, w, c, N, K, maps, bias=b1.to(DEV), bias2=b2.to(DEV), act=ops.ACT[None if act == 'none' else act])
    scale = (A.double() @ W.double().t()).abs().max().item()
    close('valid rows', c.cpu()[ok], want.float()[ok], tol=0.0, atol=2e-5 * scale)
    assert torch.equal(c.cpu()[~ok], C0[~ok]), 'rows outside the map were touched'
    c2 = C0.to(DEV).clone()
    ops.gemm_rows(a, w, c2, N, K, maps, act=ops.ACT[None if act == 'none' else act], fill=True)
    if N % 4 == 0:
        assert float(c2.cpu()[~ok].abs().max()) == 0.0
    c3 = C0.to(DEV).clone()
    ops.gemm_rows(a, w, c3, N, K, maps, beta=1.0)
    close('beta', c3.cpu()[ok], (A.double() @ W.double().t() + C0.double()).float()[ok], tol=0.0, atol=2e-5 * scale)
    assert torch.equal(c3.cpu()[~ok], C0[~ok])
    c4 = C0.to(DEV).clone()
    ops.gemm_rows(a, w, c4, N, K, maps, beta=1.0)
    assert torch.equal(c3, c4), 'bitwise repeatable'


# dy^T x over the valid rows (re2e_gemm_tn_rows: the contraction walks a row map, one table look-up per thread and k-tile, a tile ahead): equal to the
# sum over all rows when dy is zero in the others, and it must not READ the others (NaN there).  Shapes: split-K with the K slice = XCD order, one
# slice, ragged M / N edges, a strided x operand (the h_{t-1} view of a recurrent layer), beta = 1.
@pytest.mark.parametrize('T,B,M,N,ldb', [(800, 32, 1024, 260, None), (200, 64, 2048, 512, None), (97, 24, 132, 68, 72), (300, 16, 512, 256, 512)])
def test_gemm_tn_rows(T, B, M, N, ldb):
    ops, lib = _ops()
    from robust_e2e_gan_amd.model.e2e_common import lens_dev
    lens = [max(1, int(round(T * (1 - 0.3 * i / (B - 1))))) for i in range(B)]
    maps = ops.row_maps(lens_dev(lens, DEV), T, B)
    assert maps is not None
    ok = (torch.arange(T)[:, None] < torch.tensor(lens)[None, :]).reshape(-1)
    ldb = ldb or N
    A, X, C0 = rnd(T * B, M), rnd(T * B, ldb, seed=1), rnd(M, N, seed=2)
    want = A[ok].double().t() @ X[ok][:, :N].double()
    A[~ok] = float('nan')
    X[~ok] = float('nan')
    a, x = A.to(DEV), X.to(DEV)
    c = torch.empty(M, N, device=DEV)
    ops.gemm_tn_rows(a, x, c, M, N, maps, ldb=ldb)
    close('dy^T x over the map', c, want.float(), tol=2e-5)
    c2 = C0.to(DEV).clone()
    ops.gemm_tn_rows(a, x, c2, M, N, maps, beta=1.0, ldb=ldb)
    close('beta', c2, (want + C0.double()).float(), tol=2e-5)
    c3 = C0.to(DEV).clone()
    ops.gemm_tn_rows(a, x, c3, M, N, maps, beta=1.0, ldb=ldb)
    assert torch.equal(c2, c3), 'bitwise repeatable'


@pytest.mark.parametrize('B,T,I,H', [(32, 40, 260, 256), (64, 30, 512, 512), (16, 64, 257, 64)])
def test_bilstm_and_projection_over_valid_rows_only(B, T, I, H):
    """ops.bilstm + the BLSTMP projection with row maps (lengths registered through ``lens_dev``): the padded (t, b) rows of the input projections
    are never written -- here they hold NaN (poisoned allocator blocks) -- and neither results nor gradients may notice; against a packed
    nn.LSTM + Linear + tanh on the CPU, and against the same ops with the maps switched off."""
    ops, lib = _ops()
    from robust_e2e_gan_amd.model.e2e_common import lens_dev
    lens = [max(1, int(round(T * (1 - 0.45 * i / (B - 1))))) for i in range(B)]
    lstm = torch.nn.LSTM(I, H, 1, batch_first=True, bidirectional=True)
    lin = torch.nn.Linear(2 * H, 128)
    x = rnd(B, T, I)
    for b, l in enumerate(lens):
        x[b, l:] = 0
    xr = x.clone().requires_grad_(True)
    pk = torch.nn.utils.rnn.pack_padded_sequence(xr, torch.tensor(lens), batch_first=True)
    yr, _ = torch.nn.utils.rnn.pad_packed_sequence(lstm(pk)[0], batch_first=True, total_length=T)
    pr = torch.tanh(lin(yr))
    msk = (torch.arange(T)[None, :] < torch.tensor(lens)[:, None]).float()[:, :, None]
    go = rnd(B, T, 128, seed=9)                             # a gradient in the padded rows too: upstream's Linear runs over them (x = 0 there),
    #                                                         so they hold tanh(bias) and feed the bias gradient (e2e_encoder.py:145-147)
    (pr * go).sum().backward()
    names = ['weight_ih_l0', 'weight_hh_l0', 'bias_ih_l0', 'bias_hh_l0']

    def run(with_maps):
        ws = [torch.nn.Parameter(getattr(lstm, n + sfx).data.clone().to(DEV)) for sfx in ('', '_reverse') for n in names]
        Wp, bp = torch.nn.Parameter(lin.weight.data.clone().to(DEV)), torch.nn.Parameter(lin.bias.data.clone().to(DEV))
        xg = x.to(DEV).requires_grad_(True)
        ld = lens_dev(lens, DEV) if with_maps else torch.tensor(lens, dtype=torch.int32, device=DEV)
        maps = ops.row_maps(ld, T, B) if with_maps else None
        assert (maps is not None) == with_maps
        _poison_free_blocks(T * B * 4 * H * 4)
        y = ops.bilstm(ops.transpose01(xg), ld, ws)
        p_ = ops.linear(y, Wp, bp, 'tanh', maps=maps)
        (p_ * go.transpose(0, 1).contiguous().to(DEV)).sum().backward()
        return y, p_, xg.grad, [w.grad for w in ws] + [Wp.grad, bp.grad]
    y, p_, dx, gs = run(True)
    close('y', y.transpose(0, 1), yr, tol=1e-4)
    close('proj (all rows: tanh(bias) in the padded ones, as upstream)', p_.transpose(0, 1), pr, tol=1e-4)
    pad_rows = p_.detach().transpose(0, 1)[(1 - msk[:, :, 0]).bool().to(DEV)]
    assert pad_rows.numel() and float((pad_rows - torch.tanh(lin.bias.data.to(DEV))[None, :]).abs().max()) <= 1e-6
    close('dx', dx, xr.grad, tol=2e-4)
    refs = [getattr(lstm, n + sfx).grad for sfx in ('', '_reverse') for n in names] + [lin.weight.grad, lin.bias.grad]
    for i, (g, r) in enumerate(zip(gs, refs)):
        close('grad %d' % i, g, r, tol=3e-4)
    y0, p0, dx0, gs0 = run(False)
    close('y vs all rows', y, y0, tol=1e-6)
    close('dx vs all rows', dx, dx0, tol=2e-5)
    for i, (g, r) in enumerate(zip(gs, gs0)):
        close('grad %d vs all rows' % i, g, r, tol=2e-5)


@pytest.mark.parametrize('B,T,F_', [(12, 400, 80), (6, 333, 40)])
def test_vgg_conv_stack_row_limits(B, T, F_):
    """VGG2L.conv_stack over a ragged batch with per-image row limits (re2e_conv3x3_wino_rows / _wgrad_rows / re2e_fill_image_rows): rows beyond
    an utterance's reach are not computed -- they hold NaN here (poisoned allocator blocks) -- and the packed output, the input gradient and every
    parameter gradient equal those of the stack that computes every row."""
    ops, lib = _ops()
    from robust_e2e_gan_amd.model import e2e_encoder as enc
    torch.manual_seed(3)
    vgg = enc.VGG2L(1).to(DEV)
    for p_ in vgg.parameters():
        p_.data.normal_(0, 0.2 if p_.dim() > 1 else 0.05)
    lens = [max(8, int(round(T * (1 - 0.7 * i / (B - 1))))) for i in range(B)]
    x = rnd(B, T, F_)
    for b, l in enumerate(lens):
        x[b, l:] = 0
    assert vgg.row_limits(lens, T, DEV) is not None
    res = []
    for on in (True, False):
        enc.ROW_LIMITS = on
        try:
            for p_ in vgg.parameters():
                p_.grad = None
            xg = x.to(DEV).requires_grad_(True)
            if on:
                _poison_free_blocks(B * T * F_ * 64 * 4)
                _poison_free_blocks(B * (T // 2) * (F_ // 2) * 128 * 4)
            y, nl = vgg.forward_tm(xg, lens)
            g = rnd(*y.shape, seed=5).to(DEV)
            (y * g).sum().backward()
            res.append((y.detach().clone(), xg.grad.clone(), [p_.grad.clone() for p_ in vgg.parameters()], nl))
        finally:
            enc.ROW_LIMITS = True
    (y1, dx1, gs1, nl1), (y0, dx0, gs0, nl0) = res
    assert nl1 == nl0
    close('packed output', y1, y0, tol=1e-6)
    close('dx', dx1, dx0, tol=2e-5)
    for i, (a, b) in enumerate(zip(gs1, gs0)):
        close('grad %d' % i, a, b, tol=2e-5)


def _fwd2_runs(B, H):
    """csrc/lstm.hip fwd2_config: the round-4 forward serves <= 16 utterances and wide layers (multiples of 64 units)."""
    return H % 64 == 0 and H // 64 in (1, 2, 4, 5, 8) and (B <= 16 or H >= 384)


@pytest.mark.parametrize('B,T,H,uw,fwd2', [(32, 60, 256, None, '1'), (40, 37, 320, None, '1'), (64, 25, 512, None, '1'), (3, 21, 32, None, '1'),
                                            (40, 19, 320, '2', '1'), (3, 9, 32, '2', '1'), (64, 11, 512, '1', '1'),       # forced backward widths
                                            # round-4 forward: <= 16 utterances (K split over four waves), 4-k-chunk counts 2 .. 16, two utterance tiles
                                            (8, 45, 256, None, '1'), (16, 33, 512, None, '1'), (12, 17, 128, None, '1'), (9, 14, 320, None, '1'),
                                            (40, 21, 64, None, '1'), (33, 29, 128, None, '1'), (70, 13, 256, None, '1'), (24, 15, 512, None, '1'),
                                            # the round-1..3 forward kernels (RE2E_LSTM_FWD2=0), still used for widths fwd2 is not built for
                                            (32, 60, 256, None, '0'), (64, 25, 512, None, '0'), (40, 37, 320, None, '0'), (8, 45, 256, None, '0')])
def test_lstm_persistent_vs_stepwise(B, T, H, uw, fwd2, monkeypatch):
    """K4 forward: the persistent kernel (workgroups resident for the whole sequence, h_t handed over in-launch: one-bit-tagged values,
    round 4; tagged 8-byte granules, rounds 1-3) against the launch-per-step kernels on the same inputs -- every output buffer, ragged lengths."""
    ops, lib = _ops()
    monkeypatch.setenv('RE2E_LSTM_FWD2', fwd2)
    if uw is not None or fwd2 == '0':
        monkeypatch.setenv('RE2E_LSTM_BWD3', '0')        # the round-1..3 backward kernel ...
    if uw is not None:
        monkeypatch.setenv('RE2E_LSTM_BWD_UW', uw)       # ... with 8 (1) or 16 (2) hidden units per workgroup
    g = torch.Generator().manual_seed(B * 1000 + T)
    xg0 = [(torch.randn(T * B, 4 * H, generator=g) * 0.5).to(DEV) for _ in range(2)]
    whh = [(torch.randn(4 * H, H, generator=g) / H ** 0.5).to(DEV) for _ in range(2)]
    lens = torch.randint(1, T + 1, (B,), generator=g, dtype=torch.int32)
    lens[0] = T
    lens = lens.to(DEV)
    wsb = lib.query('re2e_lstm_workspace_bytes', B, H)
    outs = {}
    for mode in ('0', '1'):
        monkeypatch.setenv('RE2E_LSTM_PERSIST', mode)
        xg = [x.clone() for x in xg0]
        ws = torch.full((wsb // 4 + 16,), float('nan'), device=DEV)          # poisoned workspace: nothing may be read before it is written
        ybuf, cbuf = torch.zeros(T + 2, B, 2 * H, device=DEV), torch.zeros(T + 2, B, 2 * H, device=DEV)
        for _ in range(2):                                                    # twice: stale tags of the first call must not satisfy the second
            for d in range(2):
                xg[d].copy_(xg0[d])
            lib.call('re2e_lstm_seq_fwd', xg[0].data_ptr(), xg[1].data_ptr(), whh[0].data_ptr(), whh[1].data_ptr(), ybuf.data_ptr(), cbuf.data_ptr(),
                     lens.data_ptr(), T, B, H, ws.data_ptr(), wsb)
        torch.cuda.synchronize()
        outs[mode] = (xg[0], xg[1], ybuf, cbuf)
    for name, a, b in zip(('gates_f', 'gates_r', 'y', 'c'), outs['0'], outs['1']):
        assert torch.isfinite(b).all(), name
        close(name, b, a, tol=2e-6)
        if fwd2 == '1' and _fwd2_runs(B, H):             # the round-4 pair runs one instruction sequence: bit for bit
            assert torch.equal(a, b), name
    # backward: persistent (flagged write-through hand-off of the partial slabs) against launch-per-step, from the same forward state
    gf, gr, ybuf, cbuf = outs['0']
    dy = (torch.randn(T * B, 2 * H, generator=g) * 0.3).to(DEV)
    bouts = {}
    for mode in ('0', '1'):
        monkeypatch.setenv('RE2E_LSTM_PERSIST_BWD', mode)
        ws = torch.full((wsb // 4 + 16,), float('nan'), device=DEV)
        for _ in range(2):
            G = [gf.clone(), gr.clone()]
            dc = torch.full((B, 2 * H), float('nan'), device=DEV)
            lib.call('re2e_lstm_seq_bwd', G[0].data_ptr(), G[1].data_ptr(), whh[0].data_ptr(), whh[1].data_ptr(), dy.data_ptr(), ybuf.data_ptr(),
                     cbuf.data_ptr(), dc.data_ptr(), lens.data_ptr(), T, B, H, None, ws.data_ptr(), wsb)
        torch.cuda.synchronize()
        bouts[mode] = (G[0], G[1])
    for name, a, b in zip(('dgates_f', 'dgates_r'), bouts['0'], bouts['1']):
        assert torch.isfinite(b).all(), name
        close(name, b, a, tol=5e-6)
    assert lib.query('re2e_lstm_abort_count') == 0


@pytest.mark.parametrize('B,H', [(32, 256), (8, 256), (64, 512)])
def test_lstm_forward_nan_poisons_what_nn_lstm_poisons(B, H, monkeypatch):
    """The round-4 forward hands h(t) over with a tag in bit 30 of its fp32 pattern; a NaN travels as a reserved pattern.  A NaN
    pre-activation at (t0, b0) must poison utterance b0 from t0 on (both directions' later steps) and nothing else, exactly as in
    the launch-per-step kernels and nn.LSTM, and the recurrence must not give up on it."""
    ops, lib = _ops()
    T = 12
    g = torch.Generator().manual_seed(B + H)
    xg0 = [(torch.randn(T * B, 4 * H, generator=g) * 0.5).to(DEV) for _ in range(2)]
    t0, b0 = 4, min(5, B - 1)
    for d in range(2):
        xg0[d].view(T, B, 4 * H)[t0, b0, 7] = float('nan')
    whh = [(torch.randn(4 * H, H, generator=g) / H ** 0.5).to(DEV) for _ in range(2)]
    lens = torch.full((B,), T, dtype=torch.int32, device=DEV)
    wsb = lib.query('re2e_lstm_workspace_bytes', B, H)
    outs = {}
    for mode in ('0', '1'):
        monkeypatch.setenv('RE2E_LSTM_PERSIST', mode)
        xg = [x.clone() for x in xg0]
        ws = torch.zeros(wsb // 4 + 16, device=DEV)
        ybuf, cbuf = torch.zeros(T + 2, B, 2 * H, device=DEV), torch.zeros(T + 2, B, 2 * H, device=DEV)
        lib.call('re2e_lstm_seq_fwd', xg[0].data_ptr(), xg[1].data_ptr(), whh[0].data_ptr(), whh[1].data_ptr(), ybuf.data_ptr(), cbuf.data_ptr(),
                 lens.data_ptr(), T, B, H, ws.data_ptr(), wsb)
        torch.cuda.synchronize()
        outs[mode] = ybuf[1:T + 1].clone()
    y0, y1 = outs['0'], outs['1']
    assert torch.equal(torch.isnan(y0), torch.isnan(y1))
    nanmask = torch.isnan(y1)
    if _fwd2_runs(B, H):
        assert torch.equal(y0[~nanmask], y1[~nanmask])
    else:
        close('finite outputs', y1[~nanmask], y0[~nanmask], tol=2e-6)
    exp = torch.zeros_like(nanmask)
    exp[t0, b0, 7] = True                   # the step itself: only the unit whose pre-activation is NaN (forward half) ...
    exp[t0, b0, H + 7] = True               # ... and the same unit of the reverse half
    exp[t0 + 1:, b0, :H] = True             # forward direction: every later step of that utterance
    exp[:t0, b0, H:] = True                 # reverse direction runs T-1 .. 0: every earlier time index
    assert torch.equal(nanmask, exp)
    assert lib.query('re2e_lstm_abort_count') == 0


def test_lstm_persistent_handoffs_under_load():
    """The in-launch hand-offs of the persistent recurrences (tagged granules forward, flagged write-through slabs backward)
    beside an uneven filler load on another stream: every output word of repeated runs equals the launch-per-step result."""
    import importlib.util
    spec = importlib.util.spec_from_file_location('stress_lstm_persist', os.path.join(ROOT, 'tools', 'stress_lstm_persist.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    try:
        assert mod.run(60, 32, 256, 24) == 0
        assert mod.run(24, 64, 512, 24) == 0
    finally:
        os.environ.pop('RE2E_LSTM_PERSIST', None)
        os.environ.pop('RE2E_LSTM_PERSIST_BWD', None)


def test_ctc():
    ops, lib = _ops()
    T, B, V = 23, 4, 17
    logits = rnd(T, B, V, scale=2.0)
    hlens = [23, 20, 14, 9]
    labels = [[3, 3, 5, 1, 1, 1, 9], [2, 4], [7, 7], [16]]
    lr = logits.clone().requires_grad_(True)
    ref = F.ctc_loss(lr.log_softmax(2), torch.tensor(sum(labels, [])), torch.tensor(hlens), torch.tensor([len(l) for l in labels]),
                     blank=0, reduction='sum') / B
    (ref * 1.3).backward()
    lg = logits.to(DEV).requires_grad_(True)
    flat = torch.tensor(sum(labels, []), dtype=torch.int32, device=DEV)
    ll = [len(l) for l in labels]
    off = torch.tensor(np.concatenate([[0], np.cumsum(ll)[:-1]]), dtype=torch.int32, device=DEV)
    loss = ops.ctc_loss(lg, torch.tensor(hlens, dtype=torch.int32, device=DEV), flat, off, torch.tensor(ll, dtype=torch.int32, device=DEV), max(ll))
    (loss * 1.3).sum().backward()
    close('loss', loss.view(1), ref.view(1), tol=1e-5)
    close('dlogits', lg.grad, lr.grad, tol=1e-4)
    assert (lg.grad[20:, 1] == 0).all()


def test_ctc_projection_with_odd_vocabulary_uses_padded_gradient():
    """V % 4 != 0 and >= 2048 rows (config 4: V = 4233, 6400 rows): CtcFn.backward writes the gradient with rows padded to a multiple of
    16 and LinearFn.backward runs the projection's input / weight / bias gradients over the padded width -- against torch end to end."""
    ops, lib = _ops()
    T, B, V, E = 70, 32, 37, 24
    h = rnd(T, B, E, scale=1.0)
    W, b = rnd(V, E, seed=1, scale=0.3), rnd(V, seed=2, scale=0.1)
    g = torch.Generator().manual_seed(5)
    hlens = sorted((int(x) for x in torch.randint(40, T + 1, (B,), generator=g)), reverse=True)
    labels = [[int(x) for x in torch.randint(1, V, (int(n),), generator=g)] for n in torch.randint(1, 9, (B,), generator=g)]
    ll = [len(l) for l in labels]
    hr, Wr, br = h.clone().requires_grad_(True), W.clone().requires_grad_(True), b.clone().requires_grad_(True)
    ref = F.ctc_loss(F.linear(hr, Wr, br).log_softmax(2), torch.tensor(sum(labels, [])), torch.tensor(hlens), torch.tensor(ll), blank=0,
                     reduction='sum') / B
    (ref * 0.7).backward()
    hg = h.to(DEV).requires_grad_(True)
    Wg, bg = torch.nn.Parameter(W.to(DEV)), torch.nn.Parameter(b.to(DEV))
    flat = torch.tensor(sum(labels, []), dtype=torch.int32, device=DEV)
    off = torch.tensor(np.concatenate([[0], np.cumsum(ll)[:-1]]), dtype=torch.int32, device=DEV)
    assert not ops._PADDED_GRADS
    loss = ops.ctc_loss(ops.linear(hg, Wg, bg), torch.tensor(hlens, dtype=torch.int32, device=DEV), flat, off,
                        torch.tensor(ll, dtype=torch.int32, device=DEV), max(ll))
    (loss * 0.7).sum().backward()
    assert not ops._PADDED_GRADS                           # the padded buffer was claimed by LinearFn.backward
    close('loss', loss.view(1), ref.view(1), tol=1e-5)
    close('dh', hg.grad, hr.grad, tol=2e-4)
    close('dW', Wg.grad, Wr.grad, tol=2e-4)
    close('db', bg.grad, br.grad, tol=2e-4)


def test_decoder_loop_and_ce(golden_dir):
    import os
    ops, lib = _ops()
    from oracle import nets
    fx = dict(np.load(os.path.join(golden_dir, 'e2e_tiny.npz')))
    p = {k[2:]: torch.from_numpy(v).clone().requires_grad_(True) for k, v in fx.items() if k.startswith('p.') and not k.startswith('p.dec.att.')}
    hpad = torch.from_numpy(fx['hpad']).clone().requires_grad_(True)
    hlens = fx['hlens'].tolist()
    ys = nets.split_targets(torch.from_numpy(fx['targets']), fx['tlens'].tolist())
    V = p['dec.output.weight'].shape[0]
    loss_r, acc_r, att_r = nets.decoder_forward(p, hpad, hlens, ys, V - 1, return_att=True)
    loss_r.backward()

    from robust_e2e_gan_amd.model.e2e_decoder import decoder_forward_hip
    pg = {k: torch.nn.Parameter(v.detach().clone().to(DEV)) for k, v in p.items() if k.startswith('dec.') or k.startswith('att.')}
    hg = hpad.detach().clone().to(DEV).requires_grad_(True)
    loss, acc, att = decoder_forward_hip(pg, hg, hlens, ys, V - 1, return_att=True)
    loss.backward()
    close('att_w', att, att_r, tol=1e-4)
    close('loss_att', loss.view(1), loss_r.view(1), tol=1e-4)
    assert abs(float(acc) - acc_r) < 1e-6
    close('d_hpad', hg.grad, hpad.grad, tol=5e-4)
    for k, v in pg.items():
        ref = p[k].grad
        close(k, v.grad, ref, tol=1e-3, atol=1e-7)


def test_coral():
    ops, lib = _ops()
    from oracle import nets
    a, b = rnd(37, 20), rnd(41, 20, seed=1) * 1.3 + 0.2
    ar, br = a.clone().requires_grad_(True), b.clone().requires_grad_(True)
    lr = nets.coral(ar, br)
    (lr * 2.0).backward()
    ag, bg = a.to(DEV).requires_grad_(True), b.to(DEV).requires_grad_(True)
    l = ops.coral(ag, bg)
    (l * 2.0).backward()
    close('coral', l.view(1), lr.view(1), tol=1e-4)
    close('da', ag.grad, ar.grad, tol=1e-3)
    close('db', bg.grad, br.grad, tol=1e-3)


def test_optimizers():
    ops, lib = _ops()
    from robust_e2e_gan_amd.optim import FlatOptimizer
    torch.manual_seed(0)
    ref = torch.nn.Linear(33, 17)
    mine = torch.nn.Linear(33, 17)
    mine.load_state_dict(ref.state_dict())
    mine.to(DEV)
    for kind, kw in (('adadelta', dict(rho=0.95, eps=1e-8)), ('adam', dict(lr=0.005, betas=(0.5, 0.999)))):
        o_ref = torch.optim.Adadelta(ref.parameters(), **kw) if kind == 'adadelta' else torch.optim.Adam(ref.parameters(), **kw)
        o = FlatOptimizer(mine.parameters(), kind, **kw)
        for it in range(3):
            g = [rnd(*p.shape, seed=it) * 4 for p in ref.parameters()]
            for p, gi in zip(ref.parameters(), g):
                p.grad = gi.clone()
            o.zero_grad()
            for p, gi in zip(mine.parameters(), g):
                p.grad.copy_(gi.to(DEV))
            n_ref = torch.nn.utils.clip_grad_norm_(ref.parameters(), 5.0)
            n = o.clip_grad_norm(5.0)
            o_ref.step()
            o.step()
            assert abs(float(n) - float(n_ref)) < 1e-3 * float(n_ref)
        for p, q in zip(ref.parameters(), mine.parameters()):
            close(kind, q, p, tol=1e-4)


def test_misc_layout_kernels():
    ops, lib = _ops()
    x = rnd(4, 7, 13)
    close('transpose01', ops.transpose01(x.to(DEV)), x.transpose(0, 1), tol=0, atol=0)
    idx = torch.tensor([5, 0, 27, 13], dtype=torch.int32)
    close('gather', ops.gather_rows(x.view(28, 13).to(DEV), idx.to(DEV)), x.view(28, 13)[idx.long()], tol=0, atol=0)
    lens = [7, 3, 5, 1]
    close('mask_rows', ops.mask_rows(x.to(DEV), torch.tensor(lens, dtype=torch.int32, device=DEV)),
          x * (torch.arange(7).unsqueeze(0) < torch.tensor(lens).view(-1, 1)).unsqueeze(-1), tol=0, atol=0)
    from robust_e2e_gan_amd.data.mix_data_loader import pack_pad_device
    flat = torch.cat([x[i, :lens[i]] for i in range(4)])
    padded = pack_pad_device(flat.to(DEV), lens, 7)
    ref = torch.stack([torch.cat([x[i, :lens[i]], torch.zeros(7 - lens[i], 13)]) for i in range(4)])
    close('pack_pad', padded, ref, tol=0, atol=0)


def test_collate_device_matches_reference_output(golden_dir):
    """K1/F1: the device-side collate (one H2D copy per stream + pad kernel) against the OUTPUT of the reference's
    ``_collate_fn`` (tests/golden/make_fixtures_collate.py): bit-exact, ties in length and an empty target included."""
    import os
    import numpy as np
    from robust_e2e_gan_amd.data.mix_data_loader import collate_device
    fx = dict(np.load(os.path.join(golden_dir, 'collate_tiny.npz')))
    batch = []
    for i in range(int(fx['n'])):
        s5 = [torch.from_numpy(fx['s%d_%d' % (i, k)]) for k in range(5)]
        batch.append(('utt%d' % i, 'spk%d' % (i % 2), s5[0], s5[1], s5[2], s5[3], s5[4], fx['t%d' % i].tolist()))
    out = collate_device(batch, DEV, streams=(2, 3, 4, 5, 6))
    assert out[0] == ['utt%d' % i for i in fx['order']]
    for k in range(5):
        assert out[2 + k].is_cuda and np.array_equal(out[2 + k].cpu().numpy(), fx['expected'][k]), k
    assert np.array_equal(out[7].numpy(), fx['targets']) and np.array_equal(out[8].numpy(), fx['input_sizes'])
    assert np.array_equal(out[9].numpy(), fx['target_sizes'])


@pytest.mark.parametrize('M,N', [(1000, 64), (517, 2048), (300, 1200), (77, 4233), (5000, 8), (9, 128), (40000, 256)])
def test_colsum_and_fused_act_bwd(M, N):
    """re2e_colsum / re2e_act_bwd_colsum: vectorised tiles (N/4 a power of two, or N % 1024 == 0) and the scalar
    fall-back give the same column sums; the fused form also returns dz = dy * act'(y)."""
    ops, lib = _ops()
    dy, y = rnd(M, N, seed=5).to(DEV), rnd(M, N, seed=6).to(DEV)
    out = torch.full((N,), 0.5, device=DEV)
    ops.colsum_into(dy, M, N, out, 1.0)
    close('colsum', out, 0.5 + dy.double().sum(0).float(), tol=2e-5, atol=1e-4)
    for act, f in ((lib.ACT_RELU, lambda v: (v > 0).float()), (lib.ACT_TANH, lambda v: 1 - v * v),
                   (lib.ACT_LRELU, lambda v: torch.where(v > 0, torch.ones_like(v), torch.full_like(v, 0.2)))):
        b = torch.nn.Parameter(torch.zeros(N, device=DEV))
        b.grad = torch.full((N,), 0.25, device=DEV)
        dz, done = ops.act_bwd_bias(dy, y, act, b)
        assert done
        ref = dy * f(y)
        close('dz', dz, ref, tol=1e-6)
        close('db', b.grad, 0.25 + ref.double().sum(0).float(), tol=2e-5, atol=1e-4)


def test_ctc_prefix_score_device_vs_numpy():
    """re2e_ctc_prefix_score (all hypotheses x candidates of one position in one launch) against the numpy CTCPrefixScore that
    follows model/e2e_ctc.py:78-155, chained over four output positions: candidate labels in torch.topk order (ties -> lower
    label), prefix scores, combined local scores and the forward variables handed to the next position; repeated labels,
    the <eos> candidate and hypotheses of different lengths included."""
    import numpy as np
    ops, lib = _ops()
    from robust_e2e_gan_amd.model.beam_search import CTCPrefixScore, _topk
    rng = np.random.default_rng(3)
    T, V, cb, eos, w = 57, 23, 7, 22, 0.3
    lpz = torch.log_softmax(torch.from_numpy(rng.normal(size=(T, V)).astype(np.float32)) * 2.0, 1).numpy()
    ctc = CTCPrefixScore(lpz, 0, eos)
    lpz_d = torch.from_numpy(lpz).to(DEV)
    hyps = [{'yseq': [eos], 'state': ctc.initial_state(), 'score': np.float32(0.0)}]
    for pos in range(4):
        nh = len(hyps)
        att = torch.log_softmax(torch.from_numpy(rng.normal(size=(nh, V)).astype(np.float32)), 1).numpy()
        att[:, 5] = att[:, 9]                                          # a tie inside the candidate set of some rows
        if pos > 0:
            for k, hp in enumerate(hyps):
                att[k, hp['yseq'][-1]] = 0.5                           # the repeated label is always a candidate
            att[0, eos] = 0.9                                          # ... and so is <eos> for one hypothesis
        r_prev = torch.from_numpy(np.stack([hp['state'] for hp in hyps])).to(DEV)
        last = torch.tensor([hp['yseq'][-1] for hp in hyps], dtype=torch.int32, device=DEV)
        olen = torch.tensor([len(hp['yseq']) - 1 for hp in hyps], dtype=torch.int32, device=DEV)
        prev = torch.tensor([float(hp['score']) for hp in hyps], dtype=torch.float32, device=DEV)
        cand = torch.empty(nh, cb, dtype=torch.int32, device=DEV)
        out = torch.empty(2, nh, cb, device=DEV)
        r_new = torch.full((nh * cb, 2 * T), float('nan'), device=DEV)
        lib.call('re2e_ctc_prefix_score', lpz_d.data_ptr(), T, V, torch.from_numpy(att).to(DEV).data_ptr(), nh, r_prev.data_ptr(), last.data_ptr(),
                 olen.data_ptr(), prev.data_ptr(), cb, float(np.float32(1.0 - w)), float(np.float32(w)), 0, eos, cand.data_ptr(), out[0].data_ptr(),
                 out[1].data_ptr(), r_new.data_ptr())
        cand_h, out_h, r_h = cand.cpu().numpy(), out.cpu().numpy(), r_new.cpu().numpy().reshape(nh, cb, T, 2)
        nxt = []
        for k, hp in enumerate(hyps):
            _, want_c = _topk(att[k], cb)
            assert cand_h[k].tolist() == want_c.tolist(), (pos, k)
            sc, st = ctc(hp['yseq'], want_c, hp['state'])
            np.testing.assert_allclose(out_h[1, k], sc, rtol=2e-5, atol=2e-4)
            local = np.float32(1.0 - w) * att[k][want_c] + np.float32(w) * (sc - hp['score'])
            np.testing.assert_allclose(out_h[0, k], local, rtol=2e-5, atol=2e-4)
            first = max(len(hp['yseq']) - 1, 1) - 1                    # rows below are never read (numpy leaves them unset)
            np.testing.assert_allclose(r_h[k][:, first:], st[:, first:], rtol=2e-5, atol=2e-3)
            for j in (0, cb // 2, cb - 1):
                nxt.append({'yseq': hp['yseq'] + [int(want_c[j])], 'state': st[j].copy(), 'score': sc[j]})
        hyps = nxt[:5]
    # unsupported widths are refused, not approximated
    with pytest.raises(lib.Re2eError):
        lib.call('re2e_ctc_prefix_score', lpz_d.data_ptr(), T, V, lpz_d.data_ptr(), 1, lpz_d.data_ptr(), last.data_ptr(), olen.data_ptr(),
                 prev.data_ptr(), 65, 0.7, 0.3, 0, eos, cand.data_ptr(), out[0].data_ptr(), out[1].data_ptr(), r_new.data_ptr())


def test_ctc_prefix_score_all_labels_vs_numpy():
    """re2e_ctc_prefix_score_cands with ALL V labels as candidates in attention order (ctc_weight == 1.0 decoding, e2e_decoder.py:233-234)
    against the numpy CTCPrefixScore, chained over three positions (V = 300: two candidate blocks, the last one ragged)."""
    import numpy as np
    ops, lib = _ops()
    from robust_e2e_gan_amd.model.beam_search import CTCPrefixScore, _topk
    rng = np.random.default_rng(7)
    T, V, eos, w = 41, 300, 299, 1.0
    lpz = torch.log_softmax(torch.from_numpy(rng.normal(size=(T, V)).astype(np.float32)) * 2.0, 1).numpy()
    ctc = CTCPrefixScore(lpz, 0, eos)
    lpz_d = torch.from_numpy(lpz).to(DEV)
    hyps = [{'yseq': [eos], 'state': ctc.initial_state(), 'score': np.float32(0.0)}]
    for pos in range(3):
        nh = len(hyps)
        att = torch.log_softmax(torch.from_numpy(rng.normal(size=(nh, V)).astype(np.float32)), 1).numpy()
        att[:, 17] = att[:, 4]                                         # a tie: the lower label first
        att_d = torch.from_numpy(att).to(DEV)
        order = torch.sort(att_d, dim=1, descending=True, stable=True)[1].to(torch.int32).contiguous()
        r_prev = torch.from_numpy(np.stack([hp['state'] for hp in hyps])).to(DEV)
        last = torch.tensor([hp['yseq'][-1] for hp in hyps], dtype=torch.int32, device=DEV)
        olen = torch.tensor([len(hp['yseq']) - 1 for hp in hyps], dtype=torch.int32, device=DEV)
        prev = torch.tensor([float(hp['score']) for hp in hyps], dtype=torch.float32, device=DEV)
        out = torch.empty(2, nh, V, device=DEV)
        r_new = torch.full((nh * V, 2 * T), float('nan'), device=DEV)
        lib.call('re2e_ctc_prefix_score_cands', lpz_d.data_ptr(), T, V, att_d.data_ptr(), nh, r_prev.data_ptr(), last.data_ptr(), olen.data_ptr(),
                 prev.data_ptr(), order.data_ptr(), V, float(np.float32(1.0 - w)), float(np.float32(w)), 0, eos, out[0].data_ptr(), out[1].data_ptr(),
                 r_new.data_ptr())
        out_h, r_h = out.cpu().numpy(), r_new.cpu().numpy().reshape(nh, V, T, 2)
        nxt = []
        for k, hp in enumerate(hyps):
            _, want_c = _topk(att[k], V)
            assert order[k].cpu().numpy().tolist() == want_c.tolist(), (pos, k)
            sc, st = ctc(hp['yseq'], want_c, hp['state'])
            np.testing.assert_allclose(out_h[1, k], sc, rtol=2e-5, atol=2e-4)
            np.testing.assert_allclose(out_h[0, k], np.float32(w) * (sc - hp['score']), rtol=2e-5, atol=2e-4)
            first = max(len(hp['yseq']) - 1, 1) - 1
            np.testing.assert_allclose(r_h[k][:, first:], st[:, first:], rtol=2e-5, atol=2e-3)
            for j in (0, V // 2, V - 1):
                nxt.append({'yseq': hp['yseq'] + [int(want_c[j])], 'state': st[j].copy(), 'score': sc[j]})
        hyps = nxt[:4]


@pytest.mark.parametrize('B,E,D', [(32, 512, 300), (5, 20, 12), (32, 64, 8), (40, 128, 36)])
def test_dec_gates_cell_fused_vs_unfused(B, E, D):
    """re2e_dec_gates_cell_fwd (one launch) against the three launches it replaces (two skinny GEMMs with beta = 1 + the cell
    kernel) and against torch: ragged last unit group (D = 300: 4 units in workgroup 37), K tails (D = 300 is 37.5 k-groups),
    fewer than 32 and more than 32 utterances."""
    ops, lib = _ops()
    Dd = 16
    ldw = Dd + E
    cx, zp, cp = rnd(B, E, seed=1), rnd(B, D, seed=2), rnd(B, D, seed=3)
    w_ih, w_hh = rnd(4 * D, ldw, seed=4, scale=0.2), rnd(4 * D, D, seed=5, scale=0.2)
    g0 = rnd(B, 4 * D, seed=6)
    pre = g0 + cx @ w_ih[:, Dd:].t() + zp @ w_hh.t()
    gi, gf, gg, go = pre[:, :D].sigmoid(), pre[:, D:2 * D].sigmoid(), pre[:, 2 * D:3 * D].tanh(), pre[:, 3 * D:].sigmoid()
    c_ref = gf * cp + gi * gg
    h_ref = go * c_ref.tanh()
    d = lambda t: t.to(DEV).contiguous()
    cxd, zpd, cpd, wid, whd = d(cx), d(zp), d(cp), d(w_ih), d(w_hh)
    gates, c_out, h_out = d(g0), torch.empty(B, D, device=DEV), torch.empty(B, D, device=DEV)
    lib.call('re2e_dec_gates_cell_fwd', cxd.data_ptr(), zpd.data_ptr(), wid.data_ptr() + 4 * Dd, ldw, whd.data_ptr(), gates.data_ptr(), cpd.data_ptr(),
             c_out.data_ptr(), h_out.data_ptr(), B, E, D)
    close('c', c_out, c_ref, tol=2e-5)
    close('h', h_out, h_ref, tol=2e-5)
    close('gates', gates, torch.cat([gi, gf, gg, go], 1), tol=2e-5)
    if B <= 32:                                           # the unfused path (skinny GEMMs need M <= 32 to take the same kernels)
        g2, c2, h2 = d(g0), torch.empty(B, D, device=DEV), torch.empty(B, D, device=DEV)
        ops.gemm(cxd, wid.data_ptr() + 4 * Dd, g2, B, 4 * D, E, transb=True, ldb=ldw, beta=1.0, dev=torch.device(DEV))
        ops.gemm(zpd, whd, g2, B, 4 * D, D, transb=True, beta=1.0)
        lib.call('re2e_lstm_cell_fwd', g2.data_ptr(), cpd.data_ptr(), c2.data_ptr(), h2.data_ptr(), B, D)
        close('h vs unfused', h_out, h2.cpu(), tol=2e-5)


@pytest.mark.parametrize('B,T,L1,E,A,D,C,Fh', [(32, 200, 41, 512, 320, 300, 10, 100),     # config 4's decoder
                                               (8, 750, 12, 512, 320, 300, 10, 100),      # config 5: three frame chunks per utterance
                                               (32, 200, 5, 320, 320, 300, 10, 100),      # the reference's default widths
                                               (5, 37, 6, 20, 24, 12, 3, 4),              # ragged everything, partial slices
                                               (1, 5, 1, 16, 4, 4, 1, 0),                 # one utterance, one token, a 1-tap filter
                                               (2, 40, 3, 64, 64, 16, 12, 100),           # far more taps than frames, 12 channels
                                               (3, 300, 4, 132, 68, 36, 12, 7)])          # two chunks, E and A in different slice counts
def test_decoder_loop_persistent_vs_stepwise(B, T, L1, E, A, D, C, Fh):
    """csrc/decloop.hip (the whole teacher-forced loop in one persistent launch) against the launch-per-step sequence it replaces
    (re2e_attloc_fwd + re2e_dec_gates_cell_fwd per token): outputs and every tensor saved for the backward.  Same arithmetic, other
    summation orders (slice-partial energies, per-chunk context pieces) and a 1-ulp-rcp tanh in the energies."""
    ops, lib = _ops()
    Dd = D
    g = torch.Generator().manual_seed(B * 1000 + T)
    r = lambda *s, scale=1.0: (torch.randn(*s, generator=g) * scale).to(DEV)
    hl = torch.randint(max(1, T // 2), T + 1, (B,), generator=g)
    hl[0] = T
    hmask = r(B, T, E)
    for b in range(B):
        hmask[b, int(hl[b]):] = 0
    pre = r(B, T, A)
    Pm = dict(embed=r(50, Dd, scale=0.5), w_ih=r(4 * D, Dd + E, scale=0.08), w_hh=r(4 * D, D, scale=0.08), b_ih=r(4 * D, scale=0.1), b_hh=r(4 * D, scale=0.1),
              mlp_dec=r(A, D, scale=0.1), mlp_att=r(A, C, scale=0.5), loc_conv=r(C, 1, 1, 2 * Fh + 1, scale=0.3), gvec_w=r(1, A, scale=0.3), gvec_b=r(1, scale=0.1))
    ids = torch.randint(0, 50, (L1, B), generator=g).to(torch.int32).to(DEV)
    hlens = hl.to(torch.int32).to(DEV)
    if lib.query('re2e_dec_loop_workspace_bytes', L1, B, T, E, D, A, C, Fh) == 0:
        pytest.skip('shape outside the persistent loop on this device')

    class Ctx:
        def save_for_backward(self, *t): self.saved = t
        def mark_non_differentiable(self, *t): pass
    out = {}
    for name, flag in (('step', False), ('persist', True)):
        ops.DECODER_PERSIST = flag
        try:
            ctx = Ctx()
            zs, w = ops.DecoderLoopFn.forward(ctx, hmask, pre, ids, hlens, L1, Pm)
            torch.cuda.synchronize()
        finally:
            ops.DECODER_PERSIST = True
        out[name] = dict(z=zs.clone(), w=w.clone(), **{k: v.clone() for k, v in zip(('hmask', 'pre', 'emb', 'cx', 'zall', 'c', 'w2', 'gates', 'conv', 'dpj'), ctx.saved)})
    assert lib.query('re2e_lstm_abort_count') == 0
    for k in ('z', 'w', 'cx', 'c', 'gates', 'conv', 'dpj', 'zall'):            # first tokens: rounding only; whole loop: what 41 recurrent tokens make of it
        close(k + ' (first two tokens)', out['persist'][k][:2], out['step'][k][:2].cpu(), tol=2e-5)
        close(k, out['persist'][k], out['step'][k].cpu(), tol=2e-4)


@pytest.mark.parametrize('B,T,L1,E,A,D,C,Fh', [(32, 200, 41, 512, 320, 300, 10, 100),     # config 4's decoder
                                               (8, 750, 12, 512, 320, 300, 10, 100),      # config 5: 16 frame chunks per utterance
                                               (32, 200, 4, 320, 320, 300, 10, 100),      # the reference's default widths
                                               (5, 37, 6, 32, 24, 12, 3, 4),              # ragged everything
                                               (1, 5, 1, 16, 4, 4, 1, 0),                 # one utterance, one token, a 1-tap filter
                                               (2, 40, 3, 64, 64, 16, 12, 100),           # far more taps than frames, 12 channels
                                               (3, 24, 7, 512, 320, 300, 10, 100),        # one chunk per utterance, fewer frames than filter taps
                                               (3, 100, 4, 144, 68, 36, 12, 7)])
def test_decoder_loop_backward_persistent_vs_stepwise(B, T, L1, E, A, D, C, Fh):
    """csrc/decloop.hip's reverse loop (one persistent launch + one launch for d W_conv) against the launch-per-token backward: gradients
    of the encoder states, of pre and of every decoder / attention parameter, from the same forward."""
    ops, lib = _ops()
    g = torch.Generator().manual_seed(B * 1000 + T + 1)
    r = lambda *s, scale=1.0: (torch.randn(*s, generator=g) * scale).to(DEV)
    hl = torch.randint(max(1, T // 2), T + 1, (B,), generator=g)
    hl[0] = T
    hmask0 = r(B, T, E)
    for b in range(B):
        hmask0[b, int(hl[b]):] = 0
    pre0 = r(B, T, A)
    P0 = dict(embed=r(50, D, scale=0.5), w_ih=r(4 * D, D + E, scale=0.08), w_hh=r(4 * D, D, scale=0.08), b_ih=r(4 * D, scale=0.1), b_hh=r(4 * D, scale=0.1),
              mlp_dec=r(A, D, scale=0.1), mlp_att=r(A, C, scale=0.5), loc_conv=r(C, 1, 1, 2 * Fh + 1, scale=0.3), gvec_w=r(1, A, scale=0.3), gvec_b=r(1, scale=0.1))
    ids = torch.randint(0, 50, (L1, B), generator=g).to(torch.int32).to(DEV)
    hlens = hl.to(torch.int32).to(DEV)
    gz = r(L1, B, D)
    if lib.query('re2e_dec_loop_bwd_workspace_bytes', L1, B, T, E, D, A, C, Fh) == 0 or lib.query('re2e_dec_loop_workspace_bytes', L1, B, T, E, D, A, C, Fh) == 0:
        pytest.skip('shape outside the persistent loop on this device')
    out = {}
    for name, flag in (('step', False), ('persist', True)):
        hm, pr = hmask0.clone().requires_grad_(True), pre0.clone().requires_grad_(True)
        Pm = {k: torch.nn.Parameter(v.clone()) for k, v in P0.items()}
        ops.DECODER_PERSIST = True                      # the same (persistent) forward for both: what is compared is the backward
        zs, w = ops.DecoderLoopFn.apply(hm, pr, ids, hlens, L1, Pm)
        ops.DECODER_PERSIST = flag
        try:
            (zs * gz).sum().backward()
            torch.cuda.synchronize()
        finally:
            ops.DECODER_PERSIST = True
        out[name] = dict(d_enc=hm.grad.clone(), d_pre=pr.grad.clone(), **{k: v.grad.clone() for k, v in Pm.items() if v.grad is not None})
    assert lib.query('re2e_lstm_abort_count') == 0
    assert set(out['persist']) == set(out['step'])
    bad = []
    for k in sorted(out['step']):
        got, ref = out['persist'][k].float().cpu(), out['step'][k].float().cpu()
        err, scale = (got - ref).abs().max().item(), ref.abs().max().item()
        print('%-10s max err %.3e  scale %.3e' % (k, err, scale))
        # gvec_b: its true gradient is zero (softmax ignores a shift of the energies): both sides hold rounding noise of the sum of 260 000 terms
        lim = 3e-4 * scale + 1e-7 if k != 'gvec_b' else 1e-5 * float(L1 * B)
        if not (err <= lim):
            bad.append(k)
    assert not bad, bad


@pytest.mark.parametrize('M,K,N1,N2', [(32, 1200, 512, 300), (3, 56, 20, 14), (17, 40, 33, 1)])
def test_gemm_skinny2(M, K, N1, N2):
    """re2e_gemm_skinny2: two products sharing the skinny left operand in one launch (decoder backward: d ctx and d z from
    the same dgates), with leading dimensions larger than the widths."""
    ops, lib = _ops()
    A, B1, B2 = rnd(M, K, seed=1), rnd(K, N1 + 5, seed=2, scale=0.3), rnd(K, N2, seed=3, scale=0.3)
    Ad, B1d, B2d = A.to(DEV), B1.to(DEV), B2.to(DEV)
    C1, C2 = torch.full((M, N1), 7.0, device=DEV), torch.full((M, N2 + 2), 7.0, device=DEV)
    lib.call('re2e_gemm_skinny2', M, K, Ad.data_ptr(), K, B1d.data_ptr(), N1 + 5, N1, C1.data_ptr(), N1, B2d.data_ptr(), N2, N2, C2.data_ptr(), N2 + 2)
    close('C1', C1, A @ B1[:, :N1], tol=2e-5)
    close('C2', C2[:, :N2], A @ B2, tol=2e-5)
    assert (C2[:, N2:] == 7.0).all()                      # nothing written beyond the N2 columns
    with pytest.raises(lib.Re2eError):
        lib.call('re2e_gemm_skinny2', 33, K, Ad.data_ptr(), K, B1d.data_ptr(), N1 + 5, N1, C1.data_ptr(), N1, B2d.data_ptr(), N2, N2, C2.data_ptr(), N2 + 2)


def test_ctc_prefix_score_degenerate_rows_and_long_hypotheses():
    """Rows the top-k pre-selection cannot rank in the ordinary way must not fault (round-2 advisor finding): NaN attention scores
    are ranked as -inf, a row with fewer than ctc_beam entries above -inf still yields ctc_beam DISTINCT valid labels (lowest
    labels first among the ties, as torch.topk), and a hypothesis as long as the utterance has frames (out_len == T) keeps its
    state writes inside its own (T, 2) block."""
    import numpy as np
    ops, lib = _ops()
    rng = np.random.default_rng(11)
    T, V, cb, eos = 9, 12, 5, 11
    lpz = torch.log_softmax(torch.from_numpy(rng.normal(size=(T, V)).astype(np.float32)), 1)
    lpz_d = lpz.to(DEV)
    att = torch.full((3, V), float('-inf'))
    att[0, 7], att[0, 3] = -0.5, -1.0                 # two finite entries, ten at -inf
    att[1] = torch.log_softmax(torch.from_numpy(rng.normal(size=V).astype(np.float32)), 0)
    att[1, 4] = float('nan')                          # one NaN among finite scores
    att[2] = float('nan')                             # a diverged decoder row
    nh = 3
    r_prev = torch.full((nh, T, 2), -1e10).to(DEV)
    last = torch.tensor([1, 2, 3], dtype=torch.int32, device=DEV)
    olen = torch.tensor([1, T, T], dtype=torch.int32, device=DEV)             # two hypotheses as long as the utterance
    prev = torch.zeros(nh, device=DEV)
    cand = torch.full((nh, cb), -1, dtype=torch.int32, device=DEV)
    out = torch.empty(2, nh, cb, device=DEV)
    guard = torch.full((nh * cb + 1, 2 * T), 123.0, device=DEV)              # one spare state block behind the last candidate
    lib.call('re2e_ctc_prefix_score', lpz_d.data_ptr(), T, V, att.to(DEV).data_ptr(), nh, r_prev.data_ptr(), last.data_ptr(), olen.data_ptr(),
             prev.data_ptr(), cb, 0.7, 0.3, 0, eos, cand.data_ptr(), out[0].data_ptr(), out[1].data_ptr(), guard.data_ptr())
    torch.cuda.synchronize()
    c = cand.cpu().numpy()
    for k in range(nh):
        assert len(set(c[k].tolist())) == cb and c[k].min() >= 0 and c[k].max() < V, c[k]
    assert c[0].tolist() == [7, 3, 0, 1, 2]                                     # finite scores first, then the -inf ties by label
    want1 = att[1].clone()
    want1[4] = float('-inf')
    assert c[1].tolist() == torch.topk(want1, cb).indices.tolist()
    assert c[2].tolist() == [0, 1, 2, 3, 4]
    assert bool((guard[nh * cb] == 123.0).all()), 'state writes ran past the last candidate block'


@pytest.mark.parametrize('C,K', [(64, 64), (12, 24)])      # halo-patch kernel / general engine
def test_conv_relu_propagates_nan_like_torch(C, K):
    """ReLU epilogues follow torch.relu on non-finite pre-activations: NaN stays NaN (so that it reaches the grad-norm NaN gate of
    joint_train.py:189-193 whichever kernel the geometry selected), +inf stays +inf, -inf becomes 0."""
    ops, lib = _ops()
    N, H, W = 1, 18, 16
    x = rnd(N, H, W, C)
    x[0, 3, 3, 0] = float('nan')
    x[0, 12, 9, 1] = float('inf')
    Wt, b = rnd(K, C, 3, 3, seed=1, scale=0.1), rnd(K, seed=2)
    Wt[:, 1].abs_()                                   # +inf input x positive weights: +inf in some outputs, never inf - inf
    ref = F.relu(F.conv2d(x.permute(0, 3, 1, 2), Wt, b, padding=1)).permute(0, 2, 3, 1)
    y = ops.conv2d(x.to(DEV), torch.nn.Parameter(Wt.to(DEV)), torch.nn.Parameter(b.to(DEV)), 1, 1, 'relu').detach().cpu()
    assert bool(torch.isnan(ref).any()) and bool(torch.isinf(ref).any())
    if C % 8 == 0 and K % 64 == 0:
        # Winograd path (csrc/winograd.hip): the transforms mix the 4x4 input tile, so a non-finite input reaches every output of
        # the 2x2 blocks whose tiles contain it (a superset of torch's 3x3 neighbourhood) and +inf may arrive as NaN (inf - inf).
        # What the trainer needs holds: nothing non-finite is dropped (the NaN gate of joint_train.py:189-193 fires), and
        # outputs away from the poisoned pixels are the ordinary finite values.
        bad = ~torch.isfinite(ref)
        assert bool((~torch.isfinite(y))[bad].all())
        far = torch.ones(H, W, dtype=torch.bool)
        for (py, px) in ((3, 3), (12, 9)):
            far[max(py - 3, 0):py + 4, max(px - 3, 0):px + 4] = False
        assert bool(torch.isfinite(y[0][far]).all())
        assert (y[0][far] - ref[0][far]).abs().max() <= 2e-4 * ref[0][far].abs().max()
    else:
        assert torch.equal(torch.isnan(y), torch.isnan(ref))
        assert torch.equal(torch.isinf(y), torch.isinf(ref))
        fin = torch.isfinite(ref)
        assert (y[fin] - ref[fin]).abs().max() <= 2e-4 * ref[fin].abs().max()
    # -inf pre-activation through the general engine's Linear epilogue
    xl = rnd(40, 8)
    xl[5, 2] = float('-inf')
    Wl = rnd(6, 8, seed=3).abs_()
    yl = ops.linear(xl.to(DEV), torch.nn.Parameter(Wl.to(DEV)), None, 'relu').detach().cpu()
    refl = F.relu(xl @ Wl.t())
    assert torch.equal(yl[5], refl[5]) and float(refl[5].abs().sum()) == 0.0


@pytest.mark.parametrize('N,H,W,C,K', [(2, 16, 32, 8, 64), (1, 35, 80, 64, 64), (2, 33, 40, 16, 128), (1, 19, 7, 24, 64), (3, 8, 16, 128, 128),
                                       (1, 100, 40, 128, 64),
                                       # the LDS-staged input path (C % 64 == 0) at a width below one patch, with two and four passes of the
                                       # chunk loop, and with a single row
                                       (1, 19, 7, 64, 64), (1, 21, 24, 256, 64), (2, 16, 17, 128, 128), (1, 1, 40, 64, 64)])
def test_conv3x3_wino(N, H, W, C, K):
    """re2e_conv3x3_wino (fused Winograd F(2x2,3x3), csrc/winograd.hip) against F.conv2d: forward with bias + ReLU, forward fused with
    the 2x2 ceil-mode max pool (values and index bytes = re2e_maxpool2_fwd with relu_in), data gradient plain and through the ReLU
    mask of the layer in front; interior and border patches, odd heights / widths, both patch shapes (16x8 and 8x16 pixels), one
    and two 64-channel groups, 1..16 channel chunks."""
    ops, lib = _ops()
    x, Wt, b = rnd(N, H, W, C), rnd(K, C, 3, 3, seed=1, scale=1.0 / math.sqrt(9 * C)), rnd(K, seed=2)
    xc = x.permute(0, 3, 1, 2)
    ref = F.conv2d(xc, Wt, b, padding=1)
    xg, Wg, bg = x.to(DEV), Wt.to(DEV), b.to(DEV)
    y = ops.conv3x3_wino(xg, Wg, K, bias=bg, relu=True)
    close('fwd relu', y.permute(0, 3, 1, 2), F.relu(ref), tol=2e-4)
    y0 = ops.conv3x3_wino(xg, Wg, K)
    close('fwd plain', y0.permute(0, 3, 1, 2), ref - b.view(1, -1, 1, 1), tol=2e-4)
    # fused pool: same values and index bytes as the separate kernels on the Winograd output
    yp, idx = ops.conv3x3_wino(xg, Wg, K, bias=bg, relu=True, pool=True)
    want = F.max_pool2d(F.relu(ref), 2, 2, ceil_mode=True)
    close('pool', yp.permute(0, 3, 1, 2), want, tol=2e-4)
    yp2 = torch.empty_like(yp)
    idx2 = torch.empty_like(idx)
    lib.call('re2e_maxpool2_fwd', y.data_ptr(), N, H, W, K, yp2.data_ptr(), idx2.data_ptr(), 1)
    assert torch.equal(yp, yp2) and torch.equal(idx, idx2)
    # data gradient: dx = conv_transpose(dy, W), optionally through the ReLU of the layer in front (mask = that layer's output)
    dy = rnd(N, H, W, K, seed=4)
    dxr = torch.nn.grad.conv2d_input((N, C, H, W), Wt, dy.permute(0, 3, 1, 2).contiguous(), padding=1)
    dx = ops.conv3x3_wino(dy.to(DEV), Wg, C, dgrad=True) if K % 8 == 0 and C % 64 == 0 else None
    if dx is not None:
        close('dgrad', dx.permute(0, 3, 1, 2), dxr, tol=2e-4)
        mask = rnd(N, H, W, C, seed=6)
        dxm = ops.conv3x3_wino(dy.to(DEV), Wg, C, dgrad=True, mask=mask.to(DEV))
        close('dgrad masked', dxm.permute(0, 3, 1, 2), dxr * (mask.permute(0, 3, 1, 2) > 0), tol=2e-4)
    else:
        with pytest.raises(lib.Re2eError):
            ops.conv3x3_wino(dy.to(DEV), Wg, C, dgrad=True)


def test_conv3x3_wino_above_2gib_runs_in_image_slices():
    """A per-GPU batch of 64 gives the VGG layers 128+ images x 800 x 80 x 64 channels; 132 images = 2.16 GB per tensor, over 2^31 bytes: the fused Winograd kernels address
    with 31-bit offsets, so ops.conv3x3_wino / the weight gradient cut the IMAGE axis (round 4 declined and fell back two kernel
    generations).  Forward (+ fused pool), masked data gradient and weight gradient at that size against the same calls on the two halves
    (each under the limit: one launch) -- bitwise for the per-image results, 1e-5 for the weight gradient (a different split over patches)."""
    ops, lib = _ops()
    N, H, W, C, K = 132, 800, 80, 64, 64
    assert N * H * W * C * 4 >= 2 ** 31 and ops._wino_ok(N, H, W, C, K, (3, 3), 1, 1) and ops._wino_images(N, H, W, C, K) < N
    g = torch.Generator(device=DEV).manual_seed(3)
    x = torch.randn(N, H, W, C, device=DEV, generator=g)
    Wt = torch.randn(K, C, 3, 3, device=DEV, generator=g) / math.sqrt(9 * C)
    b = torch.randn(K, device=DEV, generator=g)
    h = N // 2
    y = ops.conv3x3_wino(x, Wt, K, bias=b, relu=True)
    for lo, hi in ((0, h), (h, N)):
        assert torch.equal(y[lo:hi], ops.conv3x3_wino(x[lo:hi].contiguous(), Wt, K, bias=b, relu=True))
    yp, idx = ops.conv3x3_wino(x, Wt, K, bias=b, relu=True, pool=True)
    yp2, idx2 = ops.conv3x3_wino(x[h:].contiguous(), Wt, K, bias=b, relu=True, pool=True)
    assert torch.equal(yp[h:], yp2) and torch.equal(idx[h:], idx2)
    del yp, idx, yp2, idx2
    dx = ops.conv3x3_wino(y, Wt, C, dgrad=True, mask=x)                # (y as a stand-in for dy: same shape)
    assert torch.equal(dx[:h], ops.conv3x3_wino(y[:h].contiguous(), Wt, C, dgrad=True, mask=x[:h].contiguous()))
    del dx
    # weight gradient: the sliced sum against the two halves accumulated by hand
    def wgrad(xs, dys, gw, beta):
        n = xs.shape[0]
        wsb = lib.query('re2e_conv3x3_wino_wgrad_workspace_bytes', n, H, W, C, K)
        ws = torch.empty(wsb // 4 + 16, device=DEV)
        lib.call('re2e_conv3x3_wino_wgrad', xs.data_ptr(), n, H, W, C, dys.data_ptr(), K, gw.data_ptr(), beta, ws.data_ptr(), wsb)
    want = torch.zeros(K, C, 3, 3, device=DEV)
    wgrad(x[:h].contiguous(), y[:h].contiguous(), want, 0.0)
    wgrad(x[h:].contiguous(), y[h:].contiguous(), want, 1.0)
    xr, Wp = x.requires_grad_(False), torch.nn.Parameter(Wt.clone())
    out = ops.conv2d(xr, Wp, None, 1, 1, None)
    assert out.shape == (N, H, W, K)
    out.backward(y)
    torch.cuda.synchronize()
    close('wgrad over image slices', Wp.grad, want.cpu(), tol=1e-5)


def test_step_gate_kernel():
    """re2e_step_gate (csrc/lstm.hip): the device half of the trainer's NaN gate and of the give-up protocol, case by case -- nothing refused when
    nothing happened; a non-finite ENHANCER sum of squares refuses the main update and makes the reported norm NaN; a give-up (counted by the
    test hook re2e_debug_force_abort through one forced sequence) refuses main and D, writes NaN as the next step's loss factor and reports
    delta; the data-parallel form refuses D from the main gate's flag; an acknowledgement puts everything back."""
    ops, lib = _ops()
    i32 = lambda v: torch.tensor([v], dtype=torch.int32, device=DEV)
    f = lambda *v: torch.tensor(list(v), dtype=torch.float32, device=DEV)
    base, hold, delta = i32(0), f(7.0), f(7.0)
    gate = lambda ack, es, sm, sd, dr, h, d: lib.call('re2e_step_gate', base.data_ptr(), ack, ptr_(es), ptr_(sm), ptr_(sd), ptr_(dr), ptr_(h), ptr_(d))
    ptr_ = lambda t: None if t is None else t.data_ptr()
    gate(1, None, None, None, None, hold, delta)                       # acknowledge whatever earlier tests left
    assert hold.item() == 1.0 and delta.item() == 0.0
    sm, sd = f(3.0, 0.5, 1.0, 3.0, 1.0, 1.0), f(2.0, 1.0, 1.0, 2.0, 1.0, 1.0)
    gate(0, f(12.5), sm, sd, None, hold, delta)
    assert sm.tolist() == [3.0, 0.5, 1.0, 3.0, 1.0, 1.0] and sd.tolist()[2] == 1.0 and hold.item() == 1.0
    gate(0, f(float('nan')), sm, sd, None, hold, delta)                # the enhancer's gradients are not finite: main refused, D untouched
    got = sm.tolist()
    assert got[2] == 0.0 and got[5] == 0.0 and got[0] != got[0] and got[3] != got[3] and sd.tolist()[2] == 1.0 and hold.item() == 1.0
    sm = f(3.0, 0.5, 1.0, 3.0, 1.0, 1.0)
    gate(0, f(float('inf')), sm, None, None, None, None)
    assert sm.tolist()[2] == 0.0
    # data-parallel form: D follows the main gate's flag
    sm0, sd = f(3.0, 0.5, 0.0, 3.0, 1.0, 0.0), f(2.0, 1.0, 1.0, 2.0, 1.0, 1.0)
    lib.call('re2e_step_gate', base.data_ptr(), 0, None, None, sd.data_ptr(), sm0.data_ptr() + 8, None, None)
    assert sd.tolist()[2] == 0.0
    # a give-up: one forced persistent forward sequence raises the counter
    T, B, H = 4, 2, 16
    wsb = lib.query('re2e_lstm_workspace_bytes', B, H)
    ws = torch.empty(wsb // 4 + 16, device=DEV)
    xg = [torch.zeros(T * B, 4 * H, device=DEV) for _ in range(2)]
    whh = [torch.zeros(4 * H, H, device=DEV) for _ in range(2)]
    ybuf, cbuf = torch.zeros(T + 2, B, 2 * H, device=DEV), torch.zeros(T + 2, B, 2 * H, device=DEV)
    lens = torch.full((B,), T, dtype=torch.int32, device=DEV)
    lib.query('re2e_debug_force_abort', 1)
    try:
        lib.call('re2e_lstm_seq_fwd', xg[0].data_ptr(), xg[1].data_ptr(), whh[0].data_ptr(), whh[1].data_ptr(), ybuf.data_ptr(), cbuf.data_ptr(),
                 lens.data_ptr(), T, B, H, ws.data_ptr(), wsb)
    finally:
        lib.query('re2e_debug_force_abort', 0)
    sm, sd = f(3.0, 0.5, 1.0, 3.0, 1.0, 1.0), f(2.0, 1.0, 1.0, 2.0, 1.0, 1.0)
    gate(0, f(1.0), sm, sd, None, hold, delta)
    assert delta.item() == 1.0 and hold.item() != hold.item() and sm.tolist()[2] == 0.0 and sd.tolist()[2] == 0.0
    gate(1, None, None, None, None, hold, delta)                       # acknowledged: the next step is whole again
    assert delta.item() == 0.0 and hold.item() == 1.0


def test_device_prefetcher_equals_host_collate():
    """data.prefetch.DevicePrefetcher (pinned staging slots reused every third batch, H2D + re2e_pack_pad on a copy stream, event
    hand-off) against the host collate the reference defines (data/mix_data_loader.py:264-302), bit for bit, over more batches
    than there are staging slots and with different batch shapes in turn -- while the consumer stream is kept busy, so that the
    copy stream really runs ahead."""
    from robust_e2e_gan_amd.data.mix_data_loader import _collate_fn
    from robust_e2e_gan_amd.data.prefetch import DevicePrefetcher
    g = torch.Generator().manual_seed(5)

    def sample(i, T, F_=33, L=4):
        mk = lambda: torch.randn(T, F_, generator=g)
        return ('u%d' % i, 's', mk(), mk(), mk(), mk(), mk(), torch.randint(1, 9, (L,), generator=g))
    batches = []
    for k in range(8):
        lens = [int(v) for v in torch.randint(5, 60 + 30 * (k % 3), (3 + k % 4,), generator=g)]
        batches.append([sample(i, T) for i, T in enumerate(lens)])
    busy = torch.randn(2048, 2048, device=DEV)
    n = 0
    for dev_b, host_list in zip(DevicePrefetcher(batches, DEV), batches):
        for _ in range(3):
            busy = busy @ busy * 1e-3                      # the consumer stream has work queued: the prefetcher is ahead of it
        ref = _collate_fn(list(host_list))
        assert dev_b[0] == ref[0]
        for k in (2, 4, 5):
            assert torch.equal(dev_b[k].cpu(), ref[k]), (n, k)
        assert dev_b[3] is None and dev_b[6] is None
        assert torch.equal(dev_b[7], ref[7]) and torch.equal(dev_b[8], ref[8]) and torch.equal(dev_b[9], ref[9])
        n += 1
    assert n == len(batches)
