#!/usr/bin/env python3
"""Micro-benchmark of the MFMA engine at the config-4 shapes (TFLOP/s, HIP events on the launch stream)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from robust_e2e_gan_amd import lib, ops
from robust_e2e_gan_amd.lib import call, query

DEV = 'cuda:0'


def timeit(fn, iters=10):
    for _ in range(2):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / iters


def conv_case(N, H, W, C, K, k, s, p):
    x = torch.randn(N, H, W, C, device=DEV)
    wt = torch.randn(K, C, k, k, device=DEV) * 0.05
    OH, OW = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
    wg = torch.empty(K, k, k, C, device=DEV)
    call('re2e_conv_weight_gather', wt.data_ptr(), wg.data_ptr(), K, C, k, k, 0, k, k, 0, 0, 1)
    y = torch.empty(N, OH, OW, K, device=DEV)
    dy = torch.randn(N, OH, OW, K, device=DEV)
    dw = torch.empty_like(wt)
    fl = 2.0 * k * k * C * K * N * OH * OW
    t = timeit(lambda: call('re2e_conv_igemm', x.data_ptr(), N, H, W, C, wg.data_ptr(), K, k, k, OH, OW, s, s, 1, 1, -p, -p, y.data_ptr(), OH, OW,
                            1, 1, 0, 0, None, lib.ACT_RELU, 0.0))
    wsb = query('re2e_conv_wgrad_workspace_bytes', N, OH, OW, C, K, k, k)
    ws = torch.empty(wsb // 4 + 16, device=DEV)
    tw = timeit(lambda: call('re2e_conv_wgrad', x.data_ptr(), N, H, W, C, dy.data_ptr(), K, k, k, OH, OW, s, s, -p, -p, dw.data_ptr(), 0.0,
                             ws.data_ptr(), wsb))
    td = timeit(lambda: ops.conv_dgrad(dy, wt, (N, H, W, C), s, p))
    return fl / t / 1e12, fl / tw / 1e12, fl / td / 1e12


def gemm_case(M, N, K):
    A, B, Bt, C = torch.randn(M, K, device=DEV), torch.randn(N, K, device=DEV), torch.randn(K, N, device=DEV), torch.empty(M, N, device=DEV)
    At, Cw = torch.randn(K, M, device=DEV), torch.empty(M, N, device=DEV)
    fl = 2.0 * M * N * K
    nt = timeit(lambda: ops.gemm(A, B, C, M, N, K, transb=True))
    nn = timeit(lambda: ops.gemm(A, Bt, C, M, N, K))
    tn = timeit(lambda: ops.gemm(At, Bt, Cw, M, N, K, transa=True))
    return fl / nt / 1e12, fl / nn / 1e12, fl / tn / 1e12


if __name__ == '__main__':
    print('variant', os.environ.get('RE2E_IGEMM_VARIANT', '0'))
    for name, cfg in (('vgg conv1_2 64x800x80 64->64', (64, 800, 80, 64, 64, 3, 1, 1)), ('vgg conv2_1 64x400x40 64->128', (64, 400, 40, 64, 128, 3, 1, 1)),
                      ('vgg conv2_2 64x400x40 128->128', (64, 400, 40, 128, 128, 3, 1, 1)), ('D conv2 32x400x40 64->128 s2', (32, 400, 40, 64, 128, 4, 2, 1)),
                      ('D conv4 32x100x10 256->512 s1', (32, 100, 10, 256, 512, 4, 1, 1))):
        f, w, d = conv_case(*cfg)
        print('%-34s fwd %6.1f  wgrad %6.1f  dgrad %6.1f TFLOP/s' % (name, f, w, d), flush=True)
    for name, cfg in (('enc L0 xproj 12800x2048x2560', (12800, 2048, 2560)), ('enh L1 xproj 25600x1024x512', (25600, 1024, 512)),
                      ('ctc_lo 6400x4233x512', (6400, 4233, 512)), ('dW_ih 2048x2560x12800 (as MxNxK)', (2048, 2560, 12800))):
        a, b, c = gemm_case(*cfg)
        print('%-34s NT %6.1f  NN %6.1f  TN %6.1f TFLOP/s' % (name, a, b, c), flush=True)
