#!/usr/bin/env python3
"""Thin-channel convolutions at the config-4 shapes: direct kernels (thinconv.hip) vs the implicit GEMM
(RE2E_NO_THIN=1).  Prints microseconds per call and the effective HBM rate over the wide tensor."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from robust_e2e_gan_amd import lib, ops
from robust_e2e_gan_amd.lib import call, query
from tools.bench_gemm import timeit

DEV = 'cuda:0'


def case(name, N, H, W, C, K, k, s, p):
    x = torch.randn(N, H, W, C, device=DEV)
    wt = torch.randn(K, C, k, k, device=DEV) * 0.05
    OH, OW = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
    wg = torch.empty(K, k, k, C, device=DEV)
    call('re2e_conv_weight_gather', wt.data_ptr(), wg.data_ptr(), K, C, k, k, 0, k, k, 0, 0, 1)
    y = torch.empty(N, OH, OW, K, device=DEV)
    dy = torch.randn(N, OH, OW, K, device=DEV)
    dw = torch.empty_like(wt)
    wide = 4.0 * max(x.numel(), y.numel())
    t = timeit(lambda: call('re2e_conv_igemm', x.data_ptr(), N, H, W, C, wg.data_ptr(), K, k, k, OH, OW, s, s, 1, 1, -p, -p, y.data_ptr(), OH, OW,
                            1, 1, 0, 0, None, lib.ACT_RELU, 0.0))
    wsb = query('re2e_conv_wgrad_workspace_bytes', N, OH, OW, C, K, k, k)
    ws = torch.empty(wsb // 4 + 16, device=DEV)
    tw = timeit(lambda: call('re2e_conv_wgrad', x.data_ptr(), N, H, W, C, dy.data_ptr(), K, k, k, OH, OW, s, s, -p, -p, dw.data_ptr(), 0.0,
                             ws.data_ptr(), wsb))
    td = timeit(lambda: ops.conv_dgrad(dy, wt, (N, H, W, C), s, p))
    print('%-34s fwd %7.1f us (%4.2f TB/s)  wgrad %7.1f us (%4.2f TB/s)  dgrad %7.1f us (%4.2f TB/s)' % (
        name, t * 1e6, wide / t / 1e12, tw * 1e6, wide / tw / 1e12, td * 1e6, wide / td / 1e12), flush=True)


if __name__ == '__main__':
    print('thin kernels', 'off' if os.environ.get('RE2E_NO_THIN') else 'on')
    case('vgg conv1_1 64x800x80 1->64', 64, 800, 80, 1, 64, 3, 1, 1)
    case('D conv1 32x800x80 1->64 k4 s2', 32, 800, 80, 1, 64, 4, 2, 1)
    case('D conv5 32x99x9 512->1 k4 s1', 32, 99, 9, 512, 1, 4, 1, 1)
