#!/usr/bin/env python3
"""Winograd F(2x2,3x3) against the direct halo-patch kernel at the VGG shapes of config 4 (2B = 64 images), alone on the chip:
average launch time over 20 launches and direct-equivalent TFLOP/s (2*9*C*K per output pixel)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from robust_e2e_gan_amd import lib, ops


def t(fn, it=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(it):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it


def main():
    dev = 'cuda:0'
    for name, N, H, W, C, K in (('conv1_2', 64, 800, 80, 64, 64), ('conv2_1', 64, 400, 40, 64, 128), ('conv2_2', 64, 400, 40, 128, 128),
                                ('conv2_1 dgrad', 64, 400, 40, 128, 64), ('conv1_2 B=32', 32, 800, 80, 64, 64)):
        x = torch.randn(N, H, W, C, device=dev)
        Wt = torch.randn(K, C, 3, 3, device=dev) * 0.04
        b = torch.zeros(K, device=dev)
        wg = torch.empty(K, 3, 3, C, device=dev)
        lib.call('re2e_conv_weight_gather', Wt.data_ptr(), wg.data_ptr(), K, C, 3, 3, 0, 3, 3, 0, 0, 1)
        y = torch.empty(N, H, W, K, device=dev)
        fl = 2.0 * 9 * C * K * N * H * W
        d = t(lambda: lib.call('re2e_conv_igemm', x.data_ptr(), N, H, W, C, wg.data_ptr(), K, 3, 3, H, W, 1, 1, 1, 1, -1, -1, y.data_ptr(), H, W, 1, 1,
                               0, 0, b.data_ptr(), lib.ACT_RELU, 0.0))
        w = t(lambda: ops.conv3x3_wino(x, Wt, K, bias=b, relu=True))
        wp = t(lambda: ops.conv3x3_wino(x, Wt, K, bias=b, relu=True, pool=True))
        yw = ops.conv3x3_wino(x, Wt, K, bias=b, relu=True)
        lib.call('re2e_conv_igemm', x.data_ptr(), N, H, W, C, wg.data_ptr(), K, 3, 3, H, W, 1, 1, 1, 1, -1, -1, y.data_ptr(), H, W, 1, 1, 0, 0,
                 b.data_ptr(), lib.ACT_RELU, 0.0)
        err = float((yw - y).abs().max() / y.abs().max())
        print('%-14s direct %.3f ms (%.1f TF/s)   winograd %.3f ms (%.1f TF/s direct-equivalent, %.1f executed)   +pool %.3f ms   max rel diff %.1e'
              % (name, d, fl / d / 1e9, w, fl / w / 1e9, fl / 2.25 / w / 1e9, wp, err), flush=True)


if __name__ == '__main__':
    main()
