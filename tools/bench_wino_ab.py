#!/usr/bin/env python3
"""The step's fused Winograd F(2x2,3x3) launches (config 4: 2B = 64 images), alone on the chip, for A/B runs of two library builds in one GPU
session (RE2E_LIB=... python tools/bench_wino_ab.py [tag]): average / minimum launch time over 20 launches, executed TFLOP/s
(2 * 16 / 4 * C * K per output pixel), and a checksum of each result so that two builds can be compared bit for bit.

  forward + ReLU (+ 2x2 max pool)  conv1_2 64->64 @800x80 (pool), conv2_1 64->128 @400x40, conv2_2 128->128 @400x40 (pool)
  data gradient (+ ReLU mask)      conv2_2, conv2_1 (128->64, mask), conv1_2 (mask)
  weight gradient                  conv1_2, conv2_1, conv2_2"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from robust_e2e_gan_amd import lib, ops


def t(fn, it=20):
    for _ in range(3):
        fn()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(it + 1)]
    torch.cuda.synchronize()
    ev[0].record()
    for i in range(it):
        fn()
        ev[i + 1].record()
    torch.cuda.synchronize()
    d = [ev[i].elapsed_time(ev[i + 1]) for i in range(it)]
    return sum(d) / it, min(d)


def digest(*ts):
    return ' '.join('%.9e' % float(x.double().abs().sum()) for x in ts)


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else os.path.basename(os.environ.get('RE2E_LIB', 'shipped'))
    dev = 'cuda:0'
    g = torch.Generator(device=dev).manual_seed(5)
    rows = []
    for name, N, H, W, C, K, pool, dgrad, mask in (
            ('conv1_2 fwd+pool', 64, 800, 80, 64, 64, True, False, False), ('conv2_1 fwd', 64, 400, 40, 64, 128, False, False, False),
            ('conv2_2 fwd+pool', 64, 400, 40, 128, 128, True, False, False), ('conv2_2 dgrad', 64, 400, 40, 128, 128, False, True, True),
            ('conv2_1 dgrad', 64, 400, 40, 128, 64, False, True, True), ('conv1_2 dgrad', 64, 800, 80, 64, 64, False, True, True)):
        x = torch.randn(N, H, W, C, device=dev, generator=g)
        Wt = torch.randn((C, K, 3, 3) if dgrad else (K, C, 3, 3), device=dev, generator=g) * 0.04
        b = None if dgrad else torch.randn(K, device=dev, generator=g)
        m = torch.randn(N, H, W, K, device=dev, generator=g) if mask else None
        fn = lambda: ops.conv3x3_wino(x, Wt, K, dgrad=dgrad, bias=b, relu=not dgrad, mask=m, pool=pool)
        avg, mn = t(fn)
        out = fn()
        fl = 2.0 * 4 * C * K * N * H * W
        rows.append((name, avg, mn, fl / avg / 1e9, digest(*(out if pool else (out,)))))
        del x, m, out
    for name, N, H, W, C, K in (('conv1_2 wgrad', 64, 800, 80, 64, 64), ('conv2_1 wgrad', 64, 400, 40, 64, 128), ('conv2_2 wgrad', 64, 400, 40, 128, 128)):
        x = torch.randn(N, H, W, C, device=dev, generator=g)
        dy = torch.randn(N, H, W, K, device=dev, generator=g)
        gw = torch.zeros(K, C, 3, 3, device=dev)
        wsb = lib.query('re2e_conv3x3_wino_wgrad_workspace_bytes', N, H, W, C, K)
        ws = ops.workspace(wsb, x.device, 'winow')

        def fn():
            lib.call('re2e_conv3x3_wino_wgrad', x.data_ptr(), N, H, W, C, dy.data_ptr(), K, gw.data_ptr(), 0.0, ws.data_ptr(), wsb)
            return gw
        avg, mn = t(fn)
        out = fn()
        fl = 2.0 * 4 * C * K * N * H * W
        rows.append((name, avg, mn, fl / avg / 1e9, digest(out)))
        del x, dy, out
    for name, avg, mn, tf, dg in rows:
        print('%-10s %-18s avg %.4f ms  min %.4f ms  %6.1f TFLOP/s executed (%.3f of 157.3)   sum|.| %s' % (tag, name, avg, mn, tf, tf / 157.3, dg), flush=True)
    print('%-10s total avg %.4f ms' % (tag, sum(r[1] for r in rows)))


if __name__ == '__main__':
    main()
