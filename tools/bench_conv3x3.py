#!/usr/bin/env python3
"""VGG 3x3 convolutions (forward + data gradient) of the config-4 step: halo-patch kernel (conv3x3.hip) against the general
implicit-GEMM engine (RE2E_NO_HALO=1), HIP-event timed on the launch stream.  The switch is read once per process, so the
tool runs itself twice."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
SHAPES = [  # name, N, H, W, Cin, Cout
    ('conv1_2', 32, 800, 80, 64, 64), ('conv1_2 2B', 64, 800, 80, 64, 64), ('conv2_1', 32, 400, 40, 64, 128), ('conv2_2', 32, 400, 40, 128, 128),
    ('conv2_1 dgrad-shape', 32, 400, 40, 128, 64),
]


def run():
    import torch
    from robust_e2e_gan_amd import lib
    dev = torch.device('cuda:0')
    for name, N, H, W, C, K in SHAPES:
        x = torch.randn(N, H, W, C, device=dev)
        wg = torch.randn(K, 3, 3, C, device=dev) * 0.04
        b = torch.zeros(K, device=dev)
        y = torch.empty(N, H, W, K, device=dev)
        for d, o in ((1, -1), (-1, 1)):
            args = (x.data_ptr(), N, H, W, C, wg.data_ptr(), K, 3, 3, H, W, 1, 1, d, d, o, o, y.data_ptr(), H, W, 1, 1, 0, 0, b.data_ptr(),
                    lib.ACT_RELU if d > 0 else lib.ACT_NONE, 0.0)
            for _ in range(3):
                lib.call('re2e_conv_igemm', *args)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
            for _ in range(20):
                lib.call('re2e_conv_igemm', *args)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 20
            fl = 2.0 * 9 * C * K * N * H * W
            print('%-22s %s N=%d %dx%d %d->%d : %7.3f ms  %6.1f TFLOP/s  (%.3f of 157.3)' % (name, 'fwd  ' if d > 0 else 'dgrad', N, H, W, C, K, ms,
                                                                                          fl / ms / 1e9, fl / ms / 1e9 / 157.3), flush=True)


if __name__ == '__main__':
    if os.environ.get('BENCH_CONV_CHILD'):
        run()
    else:
        for tag, env in (('halo-patch kernel', {}), ('general engine (RE2E_NO_HALO=1)', {'RE2E_NO_HALO': '1'})):
            print('== %s' % tag, flush=True)
            e = dict(os.environ, BENCH_CONV_CHILD='1', **env)
            subprocess.call([sys.executable, os.path.abspath(__file__)], env=e)
