"""dy^T x (re2e_gemm, transa) on the step's weight-gradient shapes, alone on the chip: us per call and TFLOP/s.  For same-session A/B runs of two
library builds:  RE2E_EXPERIMENTS=1 RE2E_LIB=<other .so> python tools/bench_tn.py tag"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from robust_e2e_gan_amd import ops

DEV = 'cuda:0'
SHAPES = [(2048, 512, 12800), (2048, 2560, 12800), (1024, 256, 25600), (1024, 512, 25600), (512, 1024, 12800), (1024, 260, 25600), (4240, 512, 6400),
          (1200, 300, 1312)]


def timed(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


tag = sys.argv[1] if len(sys.argv) > 1 else 'shipped'
tot = 0.0
for M, N, K in SHAPES:
    a, b = torch.randn(K, M, device=DEV), torch.randn(K, N, device=DEV)
    c = torch.empty(M, N, device=DEV)
    t = timed(lambda: ops.gemm(a, b, c, M, N, K, transa=True))
    tot += t
    print('%-8s %-22s %9.1f us %7.1f TFLOP/s  sum|.| %.9e' % (tag, '%dx%dx%d' % (M, N, K), t, 2.0 * M * N * K / t / 1e6, float(c.double().abs().sum())), flush=True)
print('%-8s total %.1f us' % (tag, tot))
