#!/usr/bin/env python3
"""dy^T x (igemm.hip, split-K) with the K slices mapped to XCDs (RE2E_TN_XCD_KSLICE unset) against tiles that keep their XCD for all slices
(RE2E_TN_XCD_KSLICE=0), same session, experiments build.  `--once MODE` runs each shape 5 times under one mode (for a rocprofv3 --pmc FETCH_SIZE pass)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from robust_e2e_gan_amd import ops

DEV = 'cuda:0'
SHAPES = [(2048, 512, 12800), (2048, 2560, 12800), (1024, 256, 25600), (1024, 512, 25600), (512, 1024, 12800)]


def timeit(fn, iters):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / iters


def main():
    once = sys.argv[2] if len(sys.argv) > 2 and sys.argv[1] == '--once' else None
    for (M, N, K) in SHAPES:
        A, B = torch.randn(K, M, device=DEV), torch.randn(K, N, device=DEV)
        C = [torch.empty(M, N, device=DEV) for _ in range(2)]
        fl = 2.0 * M * N * K

        def run(mode, out):
            if mode == '0':
                os.environ['RE2E_TN_XCD_KSLICE'] = '0'
            else:
                os.environ.pop('RE2E_TN_XCD_KSLICE', None)
            ops.gemm(A, B, out, M, N, K, transa=True)
        if once is not None:
            for _ in range(5):
                run(once, C[0])
            torch.cuda.synchronize()
            continue
        best = [1e9, 1e9]
        for i, m in enumerate(('0', '1')):
            run(m, C[i])
        for _ in range(4):
            for i, m in enumerate(('0', '1')):
                run(m, C[i])
                best[i] = min(best[i], timeit(lambda: run(m, C[i]), 20))
        print('%5dx%5dx%6d  tiles-on-XCD %6.1f   slices-on-XCD %6.1f TFLOP/s   bitwise equal: %s' % (M, N, K, fl / best[0] / 1e12, fl / best[1] / 1e12, torch.equal(C[0], C[1])), flush=True)


if __name__ == '__main__':
    main()
