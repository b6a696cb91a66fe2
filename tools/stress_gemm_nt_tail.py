#!/usr/bin/env python3
"""Stress of csrc/gemm_nt.hip's stream-K tail (experiments build): products whose tiles fall into MANY parts (RE2E_NT2_TAILWG = 1..3 unit ranges per CU),
every variant, x W^T and dy^T x, repeated launches beside a second stream that keeps the chip busy: each result against an fp64 product and against the
first launch (bitwise)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from robust_e2e_gan_amd import ops

DEV = 'cuda:0'


def main():
    torch.manual_seed(0)
    bad = 0
    side = torch.cuda.Stream()
    X = torch.randn(4096, 4096, device=DEV)
    for tn in (False, True):
        for (M, N, K) in ((512, 1024, 12800), (300, 1312, 4232), (2048, 512, 12800), (6400, 512, 4240), (1024, 256, 25600)):
            if tn:
                A, B = torch.randn(K, M, device=DEV), torch.randn(K, N, device=DEV)
                truth = (A.double().t() @ B.double())
            else:
                A, B = torch.randn(M, K, device=DEV), torch.randn(N, K, device=DEV)
                truth = (A.double() @ B.double().t())
            scale = truth.abs().max().item()
            for var in (('6,2',) if tn else ('3,2', '6,2', '8,2', '1,2')):
                for tail in ('1', '2', '3'):
                    os.environ['RE2E_TN2' if tn else 'RE2E_NT2'] = var
                    os.environ['RE2E_NT2_TAILWG'] = tail
                    first, worst, same = None, 0.0, True
                    for rep in range(12):
                        with torch.cuda.stream(side):                       # uneven load beside the product
                            if rep % 3:
                                torch.mm(X, X)
                        C = torch.full((M, N), float('nan'), device=DEV)
                        if tn:
                            ops.gemm(A, B, C, M, N, K, transa=True)
                        else:
                            ops.gemm(A, B, C, M, N, K, transb=True)
                        torch.cuda.synchronize()
                        err = (C.double() - truth).abs().max().item() / scale
                        worst = max(worst, err if err == err else 1e9)
                        if first is None:
                            first = C.clone()
                        elif not torch.equal(first, C):
                            same = False
                    flag = '' if (worst < 1e-4 and same) else '   <-- BAD'
                    bad += bool(flag)
                    print('%s %5dx%5dx%6d variant %s tail x%s: max rel err %.1e, bitwise repeatable %s%s' % ('TN' if tn else 'NT', M, N, K, var, tail, worst, same, flag), flush=True)
    print('BAD cases:', bad)


if __name__ == '__main__':
    main()
