#!/usr/bin/env python3
"""Decoder loop alone: the persistent launch (csrc/decloop.hip) against the launch-per-step sequence, forward and forward+backward,
at config 4's (B=32, T'=200, L+1=41) and config 5's (B=8, T'=750, L+1=151) decoder shapes.   python tools/bench_decoder.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from robust_e2e_gan_amd import ops, lib

DEV = 'cuda:0'


def run(B, T, L1, E=512, A=320, D=300, C=10, Fh=100, reps=20):
    g = torch.Generator().manual_seed(1)
    r = lambda *s, scale=1.0: (torch.randn(*s, generator=g) * scale).to(DEV)
    hmask, pre = r(B, T, E).requires_grad_(True), r(B, T, A).requires_grad_(True)
    Pm = dict(embed=r(50, D, scale=0.5), w_ih=r(4 * D, D + E, scale=0.08), w_hh=r(4 * D, D, scale=0.08), b_ih=r(4 * D, scale=0.1), b_hh=r(4 * D, scale=0.1),
              mlp_dec=r(A, D, scale=0.1), mlp_att=r(A, C, scale=0.5), loc_conv=r(C, 1, 1, 2 * Fh + 1, scale=0.3), gvec_w=r(1, A, scale=0.3), gvec_b=r(1, scale=0.1))
    Pm = {k: torch.nn.Parameter(v) for k, v in Pm.items()}
    ids = torch.randint(0, 50, (L1, B), generator=g).to(torch.int32).to(DEV)
    hlens = torch.full((B,), T, dtype=torch.int32, device=DEV)
    for flag in (False, True):
        ops.DECODER_PERSIST = flag
        res = []
        for bwd in (False, True):
            for rep in range(3 + reps):
                if rep == 3:
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                if bwd:
                    z, w = ops.DecoderLoopFn.apply(hmask, pre, ids, hlens, L1, Pm)
                    z.sum().backward()
                else:
                    with torch.no_grad():
                        z, w = ops.DecoderLoopFn.apply(hmask, pre, ids, hlens, L1, Pm)
            torch.cuda.synchronize()
            res.append((time.perf_counter() - t0) / reps * 1e3)
        print('B=%d T=%d L1=%d  %-9s forward %.3f ms (%.1f us/token)   forward+backward %.3f ms' % (B, T, L1, 'persist' if flag else 'stepwise', res[0], res[0] * 1e3 / L1, res[1]),
              flush=True)
    print('aborts', lib.query('re2e_lstm_abort_count'))


if __name__ == '__main__':
    run(32, 200, 41)
    run(8, 750, 151)
