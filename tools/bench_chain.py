#!/usr/bin/env python3
"""Microseconds per step of the persistent recurrences alone on the chip, at the shapes of the training configurations:
config 4 enhancer (T=800, B=32, H=256) and BLSTMP (T=200, B=64, H=512); config 5 enhancer (T=3000, B=8, H=256) and BLSTMP
(T=750, B=16, H=512).  One line per shape: forward, backward (best of 3 launches each)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from robust_e2e_gan_amd.lib import call, query

DEV = 'cuda:0'


def run(T, B, H, reps=5):
    g = torch.Generator().manual_seed(T + B + H)
    xg0 = [(torch.randn(T * B, 4 * H, generator=g) * 0.5).to(DEV) for _ in range(2)]
    whh = [(torch.randn(4 * H, H, generator=g) / H ** 0.5).to(DEV) for _ in range(2)]
    lens = torch.full((B,), T, dtype=torch.int32, device=DEV)
    dy = (torch.randn(T * B, 2 * H, generator=g) * 0.3).to(DEV)
    wsb = query('re2e_lstm_workspace_bytes', B, H)
    ws = torch.empty(wsb // 4 + 16, device=DEV)
    ybuf, cbuf = torch.zeros(T + 2, B, 2 * H, device=DEV), torch.zeros(T + 2, B, 2 * H, device=DEV)
    dc = torch.zeros(B, 2 * H, device=DEV)
    best = [1e9, 1e9]
    allt = [[], []]
    for rep in range(reps + 1):
        xg = [x.clone() for x in xg0]
        torch.cuda.synchronize()
        e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        e[0].record()
        call('re2e_lstm_seq_fwd', xg[0].data_ptr(), xg[1].data_ptr(), whh[0].data_ptr(), whh[1].data_ptr(), ybuf.data_ptr(), cbuf.data_ptr(),
             lens.data_ptr(), T, B, H, ws.data_ptr(), wsb)
        e[1].record()
        call('re2e_lstm_seq_bwd', xg[0].data_ptr(), xg[1].data_ptr(), whh[0].data_ptr(), whh[1].data_ptr(), dy.data_ptr(), ybuf.data_ptr(),
             cbuf.data_ptr(), dc.data_ptr(), lens.data_ptr(), T, B, H, None, ws.data_ptr(), wsb)
        e[2].record()
        torch.cuda.synchronize()
        if rep:
            best[0] = min(best[0], e[0].elapsed_time(e[1]) * 1e3 / T)
            best[1] = min(best[1], e[1].elapsed_time(e[2]) * 1e3 / T)
            allt[0].append(e[0].elapsed_time(e[1]) * 1e3 / T)
            allt[1].append(e[1].elapsed_time(e[2]) * 1e3 / T)
    fin = bool(torch.isfinite(ybuf).all()) and bool(torch.isfinite(xg[0]).all())
    med = [sorted(a)[len(a) // 2] for a in allt]
    print('T=%4d B=%2d H=%3d  fwd %5.2f (median %5.2f)  bwd %5.2f (median %5.2f) us/step%s' % (T, B, H, best[0], med[0], best[1], med[1], '' if fin else '  NON-FINITE'),
          flush=True)
    return best


if __name__ == '__main__':
    shapes = [(800, 32, 256), (200, 64, 512)]
    if '--all' in sys.argv:
        shapes += [(3000, 8, 256), (750, 16, 512)]
    for sh in shapes:
        run(*sh)
    print('aborts', query('re2e_lstm_abort_count'), ' env:', {k: v for k, v in os.environ.items() if k.startswith('RE2E_')}, flush=True)
