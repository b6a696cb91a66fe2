#!/usr/bin/env python3
"""Stress test of the persistent recurrences' in-launch hand-offs: the same sequences many times, beside an uneven filler load on
another stream, every output word compared with the launch-per-step kernels' result (and the abort counter)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from robust_e2e_gan_amd.lib import call, query

DEV = 'cuda:0'


def run(T, B, H, iters):
    g = torch.Generator().manual_seed(T * 7 + B + H)
    xg0 = [(torch.randn(T * B, 4 * H, generator=g) * 0.5).to(DEV) for _ in range(2)]
    whh = [(torch.randn(4 * H, H, generator=g) / H ** 0.5).to(DEV) for _ in range(2)]
    lens = torch.randint(max(1, T // 2), T + 1, (B,), generator=g, dtype=torch.int32)
    lens[0] = T
    lens = lens.to(DEV)
    dy = (torch.randn(T * B, 2 * H, generator=g) * 0.3).to(DEV)
    wsb = query('re2e_lstm_workspace_bytes', B, H)
    ws = torch.empty(wsb // 4 + 16, device=DEV)
    fa = torch.randn(3072, 3072, device=DEV)
    side = torch.cuda.Stream()
    main = torch.cuda.Stream()

    def once():
        xg = [x.clone() for x in xg0]
        ybuf, cbuf = torch.zeros(T + 2, B, 2 * H, device=DEV), torch.zeros(T + 2, B, 2 * H, device=DEV)
        call('re2e_lstm_seq_fwd', xg[0].data_ptr(), xg[1].data_ptr(), whh[0].data_ptr(), whh[1].data_ptr(), ybuf.data_ptr(), cbuf.data_ptr(),
             lens.data_ptr(), T, B, H, ws.data_ptr(), wsb)
        y = ybuf.clone()
        dc = torch.zeros(B, 2 * H, device=DEV)
        call('re2e_lstm_seq_bwd', xg[0].data_ptr(), xg[1].data_ptr(), whh[0].data_ptr(), whh[1].data_ptr(), dy.data_ptr(), ybuf.data_ptr(),
             cbuf.data_ptr(), dc.data_ptr(), lens.data_ptr(), T, B, H, None, ws.data_ptr(), wsb)
        return y, xg[0], xg[1]

    os.environ['RE2E_LSTM_PERSIST'] = '0'
    os.environ['RE2E_LSTM_PERSIST_BWD'] = '0'
    with torch.cuda.stream(main):
        ref = once()
    torch.cuda.synchronize()
    os.environ['RE2E_LSTM_PERSIST'] = '1'
    os.environ['RE2E_LSTM_PERSIST_BWD'] = '1'
    bad = 0
    for it in range(iters):
        if it % 3 != 2:                      # uneven load: sometimes a burst of GEMMs, sometimes nothing
            with torch.cuda.stream(side):
                for _ in range(1 + it % 4):
                    fa @ fa
        with torch.cuda.stream(main):
            out = once()
            for a, b in zip(out, ref):         # (compared on the stream that produced them)
                if not torch.isfinite(a).all() or (a - b).abs().max().item() > 2e-5 * (1.0 + b.abs().max().item()):
                    bad += 1
                    break
    torch.cuda.synchronize()
    print('T=%d B=%d H=%d: %d iterations, %d mismatching, aborts %d' % (T, B, H, iters, bad, query('re2e_lstm_abort_count')), flush=True)
    return bad


if __name__ == '__main__':
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 150
    total = run(100, 32, 256, n) + run(40, 64, 512, n) + run(60, 40, 320, n) + run(30, 5, 32, n)
    sys.exit(1 if total else 0)
