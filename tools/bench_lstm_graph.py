#!/usr/bin/env python3
"""Is the K4 recurrence launch-rate bound?  Times one sequence launched kernel by kernel against the same
launches replayed from a captured hipGraph (GPU time by events, host enqueue time by perf_counter)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from robust_e2e_gan_amd.lib import call, query


def run(T, B, H):
    dev = 'cuda:0'
    xg = [torch.randn(T * B, 4 * H, device=dev) * 0.1 for _ in range(2)]
    whh = [torch.randn(4 * H, H, device=dev) * 0.05 for _ in range(2)]
    ybuf = torch.zeros(T + 2, B, 2 * H, device=dev)
    cbuf = torch.zeros(T + 2, B, 2 * H, device=dev)
    lens = torch.full((B,), T, dtype=torch.int32, device=dev)
    dy = torch.randn(T * B, 2 * H, device=dev) * 0.1
    dc = torch.zeros(B, 2 * H, device=dev)
    wsb = query('re2e_lstm_workspace_bytes', B, H)
    ws = torch.empty(wsb // 4 + 16, device=dev)

    def fwd():
        call('re2e_lstm_seq_fwd', xg[0].data_ptr(), xg[1].data_ptr(), whh[0].data_ptr(), whh[1].data_ptr(), ybuf.data_ptr(), cbuf.data_ptr(),
             lens.data_ptr(), T, B, H, ws.data_ptr(), wsb)

    def bwd():
        call('re2e_lstm_seq_bwd', xg[0].data_ptr(), xg[1].data_ptr(), whh[0].data_ptr(), whh[1].data_ptr(), dy.data_ptr(), ybuf.data_ptr(),
             cbuf.data_ptr(), dc.data_ptr(), lens.data_ptr(), T, B, H, None, ws.data_ptr(), wsb)

    for name, fn in (('fwd', fwd), ('bwd', bwd)):
        fn(); fn()
        torch.cuda.synchronize()
        res = []
        side = torch.cuda.Stream()
        with torch.cuda.stream(side):
            fn()
            side.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=side):
                fn()
        torch.cuda.synchronize()
        for mode, f in (('direct', fn), ('graph', g.replay)):
            best = (1e9, 1e9)
            for _ in range(4):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                e0.record()
                f()
                e1.record()
                th = time.perf_counter() - t0
                torch.cuda.synchronize()
                best = min(best, (e0.elapsed_time(e1) * 1e3 / T, th * 1e6 / T))
            res.append('%s gpu %.2f us/step host %.2f us/step' % (mode, best[0], best[1]))
        print('T=%d B=%d H=%d %s: %s' % (T, B, H, name, ' | '.join(res)), flush=True)


if __name__ == '__main__':
    for (T, B, H) in ((800, 32, 256), (200, 64, 512)):
        run(T, B, H)
