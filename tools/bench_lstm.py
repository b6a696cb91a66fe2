#!/usr/bin/env python3
"""Micro-benchmark of the K4 recurrence kernels (us per time step) at the config-4 shapes."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from robust_e2e_gan_amd.lib import call, query


def run(T, B, H, reps=3):
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
    res = {}
    for name in ('fwd', 'bwd'):
        ts = []
        for r in range(reps + 1):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            if name == 'fwd':
                call('re2e_lstm_seq_fwd', xg[0].data_ptr(), xg[1].data_ptr(), whh[0].data_ptr(), whh[1].data_ptr(), ybuf.data_ptr(), cbuf.data_ptr(),
                     lens.data_ptr(), T, B, H, ws.data_ptr(), wsb)
            else:
                call('re2e_lstm_seq_bwd', xg[0].data_ptr(), xg[1].data_ptr(), whh[0].data_ptr(), whh[1].data_ptr(), dy.data_ptr(), ybuf.data_ptr(),
                     cbuf.data_ptr(), dc.data_ptr(), lens.data_ptr(), T, B, H, None, ws.data_ptr(), wsb)
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) / T * 1e6)
        res[name] = min(ts[1:])
    return res


if __name__ == '__main__':
    for (T, B, H) in ((800, 32, 256), (200, 64, 512)):
        for wf in (4, 8, 16):
            os.environ['RE2E_LSTM_WAVES_FWD'] = str(wf)
            r = run(T, B, H)
            print('T=%d B=%d H=%d waves=%2d: fwd %.2f us/step  bwd %.2f us/step (wall, incl. launch gaps)' % (T, B, H, wf, r['fwd'], r['bwd']), flush=True)
