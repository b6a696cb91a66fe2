#!/usr/bin/env python3
"""Persistent (one launch per sequence) vs launch-per-step K4 recurrence, forward and backward: comparison of every output
and the time per step alone on the chip (beside filler kernels: tools/bench_fill_under_chain.py)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from robust_e2e_gan_amd.lib import call, query


def run(T, B, H, ragged=False, which='fwd'):
    dev = 'cuda:0'
    g = torch.Generator(device='cpu').manual_seed(T * 131 + B * 7 + H)
    xg0 = [(torch.randn(T * B, 4 * H, generator=g) * 0.5).to(dev) for _ in range(2)]
    whh = [(torch.randn(4 * H, H, generator=g) * (1.0 / H ** 0.5)).to(dev) for _ in range(2)]
    lens = torch.full((B,), T, dtype=torch.int32)
    if ragged:
        lens = torch.randint(max(1, T // 3), T + 1, (B,), generator=g, dtype=torch.int32)
        lens[0] = T
    lens = lens.to(dev)
    wsb = query('re2e_lstm_workspace_bytes', B, H)
    ws = torch.empty(wsb // 4 + 16, device=dev)
    filler_a = torch.randn(4096, 4096, device=dev)
    side = torch.cuda.Stream()
    out = {}
    for mode in ('0', '1'):
        os.environ['RE2E_LSTM_PERSIST'] = mode
        res = []
        for load in (False,):
            best = 1e9
            for rep in range(3):
                xg = [x.clone() for x in xg0]
                ybuf = torch.zeros(T + 2, B, 2 * H, device=dev)
                cbuf = torch.zeros(T + 2, B, 2 * H, device=dev)
                torch.cuda.synchronize()
                if load:
                    side.wait_stream(torch.cuda.current_stream())
                    with torch.cuda.stream(side):
                        for _ in range(40):
                            filler_a @ filler_a
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                call('re2e_lstm_seq_fwd', xg[0].data_ptr(), xg[1].data_ptr(), whh[0].data_ptr(), whh[1].data_ptr(), ybuf.data_ptr(),
                     cbuf.data_ptr(), lens.data_ptr(), T, B, H, ws.data_ptr(), wsb)
                e1.record()
                torch.cuda.synchronize()
                best = min(best, e0.elapsed_time(e1) * 1e3 / T)
            res.append(best)
        out[mode] = (xg, ybuf, cbuf)
        print('T=%d B=%d H=%d %s fwd persist=%s: %.2f us/step alone' % (T, B, H, 'ragged' if ragged else 'full', mode, res[0]), flush=True)
    # backward: persistent vs stepwise from the forward state of the last run
    xgf, ybuf, cbuf = out['1']
    dy = torch.randn(T * B, 2 * H, device=dev) * 0.3
    bout = {}
    prio = torch.cuda.Stream()
    for mode in ('0', '1'):
        os.environ['RE2E_LSTM_PERSIST_BWD'] = mode
        best = 1e9
        for rep in range(3):
            G = [x.clone() for x in xgf]
            dc = torch.zeros(B, 2 * H, device=dev)
            torch.cuda.synchronize()
            with torch.cuda.stream(prio):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                call('re2e_lstm_seq_bwd', G[0].data_ptr(), G[1].data_ptr(), whh[0].data_ptr(), whh[1].data_ptr(), dy.data_ptr(), ybuf.data_ptr(),
                     cbuf.data_ptr(), dc.data_ptr(), lens.data_ptr(), T, B, H, None, ws.data_ptr(), wsb)
                e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) * 1e3 / T)
        bout[mode] = G
        print('T=%d B=%d H=%d bwd persist=%s: %.2f us/step alone' % (T, B, H, mode, best), flush=True)
    bd = [float((bout['0'][d] - bout['1'][d]).abs().max()) for d in range(2)]
    print('   bwd max |diff| dgates_f %.3g dgates_r %.3g%s' % (bd[0], bd[1], '' if all(torch.isfinite(t).all() for t in bout['1']) else '  NON-FINITE'), flush=True)
    a, b = out['0'], out['1']
    diffs = [float((a[0][0] - b[0][0]).abs().max()), float((a[0][1] - b[0][1]).abs().max()), float((a[1] - b[1]).abs().max()),
             float((a[2] - b[2]).abs().max())]
    bad = not all(torch.isfinite(t).all() for t in (b[0][0], b[0][1], b[1], b[2]))
    print('   max |diff| gates_f %.3g gates_r %.3g y %.3g c %.3g%s' % (*diffs, '  NON-FINITE' if bad else ''), flush=True)


if __name__ == '__main__':
    run(40, 32, 256)
    run(800, 32, 256)
    run(200, 64, 512)
    run(57, 40, 320, ragged=True)
    run(33, 5, 128, ragged=True)
    run(21, 3, 32, ragged=True)
