#!/usr/bin/env python3
"""How fast does filler MFMA work run while a recurrent chain occupies the main stream?  Times a batch of implicit-GEMM
conv launches on a side stream (a) alone and (b) while an 800-step bi-LSTM chain runs on the main stream, and the chain
alone / with the filler: the step schedule of the trainer hides filler work under the chains at exactly this exchange rate."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from robust_e2e_gan_amd import lib
from robust_e2e_gan_amd.lib import call, query

DEV = 'cuda:0'


def main():
    N, H, W, C, K = 32, 400, 40, 128, 128
    x = torch.randn(N, H, W, C, device=DEV)
    wg = torch.randn(K, 3, 3, C, device=DEV) * 0.04
    y = torch.empty(N, H, W, K, device=DEV)
    conv_args = (x.data_ptr(), N, H, W, C, wg.data_ptr(), K, 3, 3, H, W, 1, 1, 1, 1, -1, -1, y.data_ptr(), H, W, 1, 1, 0, 0, None, lib.ACT_RELU, 0.0)
    T, B, Hh = 800, 32, 256
    xg = [torch.randn(T * B, 4 * Hh, device=DEV) * 0.1 for _ in range(2)]
    whh = [torch.randn(4 * Hh, Hh, device=DEV) * 0.05 for _ in range(2)]
    ybuf, cbuf = torch.zeros(T + 2, B, 2 * Hh, device=DEV), torch.zeros(T + 2, B, 2 * Hh, device=DEV)
    lens = torch.full((B,), T, dtype=torch.int32, device=DEV)
    wsb = query('re2e_lstm_workspace_bytes', B, Hh)
    ws = torch.empty(wsb // 4 + 16, device=DEV)

    def chain():
        call('re2e_lstm_seq_fwd', xg[0].data_ptr(), xg[1].data_ptr(), whh[0].data_ptr(), whh[1].data_ptr(), ybuf.data_ptr(), cbuf.data_ptr(),
             lens.data_ptr(), T, B, Hh, ws.data_ptr(), wsb)

    nconv = 12
    for masked in (0, 224):
        side = lib.cu_masked_stream(masked, 256, DEV) if masked else torch.cuda.Stream()
        res = {}
        for mode in ('filler alone', 'chain alone', 'both'):
            best = None
            for _ in range(3):
                torch.cuda.synchronize()
                e = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
                if mode != 'chain alone':
                    with torch.cuda.stream(side):
                        e[0].record()
                        for _ in range(nconv):
                            call('re2e_conv_igemm', *conv_args)
                        e[1].record()
                if mode != 'filler alone':
                    e[2].record()
                    chain()
                    e[3].record()
                torch.cuda.synchronize()
                t = (e[0].elapsed_time(e[1]) if mode != 'chain alone' else 0.0, e[2].elapsed_time(e[3]) if mode != 'filler alone' else 0.0)
                best = t if best is None or sum(t) < sum(best) else best
            res[mode] = best
        fl = 2.0 * 9 * C * K * N * H * W * nconv
        print('filler stream %s:' % ('CU-masked to %d' % masked if masked else 'unmasked'))
        print('   filler alone %.2f ms (%.1f TFLOP/s) | chain alone %.2f ms (%.2f us/step)' % (res['filler alone'][0], fl / res['filler alone'][0] * 1e-9,
                                                                                              res['chain alone'][1], res['chain alone'][1] * 1e3 / T))
        print('   together: filler %.2f ms (%.1f TFLOP/s), chain %.2f ms (%.2f us/step)' % (res['both'][0], fl / res['both'][0] * 1e-9, res['both'][1],
                                                                                            res['both'][1] * 1e3 / T))


if __name__ == '__main__':
    main()
