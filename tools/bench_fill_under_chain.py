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


def main(filler='conv', which='fwd'):
    N, H, W, C, K = 32, 400, 40, 128, 128
    x = torch.randn(N, H, W, C, device=DEV)
    wg = torch.randn(K, 3, 3, C, device=DEV) * 0.04
    y = torch.empty(N, H, W, K, device=DEV)
    conv_args = (x.data_ptr(), N, H, W, C, wg.data_ptr(), K, 3, 3, H, W, 1, 1, 1, 1, -1, -1, y.data_ptr(), H, W, 1, 1, 0, 0, None, lib.ACT_RELU, 0.0)
    T, B, Hh = int(os.environ.get('CHAIN_T', '800')), int(os.environ.get('CHAIN_B', '32')), int(os.environ.get('CHAIN_H', '256'))     # enhancer layer by default; 200 / 64 / 512 = BLSTMP
    xg = [torch.randn(T * B, 4 * Hh, device=DEV) * 0.1 for _ in range(2)]
    whh = [torch.randn(4 * Hh, Hh, device=DEV) * 0.05 for _ in range(2)]
    ybuf, cbuf = torch.zeros(T + 2, B, 2 * Hh, device=DEV), torch.zeros(T + 2, B, 2 * Hh, device=DEV)
    lens = torch.full((B,), T, dtype=torch.int32, device=DEV)
    wsb = query('re2e_lstm_workspace_bytes', B, Hh)
    ws = torch.empty(wsb // 4 + 16, device=DEV)

    dy = torch.randn(T * B, 2 * Hh, device=DEV) * 0.1
    dcs = torch.zeros(B, 2 * Hh, device=DEV)
    big_a, big_b = torch.empty(256 << 20, device=DEV), torch.empty(256 << 20, device=DEV)     # 1 GiB each: HBM-streaming filler

    def chain():
        if which == 'fwd':
            call('re2e_lstm_seq_fwd', xg[0].data_ptr(), xg[1].data_ptr(), whh[0].data_ptr(), whh[1].data_ptr(), ybuf.data_ptr(), cbuf.data_ptr(),
                 lens.data_ptr(), T, B, Hh, ws.data_ptr(), wsb)
        else:
            call('re2e_lstm_seq_bwd', xg[0].data_ptr(), xg[1].data_ptr(), whh[0].data_ptr(), whh[1].data_ptr(), dy.data_ptr(), ybuf.data_ptr(),
                 cbuf.data_ptr(), dcs.data_ptr(), lens.data_ptr(), T, B, Hh, None, ws.data_ptr(), wsb)

    from robust_e2e_gan_amd import ops
    vx = torch.randn(32, 800, 80, 1, device=DEV)
    vw = [torch.randn(64, 1, 3, 3, device=DEV) * 0.3, torch.randn(64, 64, 3, 3, device=DEV) * 0.04, torch.randn(128, 64, 3, 3, device=DEV) * 0.04,
          torch.randn(128, 128, 3, 3, device=DEV) * 0.03]
    vb = [torch.zeros(64, device=DEV), torch.zeros(64, device=DEV), torch.zeros(128, device=DEV), torch.zeros(128, device=DEV)]

    def vgg(upto=4):
        with torch.no_grad():
            h = ops.conv2d(vx, vw[0], vb[0], 1, 1, 'relu')
            if upto >= 2:
                h = ops.conv2d(h, vw[1], vb[1], 1, 1, 'relu')
                h = ops.maxpool2(h)
            if upto >= 3:
                h = ops.conv2d(h, vw[2], vb[2], 1, 1, 'relu')
            if upto >= 4:
                h = ops.conv2d(h, vw[3], vb[3], 1, 1, 'relu')
                h = ops.maxpool2(h)
        return h

    gan = None
    if filler.startswith('dreal'):
        import bench
        from robust_e2e_gan_amd.joint_train import config4_opt
        from robust_e2e_gan_amd.model.gan_model import GANLoss
        opt = config4_opt()
        gan = bench.build(opt, torch.device(DEV))[3]
        crit = GANLoss().to(DEV)
        feat = torch.randn(32, 800, 80, device=DEV)
        gparams = [p for p in gan.parameters()]
        wgrad_stream = lib.cu_masked_stream(224, 256, DEV)

    def dreal():
        # dreal: forward + backward with the weight gradients on a second masked stream (as in the step); drealf: forward only;
        # dreal1: forward + backward on ONE stream
        ops.MULTI_STREAM = filler == 'dreal'
        ops.WGRAD_STREAM = wgrad_stream if filler == 'dreal' else None
        if filler == 'drealf':
            with torch.no_grad():
                gan(feat, None)
            return
        loss = crit(gan(feat, None), True)
        torch.autograd.grad(loss * 0.5, gparams, allow_unused=True)
        if ops.WGRAD_STREAM is not None:
            torch.cuda.current_stream().wait_stream(ops.WGRAD_STREAM)

    def fill():
        if gan is not None:
            dreal()
        elif filler.startswith('vgg'):
            for _ in range(2):
                vgg(int(filler[3:] or 4))
        elif filler == 'hbm':
            for _ in range(6):
                big_b.copy_(big_a)
        else:
            for _ in range(nconv):
                call('re2e_conv_igemm', *conv_args)

    nconv = 12
    # NB: never the legacy default stream -- it synchronises implicitly with the (blocking) CU-masked streams and would
    # run chain and filler one after the other
    prio = torch.cuda.Stream(priority=-1 if os.environ.get('CHAIN_PRIO') else 0)
    if os.environ.get('CHAIN_MASK'):           # chain confined to the 32 CUs the masked fillers never use
        prio = lib.cu_masked_stream(32, 256, DEV, first=224)
    for masked in (0, 224):
        side = lib.cu_masked_stream(masked, 256, DEV) if masked else torch.cuda.Stream()
        res = {}
        for mode in ('filler alone', 'chain alone', 'both'):
            best = None
            for _ in range(3):
                torch.cuda.synchronize()
                e = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
                def run_fill():
                    if mode != 'chain alone':
                        with torch.cuda.stream(side):
                            e[0].record()
                            fill()
                            e[1].record()

                def run_chain():
                    if mode != 'filler alone':
                        with torch.cuda.stream(prio):
                            e[2].record()
                            chain()
                            e[3].record()
                if os.environ.get('CHAIN_FIRST'):      # the chain's workgroups are placed on an EMPTY chip, the filler arrives later
                    run_chain(); run_fill()
                else:
                    run_fill(); run_chain()
                torch.cuda.synchronize()
                t = (e[0].elapsed_time(e[1]) if mode != 'chain alone' else 0.0, e[2].elapsed_time(e[3]) if mode != 'filler alone' else 0.0)
                best = t if best is None or sum(t) < sum(best) else best
            res[mode] = best
        # algorithmic FLOPs of ONE fill() -- per filler (round 1 priced every filler with the nconv-convolution formula and printed
        # figures above the 157.3 TFLOP/s peak for the VGG filler); fillers with no defined FLOP count print n/a
        if gan is not None:
            fl = None                       # discriminator forward / backward variants: read the ms
        elif filler.startswith('vgg'):
            upto = int(filler[3:] or 4)
            px1, px2 = 32 * 800 * 80, 32 * 400 * 40
            per = [2.0 * 9 * 1 * 64 * px1, 2.0 * 9 * 64 * 64 * px1, 2.0 * 9 * 64 * 128 * px2, 2.0 * 9 * 128 * 128 * px2]
            fl = 2 * sum(per[:upto])        # fill() runs the stack twice
        elif filler == 'hbm':
            fl = None
        else:
            fl = 2.0 * 9 * C * K * N * H * W * nconv
        tf = lambda ms: ('%.1f TFLOP/s' % (fl / ms * 1e-9)) if fl else 'n/a'
        print('chain %s (persist=%s, %s priority), filler %s%s on a stream %s:' % (which, os.environ.get('RE2E_LSTM_PERSIST', '1'),
                                                                       'high' if os.environ.get('CHAIN_PRIO') else 'normal', filler,
                                                                       ' WITHOUT memory traffic' if os.environ.get('RE2E_IGEMM_NOMEM') else '',
                                                                       'CU-masked to %d' % masked if masked else 'unmasked'))
        print('   filler alone %.2f ms (%s) | chain alone %.2f ms (%.2f us/step)' % (res['filler alone'][0], tf(res['filler alone'][0]),
                                                                                  res['chain alone'][1], res['chain alone'][1] * 1e3 / T))
        print('   together: filler %.2f ms (%s), chain %.2f ms (%.2f us/step)' % (res['both'][0], tf(res['both'][0]), res['both'][1],
                                                                                res['both'][1] * 1e3 / T))


if __name__ == '__main__':
    main(*sys.argv[1:])
