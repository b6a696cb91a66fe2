"""Time single kernels at the config-4 shapes through the package's ops (HIP events on the current stream).
usage: python tools/bench_kernels.py [name ...]"""
import sys, time
import torch
sys.path.insert(0, '.')
from robust_e2e_gan_amd import lib, ops
DEV = 'cuda:0'


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def conv1_1_dgrad():
    B, T, F = 32, 800, 80
    dz = torch.randn(B, T, F, 64, device=DEV)
    w = torch.randn(64, 1, 3, 3, device=DEV)
    us = timeit(lambda: ops.conv_dgrad(dz, w, (B, T, F, 1), 1, 1))
    print('conv1_1 dgrad  %.1f us  %.2f TB/s' % (us, dz.numel() * 4 / us / 1e6))


def fbank():
    from robust_e2e_gan_amd.model.feat_model import band_from_matrix, mel_matrix
    B, T = 32, 800
    band = band_from_matrix(torch.from_numpy(mel_matrix()), DEV)
    x = (torch.randn(B, T, 257, device=DEV) * 20).abs().requires_grad_(True)
    cm = torch.stack([torch.linspace(10, 14, 80), torch.linspace(0.3, 0.6, 80)]).to(DEV)
    us = timeit(lambda: ops.fbank(x.detach(), band, cm, True, True))
    print('fbank fwd (raw + norm)  %.1f us' % us)
    raw, nrm = ops.fbank(x, band, cm, True, True)
    g0, g1 = torch.randn_like(raw), torch.randn_like(nrm)
    off, ln, w, maxw, NF, toff, tln, tw, maxc = band
    dx, pw = torch.empty_like(x), torch.empty_like(raw)
    lib.call('re2e_fbank_fwd', x.data_ptr(), B * T, 257, NF, off.data_ptr(), ln.data_ptr(), w.data_ptr(), maxw, 0, 0, 0, pw.data_ptr())
    us = timeit(lambda: lib.call('re2e_fbank_fwd', x.data_ptr(), B * T, 257, NF, off.data_ptr(), ln.data_ptr(), w.data_ptr(), maxw, raw.data_ptr(),
                                 nrm.data_ptr(), cm.data_ptr(), pw.data_ptr()))
    print('fbank fwd direct call (raw + norm + pw)  %.1f us' % us)
    us = timeit(lambda: lib.call('re2e_fbank_bwd', x.data_ptr(), B * T, 257, NF, toff.data_ptr(), tln.data_ptr(), tw.data_ptr(), maxc, pw.data_ptr(),
                                 g0.data_ptr(), g1.data_ptr(), cm.data_ptr(), dx.data_ptr()))
    print('fbank bwd  %.1f us' % us)


def d_conv4():
    import torch.nn.functional as Fn
    N, H, W, C, K = 32, 100, 10, 256, 512
    x = torch.randn(N, H, W, C, device=DEV)
    Wt = torch.randn(K, C, 4, 4, device=DEV) * 0.02
    dz = torch.randn(N, H - 1, W - 1, K, device=DEV)
    fl = 2.0 * 16 * C * K * N * (H - 1) * (W - 1)
    us = timeit(lambda: ops.conv4x4_wino(x, Wt, K, 1))
    print('D conv4 fwd  wino44 %.1f us  %.1f TFLOP/s direct-equivalent' % (us, fl / us / 1e6))
    us = timeit(lambda: ops.conv4x4_wino(dz, Wt, C, 2, dgrad=True))
    print('D conv4 dgrad wino44 %.1f us  %.1f TFLOP/s direct-equivalent' % (us, 2.0 * 16 * C * K * N * H * W / us / 1e6))
    gw = torch.zeros_like(Wt)
    wsb = lib.query('re2e_conv4x4_wino_wgrad_workspace_bytes', N, H, W, C, K, 1)
    ws = lib.workspace(wsb, x.device, 'b')
    us = timeit(lambda: lib.call('re2e_conv4x4_wino_wgrad', x.data_ptr(), N, H, W, C, dz.data_ptr(), K, 1, gw.data_ptr(), 0.0, ws.data_ptr(), wsb))
    print('D conv4 wgrad wino44 %.1f us  %.1f TFLOP/s direct-equivalent' % (us, fl / us / 1e6))
    y = ops.conv4x4_wino(x, Wt, K, 1)
    ref = Fn.conv2d(x.permute(0, 3, 1, 2), Wt, padding=1).permute(0, 2, 3, 1)
    print('  fwd max rel err vs torch %.2e' % ((y - ref).abs().max() / ref.abs().max()).item())


def vgg_wgrad():
    for (N, H, W, C, K) in ((64, 800, 80, 64, 64), (64, 400, 40, 64, 128), (64, 400, 40, 128, 128)):
        x = torch.randn(N, H, W, C, device=DEV)
        dz = torch.randn(N, H, W, K, device=DEV)
        gw = torch.zeros(K, C, 3, 3, device=DEV)
        fl = 2.0 * 9 * C * K * N * H * W
        wsb = lib.query('re2e_conv3x3_wino_wgrad_workspace_bytes', N, H, W, C, K)
        ws = lib.workspace(wsb, x.device, 'b')
        us = timeit(lambda: lib.call('re2e_conv3x3_wino_wgrad', x.data_ptr(), N, H, W, C, dz.data_ptr(), K, gw.data_ptr(), 0.0, ws.data_ptr(), wsb), n=10)
        wsb2 = lib.query('re2e_conv_wgrad_workspace_bytes', N, H, W, C, K, 3, 3)
        ws2 = lib.workspace(wsb2, x.device, 'b2')
        gw2 = torch.zeros_like(gw)
        us2 = timeit(lambda: lib.call('re2e_conv_wgrad', x.data_ptr(), N, H, W, C, dz.data_ptr(), K, 3, 3, H, W, 1, 1, -1, -1, gw2.data_ptr(), 0.0, ws2.data_ptr(), wsb2), n=10)
        err = ((gw - gw2).abs().max() / gw2.abs().max()).item()
        print('3x3 wgrad %dx%dx%d %d->%d  wino %.1f us (%.1f TFLOP/s direct-equivalent, %.1f executed)  direct %.1f us (%.1f)  rel diff %.1e'
              % (N, H, W, C, K, us, fl / us / 1e6, fl / 2.25 / us / 1e6, us2, fl / us2 / 1e6, err))
        del x, dz


ALL = {'vgg_wgrad': vgg_wgrad, 'conv1_1_dgrad': conv1_1_dgrad, 'fbank': fbank, 'd_conv4': d_conv4}
if __name__ == '__main__':
    for n in (sys.argv[1:] or ALL):
        ALL[n]()
