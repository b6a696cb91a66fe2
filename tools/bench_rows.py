"""x W^T and dy^T x over the valid rows of a ragged time-major batch (ops.gemm_rows / gemm_tn_rows) against the same products over all rows:
us per call alone on the chip and TFLOP/s over the rows actually computed.   python tools/bench_rows.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from robust_e2e_gan_amd import ops
from robust_e2e_gan_amd.model.e2e_common import lens_dev

DEV = 'cuda:0'


def timed(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


print('%-6s %-22s %5s | %9s %7s | %9s %7s | %6s' % ('form', 'rows x N x K', 'valid', 'all us', 'TF/s', 'rows us', 'TF/s', 'ratio'))
for T, B, N, K in ((200, 64, 2048, 2560), (200, 64, 2560, 2048), (200, 64, 2048, 512), (200, 64, 512, 2048), (200, 64, 512, 1024), (800, 32, 1024, 512), (800, 32, 512, 1024),
                   (800, 32, 1024, 260)):
    lens = [int(round(T * (1 - 0.3 * i / 31))) for i in range(32)] * (B // 32)
    maps = ops.row_maps(lens_dev(lens, DEV), T, B)
    M = T * B
    a, w = torch.randn(M, K, device=DEV), torch.randn(N, K, device=DEV)
    c = torch.empty(M, N, device=DEV)
    t0 = timed(lambda: ops.gemm(a, w, c, M, N, K, transb=True))
    t1 = timed(lambda: ops.gemm_rows(a, w, c, N, K, maps))
    print('%-6s %-22s %5.3f | %9.1f %7.1f | %9.1f %7.1f | %6.3f' % ('x W^T', '%dx%dx%d' % (M, N, K), maps.nv / M, t0, 2.0 * M * N * K / t0 / 1e6, t1,
                                                                  2.0 * maps.nv * N * K / t1 / 1e6, t1 / t0), flush=True)
for T, B, Mo, No in ((200, 64, 2048, 2560), (200, 64, 2048, 512), (800, 32, 1024, 512), (800, 32, 1024, 256), (200, 64, 512, 1024)):
    lens = [int(round(T * (1 - 0.3 * i / 31))) for i in range(32)] * (B // 32)
    maps = ops.row_maps(lens_dev(lens, DEV), T, B)
    R = T * B
    a, x = torch.randn(R, Mo, device=DEV), torch.randn(R, No, device=DEV)
    c = torch.empty(Mo, No, device=DEV)
    t0 = timed(lambda: ops.gemm(a, x, c, Mo, No, R, transa=True))
    t1 = timed(lambda: ops.gemm_tn_rows(a, x, c, Mo, No, maps))
    print('%-6s %-22s %5.3f | %9.1f %7.1f | %9.1f %7.1f | %6.3f' % ('dy^T x', '%dx%dx%d' % (Mo, No, R), maps.nv / R, t0, 2.0 * Mo * No * R / t0 / 1e6, t1,
                                                                  2.0 * Mo * No * maps.nv / t1 / 1e6, t1 / t0), flush=True)
