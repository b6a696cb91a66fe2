#!/usr/bin/env python3
"""K7 in isolation at the config-4 shapes (B=32, T'=200, eprojs=512, dunits=300, adim=320, 10 channels, 201 taps): 41 dependent
forward steps and 41 backward steps, time per step by events; run under `rocprofv3 --kernel-trace --stats` for the per-kernel split."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from robust_e2e_gan_amd.lib import call, query

DEV = 'cuda:0'


def main(B=32, T=200, E=512, D=300, A=320, C=10, Fh=100, L1=41):
    g = torch.Generator().manual_seed(0)
    r = lambda *s: (torch.randn(*s, generator=g) * 0.3).to(DEV)
    pre, enc, z = r(B, T, A), r(B, T, E), r(L1, B, D)
    w_decT, w_att, w_conv, gvec, gb = r(D, A), r(A, C), r(C, 2 * Fh + 1), r(A), r(1)
    hlens = torch.full((B,), T, dtype=torch.int32, device=DEV)
    w = torch.zeros(L1, B, T, device=DEV)
    cx, conv, dpj, e_scr = torch.zeros(L1, B, E, device=DEV), torch.zeros(L1, B, T, C, device=DEV), torch.zeros(L1, B, A, device=DEV), torch.zeros(B, T, device=DEV)
    dc = r(L1, B, E)
    de_all, dw = torch.zeros(L1, B, T, device=DEV), [torch.zeros(B, T, device=DEV) for _ in range(2)]
    ddp = torch.zeros(L1, B, A, device=DEV)
    npart = query('re2e_attloc_partial_floats', A, C, Fh)
    partials = torch.zeros(B, npart, device=DEV)
    awsb = query('re2e_attloc_workspace_bytes', B, T, A, C)
    aws = torch.empty(awsb // 4 + 16, device=DEV)
    d_pre = torch.empty(B, T, A, device=DEV)

    def fwd():
        for i in range(L1):
            call('re2e_attloc_fwd', pre.data_ptr(), enc.data_ptr(), z[i].data_ptr(), w[i - 1].data_ptr() if i > 0 else None, hlens.data_ptr(),
                 w_decT.data_ptr(), w_att.data_ptr(), w_conv.data_ptr(), gvec.data_ptr(), gb.data_ptr(), B, T, E, D, A, C, Fh, w[i].data_ptr(),
                 cx[i].data_ptr(), E, conv[i].data_ptr(), dpj[i].data_ptr(), e_scr.data_ptr())

    def bwd():
        have = False
        a, b = dw
        for i in range(L1 - 1, -1, -1):
            call('re2e_attloc_bwd', pre.data_ptr(), enc.data_ptr(), w[i - 1].data_ptr() if i > 0 else None, w[i].data_ptr(), hlens.data_ptr(),
                 w_att.data_ptr(), w_conv.data_ptr(), gvec.data_ptr(), conv[i].data_ptr(), dpj[i].data_ptr(), cx[i].data_ptr(), dc[i].data_ptr(), E,
                 a.data_ptr() if have else None, B, T, E, A, C, Fh, de_all[i].data_ptr(), b.data_ptr() if i > 0 else None, ddp[i].data_ptr(),
                 partials.data_ptr(), aws.data_ptr(), awsb)
            a, b = b, a
            have = True
        call('re2e_attloc_dpre', pre.data_ptr(), conv.data_ptr(), dpj.data_ptr(), de_all.data_ptr(), w_att.data_ptr(), gvec.data_ptr(), L1, B, T, A, C, Fh,
             d_pre.data_ptr(), partials.data_ptr(), aws.data_ptr(), awsb)

    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        for name, fn in (('forward', fwd), ('backward', bwd)):
            fn()
            best = 1e9
            for _ in range(3):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                st.synchronize()
                e0.record()
                fn()
                e1.record()
                st.synchronize()
                best = min(best, e0.elapsed_time(e1) * 1e3 / L1)
            print('attloc %s: %.1f us per step (%d steps)' % (name, best, L1), flush=True)


if __name__ == '__main__':
    main()
