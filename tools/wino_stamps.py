#!/usr/bin/env python3
"""Phase timing of the Winograd kernel from s_memtime stamps (RE2E_EXPERIMENTS build): per wavefront, cycles (100 MHz memtime ticks
are converted by the measured ratio) between  0 start | 1 first loads issued | 2 main loop done | 3 R written | 4 barrier 1 |
5 stores issued | 6 barrier 2.   RE2E_LIB=.../libre2e_hip_exp.so RE2E_EXPERIMENTS=1 python tools/wino_stamps.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

dev = 'cuda:0'
st = torch.zeros(4096 * 4 * 8, dtype=torch.int64, device=dev)
os.environ['RE2E_WINO_STAMPS'] = '%x' % st.data_ptr()
os.environ.setdefault('RE2E_WINO_IPW', '1')
from robust_e2e_gan_amd import ops   # noqa: E402

N, H, W, C, K = 64, 800, 80, 64, 64
x = torch.randn(N, H, W, C, device=dev)
Wt = torch.randn(K, C, 3, 3, device=dev) * 0.04
b = torch.zeros(K, device=dev)
for _ in range(3):
    ops.conv3x3_wino(x, Wt, K, bias=b, relu=True)
torch.cuda.synchronize()
s = st.view(4096, 4, 8).cpu().double()
s = s[s[:, 0, 0] > 0]
print('workgroups stamped:', s.shape[0])
d = s[:, :, 1:7] - s[:, :, 0:6]
names = ['prologue', 'main loop', 'R calc + LDS write', 'barrier 1', 'stage 2 + stores', 'barrier 2']
tot = (s[:, :, 6] - s[:, :, 0])
print('memtime ticks per phase (mean over the stamped workgroups x 4 waves; items per workgroup = %s)' % os.environ['RE2E_WINO_IPW'])
for i, nme in enumerate(names):
    print('  %-20s %9.1f  (%4.1f %%)' % (nme, d[:, :, i].mean(), 100 * d[:, :, i].mean() / tot.mean()))
print('  %-20s %9.1f' % ('total', tot.mean()))
# workgroup lifetime and gaps: start of WG k+512 vs end of earlier ones is not available; print the spread of wave skew at barrier 1
skew = (s[:, :, 3].max(1).values - s[:, :, 3].min(1).values)
print('  wave skew at barrier 1: mean %.1f max %.1f ticks' % (skew.mean(), skew.max()))
