#!/usr/bin/env python3
"""s_memtime stamps of the pipelined Winograd kernel (experiments build): cycles between marks of the third work item of every workgroup.
RE2E_LIB=.../libre2e_hip_exp.so RE2E_EXPERIMENTS=1 python tools/wino_pipe_stamps.py [C K H W]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

dev = 'cuda:0'
st = torch.zeros(4096 * 4 * 8, dtype=torch.int64, device=dev)
os.environ['RE2E_WINO_STAMPS'] = '%x' % st.data_ptr()
from robust_e2e_gan_amd import ops   # noqa: E402

C, K, H, W = [int(v) for v in sys.argv[1:5]] if len(sys.argv) > 4 else (64, 64, 800, 80)
N = 64
x = torch.randn(N, H, W, C, device=dev)
Wt = torch.randn(K, C, 3, 3, device=dev) * 0.04
b = torch.zeros(K, device=dev)
for _ in range(3):
    ops.conv3x3_wino(x, Wt, K, bias=b, relu=True, pool=True)
torch.cuda.synchronize()
s = st.view(4096, 4, 8).cpu().double()
s = s[s[:, 0, 0] > 0]
print('workgroups stamped:', s.shape[0])
order = [0, 4, 1, 2, 3]
names = ['output stage of the previous item + chunk 0', 'chunks 1 .. last', 'barrier', 'R calc + LDS write']
tot = s[:, :, 3] - s[:, :, 0]
for i, nme in enumerate(names):
    d = s[:, :, order[i + 1]] - s[:, :, order[i]]
    print('  %-20s mean %9.1f  min %9.1f  max %9.1f' % (nme, d.mean(), d.min(), d.max()))
print('  %-20s mean %9.1f  min %9.1f  max %9.1f' % ('item', tot.mean(), tot.min(), tot.max()))
