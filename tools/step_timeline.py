#!/usr/bin/env python3
"""Host-vs-GPU timeline of one config-4 training step (no profiler attached).

For every phase boundary of JointTrainer._step prints when the host finished ENQUEUEING the phase and when the
GPU (main stream) finished EXECUTING it.  lag = gpu - host: a lag near zero means the GPU was waiting for the
host in that phase (launch-bound); a large lag means the host is ahead and the GPU is the limit."""
import os
import sys

os.environ['RE2E_TIMELINE'] = '1'
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench


def main():
    from robust_e2e_gan_amd.data.synthetic import make_batch
    from robust_e2e_gan_amd.joint_train import JointTrainer, config4_opt
    dev = torch.device('cuda:0')
    opt = config4_opt()
    enh, fb, asr, gan = bench.build(opt, dev)
    B, T, L = 32, 800, 40
    clean, mix, mix_log, targets, il, tl = make_batch(B, T, L, opt.odim, seed=1234)
    cmvn = bench.synthetic_cmvn(enh, fb, [make_batch(B, T, L, opt.odim, seed=77 + i) for i in range(2)], dev).to(dev)
    tr = JointTrainer(opt, enh, fb, asr, gan)
    data = (None, None, clean.to(dev), None, mix.to(dev), mix_log.to(dev), None, targets, il, tl)
    rows = []
    for i in range(5):
        tr.step(data, 0.0, cmvn)
        rows = tr.timeline()
    print('%-40s %10s %10s %10s' % ('phase end', 'host ms', 'gpu ms', 'lag ms'))
    for label, h, g in rows:
        print('%-40s %10.2f %10.2f %10.2f' % (label, h, g, g - h))


if __name__ == '__main__':
    main()
