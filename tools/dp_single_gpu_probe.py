#!/usr/bin/env python3
"""What does the RCCL communication stream cost on ONE GPU?  Runs the config-4 step with a 1-rank NCCL process group and
the data-parallel code path forced on (all-reduces of the three flat gradient buffers really execute, over one rank), so
that the extra HIP stream RCCL brings -- a process has a limited number of hardware queues -- shows up in ms/step.
Multi-GPU scaling itself can only be measured by the driver; this probes the single-process side of it."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
os.environ.setdefault('MASTER_PORT', '29533')
import torch
import torch.distributed as dist
import bench


def main():
    force = os.environ.get('PROBE_DP', '1') == '1'
    from robust_e2e_gan_amd import dist as rdist
    from robust_e2e_gan_amd.data.synthetic import make_batch
    from robust_e2e_gan_amd.joint_train import JointTrainer, config4_opt
    dev = torch.device('cuda:0')
    torch.cuda.set_device(0)
    if force:
        dist.init_process_group('nccl', rank=0, world_size=1)
        rdist.world_size = lambda: 2            # take the DP branches; the group itself has one rank
    opt = config4_opt()
    enh, fb, asr, gan = bench.build(opt, dev)
    clean, mix, mix_log, targets, il, tl = make_batch(32, 800, 40, opt.odim, seed=1234)
    cmvn = bench.synthetic_cmvn(enh, fb, [make_batch(32, 800, 40, opt.odim, seed=77 + i) for i in range(2)], dev).to(dev)
    tr = JointTrainer(opt, enh, fb, asr, gan)
    data = (None, None, clean.to(dev), None, mix.to(dev), mix_log.to(dev), None, targets, il, tl)
    for _ in range(3):
        tr.step(data, 0.0, cmvn)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        tr.step(data, 0.0, cmvn)
    torch.cuda.synchronize()
    out = JointTrainer.to_floats(tr.step(data, 0.0, cmvn))
    print('dp_forced=%d GPU_MAX_HW_QUEUES=%s : %.2f ms/step' % (force, os.environ.get('GPU_MAX_HW_QUEUES', 'default'), (time.perf_counter() - t0) * 100))
    print('losses after 14 steps:', ' '.join('%s=%.9g' % (k.split('/')[-1], v) for k, v in sorted(out.items())))
    if force:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
