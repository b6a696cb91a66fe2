#!/usr/bin/env python3
"""Soak of the config-4 training step over 48 batches that all DIFFER (lengths, batch composition, target lengths): what a real epoch does to the
host-side caches (length tensors, row maps, VGG row limits: model/e2e_common._LENS_CACHE, ops._ROW_MAPS), the per-stream workspaces and the
caching allocator.  Prints the step time, the allocator's numbers and the give-up counter every 50 steps; fails on a non-finite loss, a give-up
or allocator growth after the warm-up third.

    python tools/soak_step.py [steps=300] [B=32] [Tmax=800]"""
import math
import os
import random
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench


def batch(rng, B, Tmax, L, V, g):
    lens = sorted((rng.randint(Tmax // 3, Tmax) for _ in range(B)), reverse=True)
    lens[0] = Tmax if rng.random() < 0.5 else lens[0]
    T = lens[0]
    clean = torch.zeros(B, T, 257)
    mix = torch.zeros(B, T, 257)
    mix_log = torch.zeros(B, T, 257)
    for b, l in enumerate(lens):
        c = torch.rand(l, 257, generator=g) * 300.0
        clean[b, :l] = c
        mix[b, :l] = c + torch.rand(l, 257, generator=g) * 100.0
        mix_log[b, :l] = torch.randn(l, 257, generator=g)
    tl = [rng.randint(1, L) for _ in range(B)]
    targets = torch.randint(1, V - 1, (sum(tl),), generator=g)
    return (None, None, clean, None, mix, mix_log, None, targets, torch.IntTensor(lens), torch.IntTensor(tl))


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
    Tmax = int(sys.argv[3]) if len(sys.argv) > 3 else 800
    from robust_e2e_gan_amd import lib
    from robust_e2e_gan_amd.joint_train import JointTrainer, config4_opt
    dev = torch.device('cuda:0')
    opt = config4_opt()
    enh, fb, asr, gan = bench.build(opt, dev)
    tr = JointTrainer(opt, enh, fb, asr, gan)
    cm = torch.stack([torch.linspace(-10.0, -7.0, 80), torch.linspace(0.3, 0.5, 80)]).to(dev)
    rng, g = random.Random(5), torch.Generator().manual_seed(5)
    # 48 distinct batches, some half-size, resident in HBM (generating and uploading them is not what is soaked); each brings ~15 entries to the
    # length cache (256 entries, LRU), so a cycle through the pool evicts and re-creates most of them
    pool = []
    for i in range(48):
        d = batch(rng, B if i % 7 else max(1, B // 2), Tmax, 40, opt.odim, g)
        pool.append(tuple(x.to(dev) if isinstance(x, torch.Tensor) and x.dtype == torch.float32 else x for x in d))
    torch.cuda.synchronize()
    base_alloc = None
    prof = None
    if os.environ.get('SOAK_PROFILE'):
        import cProfile
        prof = cProfile.Profile()
    t_up = t_step = 0.0
    prev = prev_out = None
    t0 = time.time()
    for i in range(steps):
        d = pool[i % len(pool)]
        ta = time.time()
        tb = time.time()
        if prof is not None and i >= 60:
            prof.enable()
        out = tr.step(d, 0.0, cm)
        if prof is not None:
            prof.disable()
        # like JointTrainer.fit: the PREVIOUS step's meters are read once this one is enqueued -- the host never runs more than one step ahead
        # (without it the host gets ~10 steps ahead in this loop and every tensor another stream has been told about -- record_stream -- stays
        # reserved until the GPU catches up: 100 GiB reserved for an 8 GiB live set)
        if prev is not None and not os.environ.get('SOAK_NO_LATE_READ'):
            JointTrainer.finish_read(prev) if not os.environ.get('SOAK_DRAINING_READ') else JointTrainer.to_floats(prev_out)
        prev, prev_out = JointTrainer.start_read({k: v for k, v in out.items() if k.startswith('train/') or k in ('grad_norm', 'aborts')}), out
        t_up, t_step = t_up + tb - ta, t_step + time.time() - tb
        if (i + 1) % 50 == 0 or i + 1 == steps:
            f = JointTrainer.to_floats(out)
            torch.cuda.synchronize()
            al, rs = torch.cuda.memory_allocated() / 2**30, torch.cuda.memory_reserved() / 2**30
            print('   host: step() %.1f ms per step' % (1e3 * t_step / 50))
            t_up = t_step = 0.0
            print('step %4d  %.1f ms/step (wall, batches resident)  loss %.3f  allocated %.2f GiB  reserved %.2f GiB  give-ups %d' %
                  (i + 1, 1e3 * (time.time() - t0) / 50, f['train/loss'], al, rs, int(f.get('aborts', 0))), flush=True)
            t0 = time.time()
            assert all(math.isfinite(v) for v in f.values()), f
            assert int(f.get('aborts', 0)) == 0, f
            if i + 1 == max(50, steps // 3 // 50 * 50):
                base_alloc = rs
            elif base_alloc is not None:
                assert rs <= base_alloc * 1.25 + 1.0, ('reserved memory keeps growing', base_alloc, rs)
    if prof is not None:
        import pstats
        pstats.Stats(prof).sort_stats('cumulative').print_stats(45)
    print('soak ok: %d steps' % steps)


if __name__ == '__main__':
    main()
