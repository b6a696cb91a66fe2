#!/usr/bin/env python3
"""joint_train step throughput on N MI355X (BASELINE.json metric), one process per GPU.

    python bench.py --gpus 1 --steps 10 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" is one full joint_train.py:156-212 iteration (enhancer -> fbank -> shared E2E (+CTC,
+location-attention decoder) -> CORAL -> discriminator; G-step and D-step; backward, clip, Adadelta)
on one synthetic AISHELL-shaped batch (config 4: B=32 per GPU, T=800, F=257->80, L=40, V=4233)
that is already resident in HBM.  Weak scaling: every rank processes its own B=32 batch and the
flat gradient buffers are averaged with RCCL.  Rank 0 prints ONE JSON line.

Extra objects on the line:
  roofline     -- the dominant kernel (fp32-MFMA implicit-GEMM convolution at the VGG conv1_2 shape
                  of this workload) timed live with HIP events on the launch stream
  cpu_baseline -- the CPU oracle (oracle/joint.py, "port") timed on this box's host cores on a
                  bounded sample of the same workload (rank 0, N=1 only)
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402

PEAK_FP32_MFMA_TFLOPS = 157.3          # MI355X_MICROARCH.md, 'Peak FP32 (matrix)'
FLOP_PER_UTT = {'config4': 189.83e9}


def log(msg):
    if int(os.environ.get('RANK', '0')) == 0:
        print('[bench %7.1fs] %s' % (time.time() - _T0, msg), file=sys.stderr, flush=True)


_T0 = time.time()


def host_cores():
    """Usable host cores: scheduler affinity capped by the cgroup CPU quota (os.cpu_count() reports the whole machine)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    try:
        q, p = open('/sys/fs/cgroup/cpu.max').read().split()
        if q != 'max':
            n = min(n, max(1, int(float(q) / float(p))))
    except Exception:
        pass
    return max(1, min(n, 64))


def build(opt, dev):
    from robust_e2e_gan_amd.model.enhance_model import EnhanceModel
    from robust_e2e_gan_amd.model.feat_model import FbankModel
    from robust_e2e_gan_amd.model.e2e_model import ShareE2E
    from robust_e2e_gan_amd.model.gan_model import GANModel
    torch.manual_seed(1234)
    nets = [EnhanceModel(opt), FbankModel(opt), ShareE2E(opt), GANModel(opt)]
    return [m.to(dev).train() for m in nets]


def synthetic_cmvn(enh, fb, batches, dev):
    """F4: CMVN of the enhanced features over a few synthetic batches (feat_model.py:62-90)."""
    fb.cmvn_num = sum(int(b[4].numel()) for b in batches)
    out = None
    with torch.no_grad():
        for b in batches + batches[:1]:
            eo = enh(b[1].to(dev), b[2].to(dev), b[4])
            out = fb.compute_cmvn(eo, b[4])
    assert out is not None
    return torch.FloatTensor(out)


def conv_roofline(dev, iters=20):
    """Average launch duration of the dominant kernel -- the implicit-GEMM conv at the VGG conv1_2
    shape of this workload (2B=64 images, 800x80, 64->64, 3x3) -- measured with HIP events on the
    stream the kernel is launched on.  Algorithmic FLOPs = 2*9*64*64 per output pixel."""
    from robust_e2e_gan_amd import lib
    N, H, W, C, K = 64, 800, 80, 64, 64
    x = torch.randn(N, H, W, C, device=dev)
    wg = torch.randn(K, 3, 3, C, device=dev) * 0.04
    b = torch.zeros(K, device=dev)
    y = torch.empty(N, H, W, K, device=dev)
    args = (x.data_ptr(), N, H, W, C, wg.data_ptr(), K, 3, 3, H, W, 1, 1, 1, 1, -1, -1, y.data_ptr(), H, W, 1, 1, 0, 0, b.data_ptr(),
            lib.ACT_RELU, 0.0)
    for _ in range(3):
        lib.call('re2e_conv_igemm', *args)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(iters):
        lib.call('re2e_conv_igemm', *args)
    e1.record()
    torch.cuda.synchronize()
    sec = e0.elapsed_time(e1) * 1e-3 / iters
    flops = 2.0 * 9 * C * K * N * H * W
    ach = flops / sec / 1e12
    # HBM-side bytes per launch from the committed PMC passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE on this same
    # kernel and shape, tools/roofline_conv.py); a counter pass cannot run inside this process.
    traffic, tsrc = None, None
    try:
        pj = json.load(open(os.path.join(ROOT, 'profiles', 'r01_conv1_2_pmc_traffic.json')))
        traffic, tsrc = pj['traffic_bytes_per_launch'], 'profiles/r01_conv1_2_pmc_traffic.json'
    except Exception:
        pass
    return {'bound': 'mfma', 'kernel': 'igemm_kernel<ConvK,DenseK,256x64> (VGG conv1_2 fwd, 64x800x80, 64->64, 3x3)', 'achieved': round(ach, 2),
            'peak': PEAK_FP32_MFMA_TFLOPS, 'unit': 'TFLOP/s', 'frac': round(ach / PEAK_FP32_MFMA_TFLOPS, 4), 'traffic': traffic,
            'traffic_unit': 'bytes per launch (FETCH_SIZE + WRITE_SIZE)', 'traffic_source': tsrc,
            'algorithmic_bytes_per_launch': 4.0 * (N * H * W * C + N * H * W * K + K * 9 * C),
            'avg_launch_ms': round(sec * 1e3, 4), 'algorithmic_flop_per_launch': flops}


def cpu_baseline(opt):
    """Oracle ('port') on the host cores: one joint step on the full config-4 batch (B=32,
    same T/L/V/architecture) after a small warm-up step."""
    from oracle import joint as oj
    from robust_e2e_gan_amd.data.synthetic import make_batch
    from robust_e2e_gan_amd.model.feat_model import mel_matrix
    cores = host_cores()
    torch.set_num_threads(cores)
    log('cpu_baseline: %d threads' % cores)
    nets = build(opt, 'cpu')
    sd = [m.state_dict() for m in nets]
    cfg = dict(enhance_layers=opt.enhance_layers, elayers=opt.elayers, mtlalpha=opt.mtlalpha, enhance_loss_lambda=opt.enhance_loss_lambda,
               coral_loss_lambda=opt.coral_loss_lambda, gan_loss_lambda=opt.gan_loss_lambda, grad_clip=opt.grad_clip, eps=opt.eps, isGAN=True,
               enhance_loss_type='L2')
    st = oj.JointState(sd[0], sd[2], sd[3], torch.from_numpy(mel_matrix()), cfg)
    cm = torch.stack([torch.full((80,), -8.0), torch.full((80,), 0.5)])
    for B, T, L, timed in ((2, 200, 10, False), (32, 800, 40, True)):
        clean, mix, mix_log, targets, il, tl = make_batch(B, T, L, opt.odim, seed=1234)
        t0 = time.time()
        oj.joint_step(st, (clean, mix, mix_log, targets, il.tolist(), tl.tolist()), cm)
        dt = time.time() - t0
        log('cpu_baseline: B=%d T=%d step took %.1fs' % (B, T, dt))
    return {'value': round(32.0 / dt, 4), 'unit': 'utterances/s', 'cores': cores, 'kind': 'port',
            'sample': 'oracle/joint.py joint_step, config-4 architecture, the full B=32 batch, T=800, L=40, V=4233, 1 timed step after a '
                      'B=2,T=200 warm-up; torch CPU fp32, %d threads' % cores, 'seconds': round(dt, 2)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--batch', type=int, default=32)
    ap.add_argument('--frames', type=int, default=800)
    ap.add_argument('--labels', type=int, default=40)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-roofline', action='store_true')
    a = ap.parse_args()

    from robust_e2e_gan_amd import dist as rdist
    from robust_e2e_gan_amd import lib
    from robust_e2e_gan_amd.data.synthetic import make_batch
    from robust_e2e_gan_amd.joint_train import JointTrainer, config4_opt
    rank, world, local = rdist.init_from_env()
    assert world == a.gpus, 'launch with torch.distributed.run --nproc-per-node %d (WORLD_SIZE=%d)' % (a.gpus, world)
    assert torch.cuda.is_available(), 'bench.py needs a GPU (no CPU fallback)'
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    assert lib.query('re2e_device_ok') == 1, 'not a gfx950 device'

    opt = config4_opt()
    log('building networks')
    enh, fb, asr, gan = build(opt, dev)
    B, T, L = a.batch, a.frames, a.labels
    batch = make_batch(B, T, L, opt.odim, seed=1234 + rank)
    log('synthetic batch ready; computing cmvn')
    cmvn = synthetic_cmvn(enh, fb, [make_batch(B, T, L, opt.odim, seed=77 + i) for i in range(2)], dev)
    tr = JointTrainer(opt, enh, fb, asr, gan)
    clean, mix, mix_log, targets, il, tl = batch
    data = (None, None, clean.to(dev), None, mix.to(dev), mix_log.to(dev), None, targets, il, tl)     # inputs resident in HBM
    cmvn_d = cmvn.to(dev)

    log('warm-up (%d steps)' % a.warmup)
    for i in range(a.warmup):
        out = tr.step(data, 0.0, cmvn_d)
        torch.cuda.synchronize()
        log('warm-up step %d done' % i)
    if world > 1:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    host = 0.0
    for _ in range(a.steps):
        h0 = time.perf_counter()
        out = tr.step(data, 0.0, cmvn_d)
        host += time.perf_counter() - h0
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([dt], device=dev)
        torch.distributed.all_reduce(tmax, op=torch.distributed.ReduceOp.MAX)
        dt = float(tmax.item())
    log('timed region done: %.3fs for %d steps (host enqueue %.1f ms/step)' % (dt, a.steps, host / a.steps * 1e3))
    losses = JointTrainer.to_floats(out)
    from robust_e2e_gan_amd import lib as re2e_lib
    aborts = re2e_lib.query('re2e_lstm_abort_count')
    if aborts != 0:        # a persistent recurrence gave up on a peer workgroup: its outputs are NaN, the numbers mean nothing
        raise SystemExit('bench: %d recurrent sequences were aborted by a persistent kernel (rank %d)' % (aborts, rank))
    if not all(v == v and abs(v) != float('inf') for v in losses.values()):
        raise SystemExit('bench: non-finite losses after the timed region: %r' % (losses,))
    if rank != 0:
        return
    value = B * world * a.steps / dt
    line = {
        'metric': 'joint_train utterances/sec', 'value': round(value, 3), 'unit': 'utterances/s', 'n_gpus': world, 'steps': a.steps,
        'warmup': a.warmup, 'ms_per_step': round(dt / a.steps * 1e3, 3), 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
        'dtype': 'f32', 'data': 'synthetic',
        'config': {'workload': 'config4: joint_train.py full GAN+ASR step, B=%d per GPU, T=%d, F=257->80, L=%d, V=4233, enhancer 2xBLSTM-256, '
                               'VGG+3xBLSTMP-512, loc-attention decoder 300, D basic ndf64, Adadelta' % (B, T, L),
                   'global_batch': B * world, 'parallelism': 'dp%d' % world, 'coral_loss_lambda': opt.coral_loss_lambda},
        'step_mfma_frac': round(value * FLOP_PER_UTT['config4'] / (world * PEAK_FP32_MFMA_TFLOPS * 1e12), 4) if (B, T, L) == (32, 800, 40) else None,
        'final_losses': {k: round(v, 5) for k, v in losses.items()}, 'persistent_kernel_aborts': aborts,
    }
    if not a.no_roofline:
        line['roofline'] = conv_roofline(dev)
    if world == 1 and not a.no_cpu_baseline:
        line['cpu_baseline'] = cpu_baseline(opt)
    print(json.dumps(line))


if __name__ == '__main__':
    main()
