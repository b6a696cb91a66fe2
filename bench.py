#!/usr/bin/env python3
"""joint_train step throughput on N MI355X (BASELINE.json metric), one process per GPU.

    python bench.py --gpus 1 --steps 10 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W
    python bench.py --gpus N ...        # N > 1 without WORLD_SIZE in the environment: spawns the line above itself
                                        # (fresh child processes, started before this process touches a GPU)

A "step" is one full joint_train.py:156-212 iteration (enhancer -> fbank -> shared E2E (+CTC,
+location-attention decoder) -> CORAL -> discriminator; G-step and D-step; backward, clip, Adadelta)
on one synthetic AISHELL-shaped batch (config 4: B=32 per GPU, T=800, F=257->80, L=40, V=4233)
that is already resident in HBM.  Weak scaling: every rank processes its own B=32 batch and the
flat gradient buffers are averaged with RCCL.  Rank 0 prints ONE JSON line.

Extra objects on the line:
  roofline     -- the dominant kernel (fp32-MFMA 3x3 convolution at the VGG conv1_2 shape of this
                  workload) timed live with HIP events on the launch stream
  cpu_baseline -- the CPU oracle (oracle/joint.py, "port") timed on this box's host cores on a
                  bounded sample of the same workload (rank 0, N=1 only): 1 warm-up + 3 timed steps, median
  parity       -- the FIRST GPU step against the oracle's step on the SAME full batch, initial weights and cmvn
                  (relative errors of the losses, the clipped-gradient norm and enhance_out); the bench FAILS
                  if any exceeds 1e-3 (north_star's fp32 bar, SURVEY 8d "parity gate in the same run")
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
    """Average launch duration of the dominant kernel -- the 3x3 convolution at the VGG conv1_2 shape of this workload
    (2B=64 images, 800x80, 64->64; csrc/conv3x3.hip behind re2e_conv_igemm) -- measured with HIP events on the stream
    the kernel is launched on.  Algorithmic FLOPs = 2*9*64*64 per output pixel; algorithmic bytes = input + output +
    weights (2.10 GB)."""
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
        pj = json.load(open(os.path.join(ROOT, 'profiles', 'r02_conv1_2_pmc_traffic.json')))
        traffic, tsrc = pj['traffic_bytes_per_launch'], 'profiles/r02_conv1_2_pmc_traffic.json'
    except Exception:
        pass
    return {'bound': 'mfma', 'kernel': 'conv3x3_halo_kernel<16,16,1,true> (VGG conv1_2 fwd, 64x800x80, 64->64, 3x3)', 'achieved': round(ach, 2),
            'peak': PEAK_FP32_MFMA_TFLOPS, 'unit': 'TFLOP/s', 'frac': round(ach / PEAK_FP32_MFMA_TFLOPS, 4), 'traffic': traffic,
            'traffic_unit': 'bytes per launch (FETCH_SIZE + WRITE_SIZE)', 'traffic_source': tsrc,
            'algorithmic_bytes_per_launch': 4.0 * (N * H * W * C + N * H * W * K + K * 9 * C),
            'avg_launch_ms': round(sec * 1e3, 4), 'algorithmic_flop_per_launch': flops}


PARITY_TOL = 1e-3
PARITY_KEYS = ('loss', 'loss_ctc', 'loss_att', 'enhance_loss', 'coral_loss', 'gan_loss', 'loss_D')


def _oracle_state(opt, sd, fbank_W):
    from oracle import joint as oj
    cfg = dict(enhance_layers=opt.enhance_layers, elayers=opt.elayers, mtlalpha=opt.mtlalpha, enhance_loss_lambda=opt.enhance_loss_lambda,
               coral_loss_lambda=opt.coral_loss_lambda, gan_loss_lambda=opt.gan_loss_lambda, grad_clip=opt.grad_clip, eps=opt.eps, isGAN=True,
               enhance_loss_type='L2')
    return oj.JointState(sd[0], sd[1], sd[2], fbank_W, cfg)


def cpu_baseline_and_parity(opt, sd0, fbank_W, batch, cmvn, gpu_first, sample_b=8):
    """The oracle ('port') on the host cores.

    (1) parity: ONE joint step on the full batch of the GPU leg -- same synthetic batch, same initial weights (``sd0`` =
        the GPU nets' state_dicts before their first step), same cmvn -- compared with the GPU's first step.
    (2) cpu_baseline: a bounded sample of the same workload (the first ``sample_b`` utterances of that batch: same T, L, V
        and architecture), 1 warm-up + 3 timed steps on identical inputs, median.  The full-batch step of (1) is timed
        too and reported next to it."""
    from oracle import joint as oj
    cores = host_cores()
    torch.set_num_threads(cores)
    log('cpu_baseline: %d threads' % cores)
    clean, mix, mix_log, targets, il, tl = batch
    B = clean.shape[0]
    t0 = time.time()
    ref = oj.joint_step(_oracle_state(opt, sd0, fbank_W), (clean, mix, mix_log, targets, il.tolist(), tl.tolist()), cmvn)
    full_s = time.time() - t0
    log('cpu oracle: full B=%d step took %.1fs' % (B, full_s))
    par = {}
    for k in PARITY_KEYS:
        a, b = gpu_first['train/' + k], float(ref[k])
        par[k] = abs(a - b) / max(abs(b), 1e-12)
    par['grad_norm'] = abs(gpu_first['grad_norm'] - ref['grad_norm_asr']) / ref['grad_norm_asr']
    eo = gpu_first['enhance_out']
    par['enhance_out_max'] = float((eo - ref['enhance_out']).abs().max() / ref['enhance_out'].abs().max())
    par = {k: float('%.3e' % v) for k, v in par.items()}
    parity = {'tolerance': PARITY_TOL, 'rel_err': par, 'max_rel_err': max(par.values()), 'ok': all(v <= PARITY_TOL for v in par.values()),
              'what': 'GPU step 1 vs oracle/joint.py joint_step on the same B=%d batch, initial weights and cmvn; |a-b|/|b| for the losses and '
                      'the ASR grad norm, max|d|/max|ref| for enhance_out' % B,
              'gpu': {k: gpu_first['train/' + k] for k in PARITY_KEYS}, 'oracle': {k: float(ref[k]) for k in PARITY_KEYS}}
    del ref
    # (2) bounded timing sample
    sb = min(sample_b, B)
    L = int(tl[0])
    sub = (clean[:sb], mix[:sb], mix_log[:sb], targets[:sb * L], il[:sb].tolist(), tl[:sb].tolist())
    times = []
    for i in range(4):
        st = _oracle_state(opt, sd0, fbank_W)            # identical inputs AND weights every repetition
        t0 = time.time()
        oj.joint_step(st, sub, cmvn)
        times.append(time.time() - t0)
        log('cpu_baseline: B=%d sample step %d took %.1fs%s' % (sb, i, times[-1], ' (warm-up)' if i == 0 else ''))
    timed = sorted(times[1:])
    med = timed[len(timed) // 2]
    base = {'value': round(sb / med, 4), 'unit': 'utterances/s', 'cores': cores, 'kind': 'port',
            'sample': 'oracle/joint.py joint_step on the first %d utterances of the GPU leg\'s batch (T=%d, L=%d, V=%d, config-4 architecture, '
                      'same initial weights and cmvn): 1 warm-up + 3 timed steps, median; torch CPU fp32, %d threads'
                      % (sb, clean.shape[1], L, opt.odim, cores),
            'seconds_timed': [round(t, 2) for t in times[1:]], 'seconds_warmup': round(times[0], 2),
            'full_batch': {'utterances': B, 'seconds': round(full_s, 2), 'value': round(B / full_s, 4),
                           'note': 'the single full-batch oracle step of the parity gate (un-repeated)'}}
    return base, parity


def spawn_ranks(n, argv):
    """``bench.py --gpus N`` (N > 1) started as ONE process: run the N ranks as fresh children under torch.distributed.run
    (this process has not touched the GPU: it only imported torch) and exit with their code."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    env['RE2E_BENCH_SPAWNED'] = '1'
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(n), '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.abspath(__file__)] + argv
    log('spawning %d ranks: %s' % (n, ' '.join(cmd)))
    return subprocess.call(cmd, env=env)


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

    if a.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        sys.exit(spawn_ranks(a.gpus, sys.argv[1:]))

    from robust_e2e_gan_amd import dist as rdist
    from robust_e2e_gan_amd import lib
    from robust_e2e_gan_amd.data.synthetic import make_batch
    from robust_e2e_gan_amd.joint_train import JointTrainer, config4_opt
    rank, world, local = rdist.init_from_env()
    assert world == a.gpus, '--gpus %d but WORLD_SIZE=%d' % (a.gpus, world)
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
    want_cpu = world == 1 and not a.no_cpu_baseline
    sd0 = [{k: v.detach().cpu().clone() for k, v in m.state_dict().items()} for m in (enh, asr, gan)] if want_cpu else None

    log('warm-up (%d steps)' % a.warmup)
    gpu_first, host_ms = None, None
    nwarm = max(a.warmup, 1 if want_cpu else 0)        # the parity gate needs the first step un-timed
    for i in range(nwarm):
        torch.cuda.synchronize()
        h0 = time.perf_counter()
        out = tr.step(data, 0.0, cmvn_d)
        host_ms = (time.perf_counter() - h0) * 1e3       # enqueue time of ONE step issued into an idle GPU (no back-pressure)
        torch.cuda.synchronize()
        if i == 0 and want_cpu:                          # the step the parity gate compares: first update from the initial weights
            gpu_first = JointTrainer.to_floats(out)
            gpu_first['enhance_out'] = tr.last['enhance_out'].detach().cpu()
        log('warm-up step %d done (host enqueue %.1f ms)' % (i, host_ms))
    if world > 1:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        out = tr.step(data, 0.0, cmvn_d)
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([dt], device=dev)
        torch.distributed.all_reduce(tmax, op=torch.distributed.ReduceOp.MAX)
        dt = float(tmax.item())
    log('timed region done: %.3fs for %d steps' % (dt, a.steps))
    losses = JointTrainer.to_floats(out)
    from robust_e2e_gan_amd import lib as re2e_lib
    aborts = re2e_lib.query('re2e_lstm_abort_count')
    if aborts != 0:        # a persistent recurrence gave up on a peer workgroup: its outputs are NaN, the numbers mean nothing
        raise SystemExit('bench: %d recurrent sequences were aborted by a persistent kernel (rank %d)' % (aborts, rank))
    if not all(v == v and abs(v) != float('inf') for v in losses.values()):
        raise SystemExit('bench: non-finite losses after the timed region: %r' % (losses,))
    replicas_identical = None
    rccl_ranks = torch.distributed.get_world_size() if (world > 1 and torch.distributed.is_initialized()) else 1
    if world > 1:
        # replicas must still be identical after the timed steps: the all-reduced gradients and the shared NaN gate give
        # every rank the same update (D's BatchNorm running statistics are per replica and not part of this check)
        chk = torch.stack([tr.asr_optimizer.flat.double().sum(), tr.enhance_optimizer.flat.double().sum(), tr.gan_optimizer.flat.double().sum()])
        lo, hi = chk.clone(), chk.clone()
        torch.distributed.all_reduce(lo, op=torch.distributed.ReduceOp.MIN)
        torch.distributed.all_reduce(hi, op=torch.distributed.ReduceOp.MAX)
        replicas_identical = bool(torch.equal(lo, hi))
        if not replicas_identical:
            log('WARNING: replicas differ after %d data-parallel steps: %r vs %r' % (a.steps, lo.tolist(), hi.tolist()))
    if rank != 0:
        return
    value = B * world * a.steps / dt
    line = {
        'metric': 'joint_train utterances/sec', 'value': round(value, 3), 'unit': 'utterances/s', 'n_gpus': world, 'steps': a.steps,
        'warmup': nwarm, 'ms_per_step': round(dt / a.steps * 1e3, 3), 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
        'dtype': 'f32', 'data': 'synthetic',
        'config': {'workload': 'config4: joint_train.py full GAN+ASR step, B=%d per GPU, T=%d, F=257->80, L=%d, V=4233, enhancer 2xBLSTM-256, '
                               'VGG+3xBLSTMP-512, loc-attention decoder 300, D basic ndf64, Adadelta' % (B, T, L),
                   'global_batch': B * world, 'parallelism': 'dp%d' % world, 'coral_loss_lambda': opt.coral_loss_lambda},
        'step_mfma_frac': round(value * FLOP_PER_UTT['config4'] / (world * PEAK_FP32_MFMA_TFLOPS * 1e12), 4) if (B, T, L) == (32, 800, 40) else None,
        'final_losses': {k: round(v, 5) for k, v in losses.items()}, 'persistent_kernel_aborts': aborts,
        'rccl_ranks': rccl_ranks, 'replicas_identical': replicas_identical, 'self_spawned': os.environ.get('RE2E_BENCH_SPAWNED') == '1',
        'host_enqueue_ms_per_step': round(host_ms, 2) if host_ms is not None else None,
    }
    if not a.no_roofline:
        line['roofline'] = conv_roofline(dev)
    if want_cpu:
        from robust_e2e_gan_amd.model.feat_model import mel_matrix
        line['cpu_baseline'], line['parity'] = cpu_baseline_and_parity(opt, sd0, torch.from_numpy(mel_matrix()), batch, cmvn, gpu_first)
        print(json.dumps(line))
        if not line['parity']['ok']:
            raise SystemExit('bench: parity gate FAILED (tolerance %g): %r' % (PARITY_TOL, line['parity']['rel_err']))
        return
    print(json.dumps(line))


if __name__ == '__main__':
    main()
