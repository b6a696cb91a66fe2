"""CPU-only, world_size 2 over gloo: the data-parallel gradient exchange (dist.py)."""
import os

import torch
import torch.multiprocessing as mp


class _Opt:
    def __init__(self, g):
        self.grad = g


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    from robust_e2e_gan_amd import dist as rdist
    r, w, _ = rdist.init_from_env()
    assert (r, w) == (rank, world)
    # (a) flat all-reduce == mean of per-rank buffers
    g = torch.arange(10, dtype=torch.float32) * (rank + 1)
    rdist.allreduce_mean_(g)
    # (b) GradSync: early buffer via autograd hook + late buffer
    early, late = _Opt(torch.full((5,), float(rank + 1))), _Opt(torch.full((3,), float(10 * (rank + 1))))
    x = torch.ones(4, requires_grad=True)
    y = x * 2
    sync = rdist.GradSync()
    sync.arm(y, early)
    (y * 3).sum().backward()
    sync.finish([late])
    # (c) a failure seen by ONE replica is seen by all (fit's abort check, compute_cmvn_epoch): max over ranks
    flags = (rdist.any_rank(3 if rank == 1 else 0), rdist.any_rank(0))
    q.put((rank, g.tolist(), early.grad.tolist(), late.grad.tolist(), rdist.shard_indices(7, rank, world), flags))
    torch.distributed.destroy_process_group()


def test_allreduce_mean_world2():
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 29650 + os.getpid() % 200
    ps = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in ps:
        p.join(60)
        assert p.exitcode == 0
    for rank, g, early, late, shard, flags in res:
        assert flags == (3, 0)
        assert g == [1.5 * i for i in range(10)]
        assert early == [1.5] * 5 and late == [15.0] * 3
        assert shard == list(range(rank, 7, 2))


def test_world_size_one_needs_no_collective():
    from robust_e2e_gan_amd import dist as rdist
    g = torch.ones(4)
    assert rdist.allreduce_mean_(g) is None and g.tolist() == [1.0] * 4
    s = rdist.GradSync()
    x = torch.ones(2, requires_grad=True)
    s.arm(x * 1.0, _Opt(g))
    s.finish([_Opt(g)])
    assert rdist.any_rank(2) == 2


def _dp_worker(rank, world, port, q, golden):
    """the data-parallel result is the MEAN of the per-shard gradients, and the clip norm is taken on the
    reduced gradients.  Each rank runs the oracle's joint step on its utterance shard (r::N of the length-sorted batch),
    puts the gradients into the product's flat buffers (optim.FlatOptimizer) and exchanges them with dist.GradSync."""
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    import numpy as np
    from oracle import joint as oj
    from robust_e2e_gan_amd import dist as rdist
    from robust_e2e_gan_amd.optim import FlatOptimizer
    rdist.init_from_env()
    fx = dict(np.load(os.path.join(golden, 'joint_tiny.npz')))
    W = torch.from_numpy(dict(np.load(os.path.join(golden, 'fbank_tiny.npz')))['W'])
    cfg = dict(enhance_layers=2, elayers=2, mtlalpha=0.5, enhance_loss_lambda=1.0, coral_loss_lambda=0.5, gan_loss_lambda=1.0, grad_clip=5.0,
               eps=1e-8, isGAN=False, enhance_loss_type='L2')
    sub = lambda pre: {k[len(pre):]: torch.from_numpy(v) for k, v in fx.items() if k.startswith(pre)}
    lens, tls = fx['lens'].tolist(), fx['tlens'].tolist()
    offs = np.concatenate([[0], np.cumsum(tls)])

    def shard_grads(r):
        idx = rdist.shard_indices(len(lens), r, world)
        T = max(lens[i] for i in idx)
        t = lambda k: torch.from_numpy(fx[k])[idx, :T]
        tg = torch.cat([torch.from_numpy(fx['targets'])[offs[i]:offs[i + 1]] for i in idx])
        st = oj.JointState(sub('enh.'), sub('asr.'), sub('gan.'), W, cfg)
        out = oj.joint_step(st, (t('clean'), t('mix'), t('mix_log'), tg, [lens[i] for i in idx], [tls[i] for i in idx]),
                            torch.from_numpy(fx['cmvn']), update=False)
        return out['g_asr'], out['g_enh']
    mine = shard_grads(rank)
    opts = []
    for gd in mine:
        params = [torch.nn.Parameter(torch.zeros_like(v)) for v in gd.values()]
        o = FlatOptimizer(params)
        for p, v in zip(params, gd.values()):
            p.grad.copy_(v)                       # p.grad is a view of the flat buffer
        opts.append((o, params))
    x = torch.ones(2, requires_grad=True)
    y = x * 1.0
    sync = rdist.GradSync()
    sync.arm(y, opts[0][0])                       # ASR buffer through the early (hook) path, the enhancer's at the end
    y.sum().backward()
    sync.finish([opts[1][0]])
    both = [shard_grads(r) for r in range(world)]
    worst = 0.0
    for n, (o, params) in enumerate(opts):
        for p, k in zip(params, mine[n].keys()):
            want = sum(both[r][n][k] for r in range(world)) / world
            worst = max(worst, float((p.grad - want).abs().max() / (want.abs().max() + 1e-12)))
    # clip on the REDUCED gradients: norm of the mean, not the mean of the norms
    norm_reduced = oj.clip_grad_norm([p.grad.clone() for p in opts[0][1]], 5.0)
    norm_want = oj.clip_grad_norm([sum(both[r][0][k] for r in range(world)) / world for k in mine[0]], 5.0)
    norm_own = oj.clip_grad_norm([v.clone() for v in mine[0].values()], 5.0)
    q.put((rank, worst, norm_reduced, norm_want, norm_own))
    torch.distributed.destroy_process_group()


def test_dp_is_mean_of_shard_gradients_world2():
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 29850 + os.getpid() % 100
    ps = [ctx.Process(target=_dp_worker, args=(r, 2, port, q, here)) for r in range(2)]
    for p in ps:
        p.start()
    res = sorted(q.get(timeout=300) for _ in range(2))
    for p in ps:
        p.join(60)
        assert p.exitcode == 0
    for rank, worst, nr, nw, no in res:
        assert worst < 1e-6, (rank, worst)
        assert abs(nr - nw) < 1e-5 * nw
    assert abs(res[0][2] - res[1][2]) < 1e-6 * res[0][2]            # both replicas clip with the same norm
    assert abs(res[0][4] - res[1][4]) > 1e-3 * res[0][4]            # ... although their own shards' norms differ


def _sync_rows_worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    from robust_e2e_gan_amd import dist as rdist
    from robust_e2e_gan_amd import lib, ops
    rdist.init_from_env()
    out = {}
    t = rdist.allreduce_max_(torch.tensor([float(rank), -float(rank)]))
    out['max'] = t.tolist()
    dev = torch.device('cpu')
    z = ops._sync_rows_poison(96, dev)                                      # equal rows on both ranks: no poison, nothing counted
    out['equal_ok'] = float(z) == 0.0 and sum(float(c) for c in ops.sync_bn_flags()) == 0.0
    ops.check_sync_bn()
    # ragged shards: BOTH ranks get the NaN and count the call -- the decision comes from the reduced pair alone, not from what a rank has
    # seen before (round 4: a rank that knew its own shape skipped the check and its peer raised alone) ...
    z = ops._sync_rows_poison(96 + 16 * rank, dev)
    out['ragged'] = 'poisoned' if z != z else 'clean'
    # ... exact where fp32 sums of rows^2 were not: 3 utterances x 3069 rows = 9207 rows (odd part 9207 > 4096), equal on both ranks
    z = ops._sync_rows_poison(9207, dev)
    out['big_equal'] = float(z) == 0.0
    z = ops._sync_rows_poison((1 << 30) + rank, dev)                        # and a difference of ONE row in 2^30 is seen
    out['big_ragged'] = 'poisoned' if z != z else 'clean'
    try:
        from robust_e2e_gan_amd.joint_train import JointTrainer as JT
        JT.to_floats({'grad_norm': torch.tensor(1.0)})               # the trainers' one read-back per step carries the counters
        out['raise'] = 'no error'
    except lib.Re2eError as e:
        out['raise'] = 'raised' if '2 BatchNorm calls' in str(e) else str(e)
    out['cleared'] = sum(float(c) for c in ops.sync_bn_flags()) == 0.0

    # the dropout stream of a checkpoint written by rank 0 is re-derived per rank
    from robust_e2e_gan_amd.joint_train import JointTrainer
    JointTrainer.restore_dropout({'dropout_state': {'base_seed': 7, 'call': 5}})
    out['dropout'] = ops.dropout_state()
    q.put((rank, out))
    torch.distributed.destroy_process_group()


def test_sync_bn_row_check_and_per_rank_dropout_stream_world2():
    """Round 4/5 (advisor findings): synchronised BatchNorm refuses ragged shards on every rank in the same step, exactly; a checkpoint's dropout state
    (base seed, mask index) gives each rank ITS stream back."""
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 29850 + os.getpid() % 100
    ps = [ctx.Process(target=_sync_rows_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = dict(q.get(timeout=120) for _ in range(2))
    for p in ps:
        p.join(60)
        assert p.exitcode == 0
    for rank in (0, 1):
        o = res[rank]
        assert o['max'] == [1.0, 0.0] and o['equal_ok'] and o['ragged'] == 'poisoned' and o['big_equal'] and o['big_ragged'] == 'poisoned', o
        assert o['raise'] == 'raised' and o['cleared'], o
        assert o['dropout'] == ((7 * 1000003 + rank) & 0xFFFFFFFFFFFF, 5)
