"""CPU-only tests of the host-side mirror of the reference interface: collate, options, loop
utilities, state_dict contract."""
import os

import numpy as np
import torch


def _fx(golden_dir, name):
    return dict(np.load(os.path.join(golden_dir, name)))


def test_collate_matches_reference_output(golden_dir):
    """F1: ``_collate_fn`` against the OUTPUT of the reference's own function (tests/golden/make_fixtures_collate.py imports
    data/mix_data_loader.py:264-302): a tie in lengths (stable order), an empty target list, a one-frame sample."""
    from robust_e2e_gan_amd.data.mix_data_loader import _collate_fn
    fx = _fx(golden_dir, 'collate_tiny.npz')
    batch = []
    for i in range(int(fx['n'])):
        s = [torch.from_numpy(fx['s%d_%d' % (i, k)]) for k in range(5)]
        batch.append(('utt%d' % i, 'spk%d' % (i % 2), s[0], s[1], s[2], s[3], s[4], fx['t%d' % i].tolist()))
    out = _collate_fn(batch)
    assert out[0] == ['utt%d' % i for i in fx['order']]
    assert out[1] == ['spk%d' % i for i in fx['spk']]
    for k in range(5):
        assert out[2 + k].dtype == torch.float32 and np.array_equal(out[2 + k].numpy(), fx['expected'][k])
    assert out[8].dtype == torch.int32 and np.array_equal(out[8].numpy(), fx['input_sizes'])
    assert out[9].dtype == torch.int32 and np.array_equal(out[9].numpy(), fx['target_sizes'])
    assert out[7].dtype == torch.int64 and np.array_equal(out[7].numpy(), fx['targets'])
    assert 0 in fx['target_sizes'].tolist() and len(set(fx['input_sizes'].tolist())) < int(fx['n'])      # the edge cases are in


def test_asr_collate_matches_reference_output(golden_dir):
    """the ASR twin, data/data_loader.py:236-265"""
    from robust_e2e_gan_amd.data.data_loader import _collate_fn
    fx = _fx(golden_dir, 'collate_tiny.npz')
    batch = [('a%d' % i, 's', torch.from_numpy(fx['asr.s%d_0' % i]), torch.from_numpy(fx['asr.s%d_1' % i]), fx['asr.t%d' % i].tolist())
             for i in range(int(fx['asr.n']))]
    out = _collate_fn(batch)
    assert out[0] == ['a%d' % i for i in fx['asr.order']]
    assert np.array_equal(out[2].numpy(), fx['asr.expected'][0]) and np.array_equal(out[3].numpy(), fx['asr.expected'][1])
    assert out[4].dtype == torch.int64 and np.array_equal(out[4].numpy(), fx['asr.targets'])
    assert out[5].dtype == torch.int32 and np.array_equal(out[5].numpy(), fx['asr.input_sizes'])
    assert np.array_equal(out[6].numpy(), fx['asr.target_sizes'])


def test_options_surface():
    from robust_e2e_gan_amd.options.train_options import TrainOptions
    o = TrainOptions().parse(['--isGAN', '--fbank_dim', '80', '--gpu_ids', '-1', '--mtlalpha', '1.0', '--enhace_resume', 'x.pth'], save=False)
    assert o.gpu_ids == [] and o.mtl_mode == 'ctc' and o.enhance_resume == 'x.pth'
    for attr in ('aconv_chans', 'aconv_filts', 'fbank_opti_type', 'subsample_type', 'lsm_type', 'dropout_rate', 'grad_clip', 'eps_decay',
                 'sche_samp_final_epoch', 'enhance_loss_lambda', 'coral_loss_lambda', 'sche_samp_start_iter', 'sche_samp_final_iter',
                 'batch_size', 'validate_freq', 'print_freq', 'num_save_attention'):
        assert hasattr(o, attr), attr
    assert o.coral_loss_lambda == 0.0 and o.enhance_loss_lambda == 1.0 and o.isGAN is True


def test_loop_utilities():
    from robust_e2e_gan_amd.utils.utils import ScheSampleRampup, adadelta_eps_decay
    r = ScheSampleRampup(5, 15, 0.6)
    assert r.update(3) == 0.0 and abs(r.update(10) - 0.3) < 1e-12 and r.update(20) == 0.6

    class O:
        param_groups = [{'eps': 1e-8}, {'eps': 1e-8}]
    assert abs(adadelta_eps_decay(O, 0.01) - 1e-10) < 1e-20 and O.param_groups[1]['eps'] == 1e-8   # group 0 only


def test_state_dict_names_match_reference(golden_dir):
    import __graft_entry__ as g
    from robust_e2e_gan_amd.model.enhance_model import EnhanceModel
    from robust_e2e_gan_amd.model.e2e_model import ShareE2E, E2E
    from robust_e2e_gan_amd.model.gan_model import GANModel
    from robust_e2e_gan_amd.model.feat_model import FbankModel
    fx = _fx(golden_dir, 'joint_tiny.npz')
    opt = g._tiny_opt()
    for cls, pre in ((EnhanceModel, 'enh.'), (ShareE2E, 'asr.'), (E2E, 'asr.'), (GANModel, 'gan.')):
        want = {k[len(pre):]: tuple(v.shape) for k, v in fx.items() if k.startswith(pre)}
        m = cls(opt)
        have = {k: tuple(v.shape) for k, v in m.state_dict().items()}
        assert want == have, (pre, sorted(set(want) ^ set(have)))
        m.load_state_dict({k[len(pre):]: torch.from_numpy(v) for k, v in fx.items() if k.startswith(pre)})
    assert list(FbankModel(opt).state_dict().keys()) == ['fc']


def test_init_statistics():
    """Appendix A.15: LeCun normal by rank, embed N(0,1), decoder forget bias 1, D N(0,0.02)."""
    from robust_e2e_gan_amd.joint_train import config4_opt
    from robust_e2e_gan_amd.model.e2e_model import E2E
    from robust_e2e_gan_amd.model.gan_model import GANModel
    torch.manual_seed(0)
    opt = config4_opt(odim=50, char_list=[str(i) for i in range(50)], eunits=64, eprojs=64, elayers=1)
    m = E2E(opt)
    sd = m.state_dict()
    assert abs(sd['enc.enc1.conv1_2.weight'].std().item() - 1 / np.sqrt(64 * 9)) < 3e-3
    assert abs(sd['dec.embed.weight'].std().item() - 1.0) < 0.05
    b = sd['dec.decoder.0.bias_ih']
    n = b.numel()
    assert (b[n // 4:n // 2] == 1).all() and (b[:n // 4] == 0).all()
    assert sd['att.mlp_enc.weight'].data_ptr() == sd['dec.att.mlp_enc.weight'].data_ptr()      # shared module
    g = GANModel(opt).state_dict()
    assert abs(g['model.2.weight'].std().item() - 0.02) < 2e-3 and abs(g['model.3.weight'].mean().item() - 1.0) < 0.01
    assert (g['model.0.bias'] == 0).all()
    counts = sum(p.numel() for p in E2E(config4_opt()).parameters())
    assert counts == 29147557, counts                                                            # Appendix B


def test_synthetic_batch_shapes():
    from robust_e2e_gan_amd.data.synthetic import make_batch, lengths
    clean, mix, mix_log, targets, ilens, tlens = make_batch(B=4, Tmax=50, L=5, V=30)
    assert ilens.tolist() == lengths(4, 50) == [50, 45, 40, 35]
    assert clean.shape == (4, 50, 257) and (clean[3, 35:] == 0).all() and (mix_log[3, 35:] == 0).all()
    assert targets.min() >= 1 and targets.max() <= 28 and targets.numel() == 20


def test_bench_strong_scaling_shards_cover_the_global_batch():
    """bench.py --scaling strong: rank r keeps utterances r::N of the SAME global batch; the shards partition it (every utterance and
    its labels exactly once) and keep the length-sorted order."""
    import importlib
    import sys
    import os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    bench = importlib.import_module('bench')
    from robust_e2e_gan_amd.data.synthetic import make_batch
    from robust_e2e_gan_amd.dist import shard_indices
    B, T, L, V = 8, 40, 5, 50
    batch = make_batch(B, T, L, V, seed=3)
    seen, rows = [], 0
    for r in range(4):
        idx = shard_indices(B, r, 4)
        sh = bench.shard_batch(batch, idx, L)
        assert sh[0].shape[0] == len(idx) == 2 and sh[3].numel() == len(idx) * L
        for j, i in enumerate(idx):
            assert torch.equal(sh[0][j], batch[0][i]) and torch.equal(sh[1][j], batch[1][i]) and int(sh[4][j]) == int(batch[4][i])
            assert torch.equal(sh[3][j * L:(j + 1) * L], batch[3][i * L:(i + 1) * L])
        assert sh[4].tolist() == sorted(sh[4].tolist(), reverse=True)
        seen += idx
        rows += sh[0].shape[0]
    assert sorted(seen) == list(range(B)) and rows == B
    assert set(bench.FLOP_PER_UTT) == set(bench.CONFIG_SHAPES) == {2, 3, 4, 5}


def test_dropout_stream_is_checkpointed():
    """JointTrainer.state() carries (seed, next mask index) of the dropout stream; restore_dropout continues it."""
    from robust_e2e_gan_amd import ops
    from robust_e2e_gan_amd.joint_train import JointTrainer
    ops.dropout_seed(99, 17)
    pkg = {'dropout_state': ops.dropout_state()}
    ops.dropout_seed(1, 0)
    JointTrainer.restore_dropout(pkg)
    assert ops.dropout_state() == (99, 17)
    JointTrainer.restore_dropout({})                   # checkpoints without the key: untouched
    assert ops.dropout_state() == (99, 17)
    # what state() writes since round 4: the BASE seed and the mask index; every rank re-derives ITS stream from them (a checkpoint is
    # written by rank 0: restoring its own stream seed would give every replica rank 0's masks)
    from robust_e2e_gan_amd import dist as rdist
    seeds = []
    for r in (0, 3):
        orig = rdist.rank
        rdist.rank = lambda r=r: r
        try:
            JointTrainer.restore_dropout({'dropout_state': {'base_seed': 1234, 'call': 41}})
        finally:
            rdist.rank = orig
        seeds.append(ops.dropout_state())
    assert seeds[0] == ((1234 * 1000003 + 0) & 0xFFFFFFFFFFFF, 41) and seeds[1] == ((1234 * 1000003 + 3) & 0xFFFFFFFFFFFF, 41)


def test_trainer_state_round_trip_carries_the_decayed_eps():
    """state() -> load_state() after an adadelta_eps_decay: upstream builds its optimizers from the package's eps / lr
    (joint_train.py:98-111,127-140), so the resumed trainer's THREE optimizers must hold the decayed eps (round 4 wrote it to opt only),
    the weights must be the checkpoint's, and iters comes back as the checkpoint's - 1 (:108)."""
    import __graft_entry__ as g
    from robust_e2e_gan_amd.joint_train import JointTrainer
    from robust_e2e_gan_amd.model.enhance_model import EnhanceModel
    from robust_e2e_gan_amd.model.feat_model import FbankModel
    from robust_e2e_gan_amd.model.e2e_model import ShareE2E
    from robust_e2e_gan_amd.model.gan_model import GANModel
    from robust_e2e_gan_amd.utils.utils import adadelta_eps_decay

    def trainer(seed, **over):
        torch.manual_seed(seed)
        opt = g._tiny_opt()
        for k, v in over.items():
            setattr(opt, k, v)
        return JointTrainer(opt, EnhanceModel(opt), FbankModel(opt), ShareE2E(opt), GANModel(opt))
    a = trainer(1)
    a.opt.eps = adadelta_eps_decay(a.asr_optimizer, 0.01)
    assert a.opt.eps == 1e-10
    pkg = a.state(3, 100, 1.5, 0.7)
    b = trainer(2)
    assert b.asr_optimizer.param_groups[0]['eps'] == 1e-8
    assert b.load_state(pkg) == (3, 99, 1.5, 0.7)
    assert b.opt.eps == 1e-10
    assert [o.param_groups[0]['eps'] for o in (b.enhance_optimizer, b.asr_optimizer, b.gan_optimizer)] == [1e-10] * 3
    for ma, mb in ((a.asr_model, b.asr_model), (a.enhance_model, b.enhance_model), (a.gan_model, b.gan_model)):
        for (k, va), (_, vb) in zip(ma.state_dict().items(), mb.state_dict().items()):
            assert torch.equal(va, vb), k
    # the flat optimizer buffers ARE the parameters' storage: load_state_dict must have written through them, not rebound the tensors
    assert all(p.data_ptr() == b.asr_optimizer.flat[o:o + 1].data_ptr() for p, o in zip(b.asr_optimizer.params, b.asr_optimizer.offsets))
    # Adam resumes with the package's lr
    c = trainer(3, opt_type='adam', lr=0.005, beta1=0.5)
    pkg2 = c.state(0, 10)
    pkg2['lr'] = 0.00125
    d = trainer(4, opt_type='adam', lr=0.005, beta1=0.5)
    d.load_state(pkg2)
    assert d.opt.lr == 0.00125 and [o.param_groups[0]['lr'] for o in (d.enhance_optimizer, d.asr_optimizer, d.gan_optimizer)] == [0.00125] * 3


def test_filterbank_band_tables_rebuild_the_matrix():
    """band_from_matrix: the by-filter and the by-bin banded forms (forward / backward kernels of re2e_fbank_*) are both exact
    re-statements of the dense (257, 80) mel matrix."""
    import numpy as np
    from robust_e2e_gan_amd.model.feat_model import band_from_matrix, mel_matrix
    W = mel_matrix()
    off, ln, taps, maxw, NF, toff, tln, tw, maxc = band_from_matrix(W, 'cpu')
    F_ = W.shape[0]
    A, Bm = np.zeros_like(W), np.zeros_like(W)
    for j in range(NF):
        A[int(off[j]):int(off[j]) + int(ln[j]), j] = taps[j, :int(ln[j])].numpy()
    for f in range(F_):
        Bm[f, int(toff[f]):int(toff[f]) + int(tln[f])] = tw[f, :int(tln[f])].numpy()
    assert np.array_equal(A, W) and np.array_equal(Bm, W)
    assert maxw <= 32 and maxc <= 4 and NF == 80


def test_winograd_f24_matrices_reproduce_the_correlation():
    """tools/wino_f24_matrices.py: the F(2,4) matrices hard-coded in csrc/wino44.hip (points 0, 1, -1, 2, inf)."""
    import importlib.util
    import os
    from fractions import Fraction as Fr
    spec = importlib.util.spec_from_file_location('wino_f24', os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tools', 'wino_f24_matrices.py'))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    AT, G, BT = m.matrices((0, 1, -1, 2))
    m.check(AT, G, BT)
    assert AT == [[1, 1, 1, 1, 0], [0, 1, -1, 2, 1]]
    assert BT == [[2, -1, -2, 1, 0], [0, -2, -1, 1, 0], [0, 2, -3, 1, 0], [0, -1, 0, 1, 0], [0, 2, -1, -2, 1]]
    assert G == [[Fr(1, 2), 0, 0, 0], [Fr(-1, 2)] * 4, [Fr(-1, 6), Fr(1, 6), Fr(-1, 6), Fr(1, 6)], [Fr(1, 6), Fr(1, 3), Fr(2, 3), Fr(4, 3)], [0, 0, 0, 1]]


def test_row_maps_and_vgg_row_limits_host_logic():
    """ops.row_maps: the valid / padded physical rows of a ragged time-major (T, B, .) batch, only for length tensors made by ``lens_dev`` (a tensor
    that merely has the same content, or reuses the address, is not trusted), not for batches with next to no padding; VGG2L.row_limits: every
    layer's limit covers its need (the reach of four 3x3 convolutions and two 2x2 pools behind the cut at the pooled length) and is a whole patch."""
    import math
    from robust_e2e_gan_amd import ops
    from robust_e2e_gan_amd.model.e2e_common import lens_dev
    from robust_e2e_gan_amd.model.e2e_encoder import VGG2L
    T, B = 50, 12
    lens = [max(1, int(round(T * (1 - 0.5 * i / (B - 1))))) for i in range(B)]
    ld = lens_dev(lens, 'cpu')
    mp = ops.row_maps(ld, T, B)
    valid = [t * B + b for t in range(T) for b in range(B) if t < lens[b]]
    pad = [t * B + b for t in range(T) for b in range(B) if t >= lens[b]]
    assert mp is not None and mp.nv == len(valid) == sum(lens) and mp.ni == len(pad) and mp.rows == T * B
    assert mp.valid.tolist() == valid and mp.invalid.tolist() == pad and mp.ident == pad[0]
    assert ops.row_maps(torch.tensor(lens, dtype=torch.int32), T, B) is None             # not a registered length tensor
    assert ops.row_maps(ld, T, B + 1) is None                                            # not this batch
    assert ops.row_maps(lens_dev([T] * B, 'cpu'), T, B) is None                          # nothing to skip
    assert ops.row_maps(lens_dev([T] * (B - 1) + [T - 3], 'cpu'), T, B) is None          # next to nothing to skip
    vgg = VGG2L(1)
    Tm = 800
    lens = [int(round(Tm * (1 - 0.3 * i / 31))) for i in range(32)]
    l12, l21, l22 = vgg.row_limits(lens, Tm, 'cpu')
    for i, l in enumerate(lens):
        P = int(math.ceil(math.ceil(l / 2.0) / 2.0))
        o12, i12, o21, i21, o22, i22 = (int(t[i]) for t in (l12.out, l12.inp, l21.out, l21.inp, l22.out, l22.inp))
        assert o22 >= min(400, 2 * P) and o21 >= min(400, 2 * P + 1) and i21 >= min(400, 2 * P + 2) and o12 >= min(800, 4 * P + 4) and i12 >= min(800, 4 * P + 5)
        assert i22 == o21 and i21 == (o12 + 1) // 2
        assert all(v % 16 == 0 or v in (400, 800) for v in (o12, i12, o21, o22))
    assert l12.out_tail == Tm - min(int(v) for v in l12.out) and vgg.row_limits([Tm] * 8, Tm, 'cpu') is None
