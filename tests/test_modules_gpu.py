"""Module-surface parity on the GPU against the golden vectors produced from the reference import
(tests/golden/*.npz) -- EnhanceModel, FbankModel, E2E, GANModel/GANLoss and one composed
joint_train step.  Calls go through the C ABI (libre2e_hip.so).  Tolerance 1e-3 (north_star) on outputs and losses; gradients are held to
1.5e-3 of each tensor's largest entry (round 4: tightened from 3e-3 / 4e-3 -- the largest measured ratio over every comparison of this file
and tests/test_trainers_gpu.py is 1.0e-3, RE2E_PRINT_WORST=1 prints them; the fp64 arbitration of test_joint_step_gradients_vs_fp64_reference
shows the fp32 reference run itself up to 5.3e-4 away from the fp64 one, so two fp32 runs of one gradient legitimately differ by ~1e-3)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def _fx(golden_dir, name):
    return dict(np.load(os.path.join(golden_dir, name)))


def _opt():
    import __graft_entry__ as g
    return g._tiny_opt()


def _load(m, fx, prefix):
    sd = {k[len(prefix):]: torch.from_numpy(v) for k, v in fx.items() if k.startswith(prefix)}
    missing, unexpected = m.load_state_dict(sd, strict=True), None
    return m.to(DEV).train()


WORST = {}         # name -> err / scale of every comparison of the session (tools: RE2E_PRINT_WORST=1 prints the largest at exit)


def rel(name, got, ref, tol=1e-3, atol=1e-7):
    got = got.detach().float().cpu().numpy() if isinstance(got, torch.Tensor) else np.asarray(got)
    ref = np.asarray(ref)
    assert got.shape == ref.shape, (name, got.shape, ref.shape)
    err = np.abs(got - ref).max()
    if os.environ.get('RE2E_PRINT_WORST'):
        WORST[name] = max(WORST.get(name, 0.0), float(err / max(np.abs(ref).max(), 1e-30)))
        import atexit
        if len(WORST) == 1:
            atexit.register(lambda: print('WORST', sorted(((v, k) for k, v in WORST.items() if v > 3e-4), reverse=True)[:25]))
    assert np.isfinite(err) and err <= tol * np.abs(ref).max() + atol, '%s: err %.3e scale %.3e' % (name, err, np.abs(ref).max())


def test_state_dict_names_match_reference(golden_dir):
    from robust_e2e_gan_amd.model.enhance_model import EnhanceModel
    from robust_e2e_gan_amd.model.e2e_model import ShareE2E
    from robust_e2e_gan_amd.model.gan_model import GANModel
    fx = _fx(golden_dir, 'joint_tiny.npz')
    opt = _opt()
    for cls, pre in ((EnhanceModel, 'enh.'), (ShareE2E, 'asr.'), (GANModel, 'gan.')):
        want = {k[len(pre):]: v.shape for k, v in fx.items() if k.startswith(pre)}
        have = {k: tuple(v.shape) for k, v in cls(opt).state_dict().items()}
        assert want == have, (pre, set(want) ^ set(have))


def test_enhance_model(golden_dir):
    from robust_e2e_gan_amd.model.enhance_model import EnhanceModel
    fx = _fx(golden_dir, 'enhance_tiny.npz')
    m = _load(EnhanceModel(_opt()), fx, 'p.')
    t = lambda k: torch.from_numpy(fx[k])
    lens = torch.IntTensor(fx['lens'])
    out = m(t('mix'), t('mix_log'), lens)                      # CPU inputs: the module moves them (to_cuda)
    rel('enhance_out', out, fx['enhance_out'])
    for b, l in enumerate(fx['lens']):
        assert (out[b, l:] == 0).all()
    loss, out2 = m(t('mix'), t('mix_log'), lens, t('clean'), t('cos'))
    rel('l1_loss', loss.view(()), fx['l1_loss'])
    (loss + (out2 * torch.linspace(0.5, 1.5, 257).to(DEV)).mean()).backward()
    for k, p in m.named_parameters():
        rel('g.' + k, p.grad, fx['g.' + k], tol=1.5e-3)


def test_fbank_model(golden_dir):
    from robust_e2e_gan_amd.model.feat_model import FbankModel
    import argparse
    fx = _fx(golden_dir, 'fbank_tiny.npz')
    opt = _opt()
    m = FbankModel(opt)
    m.load_state_dict({'fc': torch.from_numpy(fx['W'])})
    m = m.to(DEV)
    x = torch.from_numpy(fx['x']).to(DEV).requires_grad_(True)
    cm = torch.from_numpy(fx['cmvn'])
    rel('y_nocmvn', m(x), fx['y_nocmvn'], tol=1e-5)
    y1 = m(x, cm)
    rel('y_cmvn', y1, fx['y_cmvn'], tol=1e-5)
    (y1 * torch.linspace(-1, 1, 80).to(DEV)).sum().backward()
    rel('dx', x.grad, fx['dx'], tol=1e-4)
    mc = FbankModel(argparse.Namespace(**{**vars(opt), 'train_dataset_len': 3}))
    mc.load_state_dict({'fc': torch.from_numpy(fx['W'])})
    mc = mc.to(DEV)
    lens = torch.IntTensor(fx['lens'])
    assert mc.compute_cmvn(x.detach(), lens) is None
    rel('cmvn_est', mc.compute_cmvn(x.detach(), lens), fx['cmvn_est'], tol=1e-4)


def test_e2e(golden_dir):
    from robust_e2e_gan_amd.model.e2e_model import E2E
    fx = _fx(golden_dir, 'e2e_tiny.npz')
    m = _load(E2E(_opt()), fx, 'p.')
    feat = torch.from_numpy(fx['feat']).to(DEV).requires_grad_(True)
    lens, tl = torch.IntTensor(fx['lens']), torch.IntTensor(fx['tlens'])
    hpad, hl = m.enc(feat, lens)
    rel('hpad', hpad, fx['hpad'])
    assert list(hl) == fx['hlens'].tolist()
    lc, la, acc = m(feat, torch.from_numpy(fx['targets']), lens, tl, 0.0)
    rel('loss_ctc', lc.view(1), fx['loss_ctc'])
    rel('loss_att', la.view(()), fx['loss_att'])
    assert abs(float(acc) - float(fx['acc'])) < 1e-6
    (0.5 * lc.view(()) + 0.5 * la).backward()
    rel('dfeat', feat.grad, fx['dfeat'], tol=1.5e-3)
    for k, p in m.named_parameters():
        rel('g.' + k, p.grad, fx['g.' + k], tol=1.5e-3)


def test_encoder_padded_rows_at_production_width():
    """Encoder.forward at eunits = 512 / eprojs = 512 on a ragged batch: the BLSTMP products run over the valid (t, b) rows (row maps: K = 1024 >= 384,
    > 256 valid rows) and the WHOLE ``hpad`` -- padded frames included, where upstream's tanh(Linear) over zero-padded frames leaves tanh(bias)
    (model/e2e_encoder.py:145-147) -- equals oracle.nets.encoder_forward; so do the gradients of a loss that reads every row."""
    from robust_e2e_gan_amd import ops
    from robust_e2e_gan_amd.model.e2e_encoder import Encoder
    from robust_e2e_gan_amd.model.e2e_common import lens_dev
    from oracle import nets
    torch.manual_seed(11)
    B, T, idim, elayers, eunits, eprojs = 8, 240, 40, 2, 512, 512
    lens = [240, 228, 200, 170, 150, 120, 100, 80]
    enc = Encoder('vggblstmp', idim, elayers, eunits, eprojs, [1] * (elayers + 1), 'skip', 0.0)
    for n_, p_ in enc.named_parameters():
        p_.data.uniform_(-0.08, 0.08) if p_.dim() > 1 else p_.data.uniform_(-0.3, 0.3)       # biases large enough that tanh(bias) is not ~0
    x = torch.randn(B, T, idim) * 0.7
    for b, l in enumerate(lens):
        x[b, l:] = 0
    p32 = {'enc.' + k: v.detach().clone() for k, v in enc.state_dict().items()}
    with torch.no_grad():
        href, hl_ref = nets.encoder_forward(p32, x, lens, elayers)
    go = torch.randn(href.shape, generator=torch.Generator().manual_seed(5)) * 0.1
    # gradients: the same oracle function in float64 arbitrates (two fp32 runs of a gradient through a VGG + 2 x BLSTMP-512 stack differ by more
    # than 1.5e-3 of its largest entry by themselves; DESIGN.md section 2, "Arbitration in float64")
    p = {k: v.double().requires_grad_(True) for k, v in p32.items()}
    h64, _ = nets.encoder_forward(p, x.double(), lens, elayers)
    (h64 * go.double()).sum().backward()
    assert float((h64.detach().float() - href).abs().max()) <= 1e-4
    enc = enc.to(DEV).train()
    # the products around the recurrences must take the valid-rows path at this width (that is the path whose padded rows this test pins)
    nl = enc.enc1.pooled_lens(lens)
    maps = ops.row_maps(lens_dev(nl, torch.device(DEV)), max(nl), B)
    assert maps is not None and maps.nv >= 256 and 2 * eunits >= ops.ROW_MAPS_MIN_K
    seen = []
    orig = ops.lib.call_supported

    def spy(name, *a):
        ok = orig(name, *a)
        if ok:
            seen.append(name)
        return ok
    ops.lib.call_supported = spy
    try:
        hpad, hl = enc(x.to(DEV), lens)
        (hpad * go.to(DEV)).sum().backward()
    finally:
        ops.lib.call_supported = orig
    assert seen.count('re2e_gemm_nt_rows') >= 2 * elayers and 're2e_gemm_tn_rows' in seen, seen
    assert list(hl) == list(hl_ref)
    rel('hpad, every row', hpad, href.numpy(), tol=1e-3)
    pad = torch.arange(href.shape[1])[None, :] >= torch.tensor(hl_ref)[:, None]
    assert pad.any()
    bias = p32['enc.enc2.bt%d.bias' % (elayers - 1)]
    assert float((hpad.detach().cpu()[pad] - torch.tanh(bias)[None, :]).abs().max()) <= 1e-6, 'padded frames hold tanh(bias), as upstream'
    for k, v in enc.named_parameters():
        rel('d ' + k, v.grad, p['enc.' + k].grad.float().numpy(), tol=1.5e-3)


@pytest.mark.parametrize('tag', ['a.', 'b.'])
def test_persistent_decoder_loop_vs_reference(golden_dir, tag):
    """The ONE-launch decoder loop (csrc/decloop.hip, forward and backward) against vectors from the reference's own Decoder + AttLoc
    (tests/golden/make_fixtures_dec.py: dunits=16, eprojs=32, adim=20 -- every other reference fixture has dunits=14, which the resident
    form declines, so those tests run the launch-per-token kernels).  Case b has T' = 300 > 256 frames: two frame chunks per utterance.
    The test asserts that the resident form IS what runs (non-zero workspace request, both directions)."""
    from robust_e2e_gan_amd import ops
    from robust_e2e_gan_amd.lib import query
    from robust_e2e_gan_amd.model.e2e_decoder import decoder_forward_hip
    fx = _fx(golden_dir, 'dec_persist_tiny.npz')
    hl, tl = fx[tag + 'hlens'].tolist(), fx[tag + 'tlens'].tolist()
    pg = {}
    for k, v in fx.items():
        if k.startswith(tag + 'p.dec.'):
            n = k[len(tag) + 2:]
            pg[n[4:] if n.startswith('dec.att.') else n] = torch.nn.Parameter(torch.from_numpy(v).clone().to(DEV))
    ys, o = [], 0
    for n in tl:
        ys.append(torch.from_numpy(fx[tag + 'ys'][o:o + n]))
        o += n
    B, T, E = fx[tag + 'hpad'].shape
    D, A = pg['dec.decoder.0.weight_hh'].shape[1], pg['att.mlp_enc.weight'].shape[0]
    C, Kf = pg['att.loc_conv.weight'].shape[0], pg['att.loc_conv.weight'].shape[3]
    L1 = max(tl) + 1
    assert ops.DECODER_PERSIST and ops.DECODER_FUSED
    assert query('re2e_dec_loop_workspace_bytes', L1, B, T, E, D, A, C, (Kf - 1) // 2) > 0, 'the resident forward declines this shape: the test would pass on the stepwise kernels'
    assert query('re2e_dec_loop_bwd_workspace_bytes', L1, B, T, E, D, A, C, (Kf - 1) // 2) > 0, 'the resident backward declines this shape'
    from robust_e2e_gan_amd import lib
    aborts = lib.query('re2e_lstm_abort_count')
    hg = torch.from_numpy(fx[tag + 'hpad']).to(DEV).requires_grad_(True)
    loss, acc, att = decoder_forward_hip(pg, hg, hl, ys, 11, return_att=True)
    rel(tag + 'att_w', att, fx[tag + 'att_w'], tol=1e-3)
    rel(tag + 'loss_att', loss.view(1), fx[tag + 'loss_att'], tol=1e-3)
    assert abs(float(acc) - float(fx[tag + 'acc'])) < 1e-6
    loss.backward()
    rel(tag + 'd_hpad', hg.grad, fx[tag + 'd_hpad'], tol=1e-3)
    for k, v in pg.items():
        rel(tag + 'g.' + k, v.grad, fx[tag + 'g.dec.' + (k[4:] if k.startswith('dec.') else k)], tol=1e-3)
    assert lib.query('re2e_lstm_abort_count') == aborts


def test_gan(golden_dir):
    from robust_e2e_gan_amd.model.gan_model import GANModel, GANLoss
    fx = _fx(golden_dir, 'gan_tiny.npz')
    m = _load(GANModel(_opt()), fx, 'p.')
    crit = GANLoss(use_lsgan=True)
    x = torch.from_numpy(fx['x']).to(DEV).requires_grad_(True)
    d = m(x)
    rel('d_out', d, fx['d_out'])
    lr = crit(d, True)
    lf = crit(m(x * 0.9 + 0.1), False)
    rel('l_real', lr.view(()), fx['l_real'])
    rel('l_fake', lf.view(()), fx['l_fake'])
    ((lr + lf) * 0.5).backward()
    rel('dx', x.grad, fx['dx'], tol=1.5e-3)
    for k, p in m.named_parameters():
        rel('g.' + k, p.grad, fx['g.' + k], tol=1.5e-3)
    for k, v in m.state_dict().items():
        if 'running' in k or 'num_batches' in k:
            rel('after.' + k, v, fx['after.' + k], tol=1e-4)


def test_joint_step(golden_dir):
    """One composed joint_train.py:156-212 step vs the reference-generated vectors (S1-S3 semantics)."""
    import __graft_entry__ as g
    from robust_e2e_gan_amd.joint_train import JointTrainer
    from robust_e2e_gan_amd.model.enhance_model import EnhanceModel
    from robust_e2e_gan_amd.model.feat_model import FbankModel
    from robust_e2e_gan_amd.model.e2e_model import ShareE2E
    from robust_e2e_gan_amd.model.gan_model import GANModel
    fx = _fx(golden_dir, 'joint_tiny.npz')
    W = _fx(golden_dir, 'fbank_tiny.npz')['W']
    opt = g._tiny_opt()
    enh, asr, gan = _load(EnhanceModel(opt), fx, 'enh.'), _load(ShareE2E(opt), fx, 'asr.'), _load(GANModel(opt), fx, 'gan.')
    fb = FbankModel(opt)
    fb.load_state_dict({'fc': torch.from_numpy(W)})
    fb = fb.to(DEV).train()
    tr = JointTrainer(opt, enh, fb, asr, gan)
    t = lambda k: torch.from_numpy(fx[k])
    data = (None, None, t('clean'), None, t('mix'), t('mix_log'), None, t('targets'), torch.IntTensor(fx['lens']), torch.IntTensor(fx['tlens']))
    # capture gradients before the optimizer step overwrites nothing (grads stay in the flat buffers)
    out = JointTrainer.to_floats(tr.step(data, 0.0, t('cmvn')))
    rel('enhance_out', tr.last['enhance_out'], fx['enhance_out'])
    rel('enhance_feat', tr.last['enhance_feat'], fx['enhance_feat'], tol=1e-4)
    for k in ('loss', 'loss_ctc', 'loss_att', 'enhance_loss', 'coral_loss', 'loss_D'):
        assert abs(out['train/' + k] - float(fx[k][0])) <= 1e-3 * max(1.0, abs(float(fx[k][0]))), (k, out['train/' + k], fx[k])
    assert abs(out['train/gan_loss'] - float(fx['gan_loss'][0])) <= 1e-3
    assert abs(out['train/acc'] - float(fx['acc'])) < 1e-6
    assert abs(out['grad_norm'] - float(fx['grad_norm_asr'])) <= 2e-3 * float(fx['grad_norm_asr'])
    for pre, m in (('genh.', enh), ('gasr.', asr), ('ggan.', gan)):
        for k, p in m.named_parameters():
            rel(pre + k, p.grad, fx[pre + k], tol=1.5e-3)
    for pre, m in (('enh_after.', enh), ('asr_after.', asr), ('gan_after.', gan)):
        for k, v in m.state_dict().items():
            if v.dtype.is_floating_point:
                rel(pre + k, v, fx[pre + k], tol=1e-3, atol=2e-5)


def test_joint_step_gradients_vs_fp64_reference(golden_dir):
    """Arbitrated gradient bar: every parameter gradient of the composed step within 1e-3 (relative to the tensor's max) of
    the REFERENCE modules run in float64 (tests/golden/make_fixtures_fp64.py).  The fp32 reference run itself sits at up
    to 5.3e-4 from these vectors (tests/test_oracle_golden.py::test_fp32_sides_within_1e3_of_fp64), which is why the
    HIP-vs-fp32-fixture comparisons elsewhere in this file allow up to 3e-3: two fp32 roundings of the same quantity."""
    import __graft_entry__ as g
    from robust_e2e_gan_amd.joint_train import JointTrainer
    from robust_e2e_gan_amd.model.enhance_model import EnhanceModel
    from robust_e2e_gan_amd.model.feat_model import FbankModel
    from robust_e2e_gan_amd.model.e2e_model import ShareE2E
    from robust_e2e_gan_amd.model.gan_model import GANModel
    fx = _fx(golden_dir, 'joint_tiny.npz')
    f64 = _fx(golden_dir, 'joint_tiny_fp64.npz')
    W = _fx(golden_dir, 'fbank_tiny.npz')['W']
    opt = g._tiny_opt()
    enh, asr, gan = _load(EnhanceModel(opt), fx, 'enh.'), _load(ShareE2E(opt), fx, 'asr.'), _load(GANModel(opt), fx, 'gan.')
    fb = FbankModel(opt)
    fb.load_state_dict({'fc': torch.from_numpy(W)})
    tr = JointTrainer(opt, enh, fb.to(DEV).train(), asr, gan)
    t = lambda k: torch.from_numpy(fx[k])
    data = (None, None, t('clean'), None, t('mix'), t('mix_log'), None, t('targets'), torch.IntTensor(fx['lens']), torch.IntTensor(fx['tlens']))
    out = JointTrainer.to_floats(tr.step(data, 0.0, t('cmvn')))
    for k in ('loss', 'loss_ctc', 'loss_att', 'enhance_loss', 'coral_loss', 'loss_D'):
        assert abs(out['train/' + k] - float(f64[k])) <= 1e-4 * max(1.0, abs(float(f64[k]))), (k, out['train/' + k], f64[k])
    assert abs(out['grad_norm'] - float(f64['grad_norm_asr'])) <= 1e-3 * float(f64['grad_norm_asr'])
    rel('enhance_out', tr.last['enhance_out'], f64['enhance_out'], tol=1e-4)
    worst = (0.0, None)
    for pre, m in (('genh.', enh), ('gasr.', asr), ('ggan.', gan)):
        for k, p in m.named_parameters():
            ref = f64[pre + k]
            err, scale = np.abs(p.grad.double().cpu().numpy() - ref).max(), np.abs(ref).max()
            if scale < 1e-12:          # att.gvec.bias: the true gradient is 0 (softmax shift invariance)
                assert err < 1e-7, (k, err)
                continue
            worst = max(worst, (err / scale, pre + k))
    assert worst[0] < 1e-3, worst


def test_joint_step_overlap_equals_single_stream(golden_dir):
    """The multi-stream schedule (two-phase backward, side / weight-gradient streams) must produce the gradients of the
    plain single-stream ``loss.backward()`` step: every parameter of the four nets after one update, to rounding."""
    import __graft_entry__ as g
    from robust_e2e_gan_amd.joint_train import JointTrainer
    from robust_e2e_gan_amd.model.enhance_model import EnhanceModel
    from robust_e2e_gan_amd.model.feat_model import FbankModel
    from robust_e2e_gan_amd.model.e2e_model import ShareE2E
    from robust_e2e_gan_amd.model.gan_model import GANModel
    fx = _fx(golden_dir, 'joint_tiny.npz')
    W = _fx(golden_dir, 'fbank_tiny.npz')['W']
    t = lambda k: torch.from_numpy(fx[k])
    data = (None, None, t('clean'), None, t('mix'), t('mix_log'), None, t('targets'), torch.IntTensor(fx['lens']), torch.IntTensor(fx['tlens']))
    grads = {}
    for overlap in (False, True):
        opt = g._tiny_opt()
        opt.coral_loss_lambda = 2e6             # raw CORAL is ~4e-9 at initialisation: at this weight the clean branch's (CORAL-only)
                                                # contribution is ~15 % of the conv-stack gradients (a dropped one cannot hide in the tolerance)
        enh, asr, gan = _load(EnhanceModel(opt), fx, 'enh.'), _load(ShareE2E(opt), fx, 'asr.'), _load(GANModel(opt), fx, 'gan.')
        fb = FbankModel(opt)
        fb.load_state_dict({'fc': torch.from_numpy(W)})
        tr = JointTrainer(opt, enh, fb.to(DEV).train(), asr, gan)
        tr.overlap_dstep = overlap
        tr.step(data, 0.0, t('cmvn'))
        torch.cuda.synchronize()
        from robust_e2e_gan_amd import ops
        # the step's stream routing must not leak into whatever runs next in the process (validation, other trainers)
        assert ops.MULTI_STREAM is False and ops.WGRAD_STREAM is None and ops.AUX_STREAM is None
        grads[overlap] = {n + '.' + k: p.grad.clone() for n, m in (('enh', enh), ('asr', asr), ('gan', gan)) for k, p in m.named_parameters()}
    for k, ref in grads[False].items():
        rel(k, grads[True][k], ref.cpu().numpy(), tol=2e-5, atol=1e-9)


@pytest.mark.parametrize('overlap', [True, False])
def test_joint_step_without_gan_vs_oracle(golden_dir, overlap):
    """isGAN=False (the reference's default for --isGAN) with a CORAL weight: the clean branch's conv stack is reached ONLY
    through CORAL and the shared BLSTMP.  With the multi-stream schedule that stack runs on the side stream behind a graph
    cut; its gradients must still arrive (they were dropped on this path once), in both schedules, against the oracle."""
    import __graft_entry__ as g
    from oracle import joint as oj
    from robust_e2e_gan_amd.joint_train import JointTrainer
    from robust_e2e_gan_amd.model.enhance_model import EnhanceModel
    from robust_e2e_gan_amd.model.feat_model import FbankModel
    from robust_e2e_gan_amd.model.e2e_model import ShareE2E
    fx = _fx(golden_dir, 'joint_tiny.npz')
    W = torch.from_numpy(_fx(golden_dir, 'fbank_tiny.npz')['W'])
    opt = g._tiny_opt()
    opt.isGAN, opt.coral_loss_lambda = False, 2e6       # raw CORAL is ~4e-9 at initialisation: the weight that makes it matter
    t = lambda k: torch.from_numpy(fx[k])
    lens, tls = fx['lens'].tolist(), fx['tlens'].tolist()
    cfg = dict(enhance_layers=2, elayers=2, mtlalpha=0.5, enhance_loss_lambda=1.0, coral_loss_lambda=2e6, gan_loss_lambda=1.0, grad_clip=5.0,
               eps=1e-8, isGAN=False, enhance_loss_type='L2')
    sub = lambda pre: {k[len(pre):]: torch.from_numpy(v) for k, v in fx.items() if k.startswith(pre)}
    st = oj.JointState(sub('enh.'), sub('asr.'), sub('gan.'), W, cfg)
    ref = oj.joint_step(st, (t('clean'), t('mix'), t('mix_log'), t('targets'), lens, tls), t('cmvn'))
    # the oracle itself must see the clean-only path: CORAL-only gradient of the conv stack is not negligible
    cfg0 = dict(cfg, coral_loss_lambda=0.0)
    ref0 = oj.joint_step(oj.JointState(sub('enh.'), sub('asr.'), sub('gan.'), W, cfg0),
                         (t('clean'), t('mix'), t('mix_log'), t('targets'), lens, tls), t('cmvn'))
    k0 = 'enc.enc1.conv1_1.weight'
    assert (ref['g_asr'][k0] - ref0['g_asr'][k0]).abs().max() > 5e-2 * ref['g_asr'][k0].abs().max()
    enh, asr = _load(EnhanceModel(opt), fx, 'enh.'), _load(ShareE2E(opt), fx, 'asr.')
    fb = FbankModel(opt)
    fb.load_state_dict({'fc': W})
    tr = JointTrainer(opt, enh, fb.to(DEV).train(), asr, None)
    tr.overlap_dstep = overlap
    data = (None, None, t('clean'), None, t('mix'), t('mix_log'), None, t('targets'), torch.IntTensor(lens), torch.IntTensor(tls))
    out = JointTrainer.to_floats(tr.step(data, 0.0, t('cmvn')))
    assert getattr(asr, 'clean_cut', None) is None           # no stale graph kept alive
    assert 'train/loss_D' not in out and 'train/gan_loss' not in out
    for k in ('loss', 'loss_ctc', 'loss_att', 'enhance_loss', 'coral_loss'):
        a, b = out['train/' + k], float(ref[k])
        assert abs(a - b) <= 1e-3 * max(1.0, abs(b)), (k, a, b)
    assert abs(out['grad_norm'] - ref['grad_norm_asr']) <= 2e-3 * ref['grad_norm_asr']
    for pre, m, gd in (('enh', enh, ref['g_enh']), ('asr', asr, ref['g_asr'])):
        for k, p in m.named_parameters():
            if k in gd:
                rel(pre + '.' + k, p.grad, gd[k].numpy(), tol=1.5e-3, atol=1e-7)


@pytest.mark.parametrize('lens,tls', [([33], [1]), ([37, 6], [2, 1]), ([33, 33, 32, 17, 5], [4, 1, 3, 2, 1]), ([64, 8], [7, 1])])
def test_joint_step_ragged_shapes_vs_oracle(golden_dir, lens, tls):
    """Edge shapes against the CPU oracle: a single utterance, lengths that are not multiples of 4 (ceil-mode pooling),
    one-label targets, a batch mixing long and very short utterances (T' as small as 2; the discriminator needs T >= 32)."""
    import __graft_entry__ as g
    from oracle import joint as oj
    from robust_e2e_gan_amd.joint_train import JointTrainer
    from robust_e2e_gan_amd.model.enhance_model import EnhanceModel
    from robust_e2e_gan_amd.model.feat_model import FbankModel
    from robust_e2e_gan_amd.model.e2e_model import ShareE2E
    from robust_e2e_gan_amd.model.gan_model import GANModel
    fx = _fx(golden_dir, 'joint_tiny.npz')
    W = torch.from_numpy(_fx(golden_dir, 'fbank_tiny.npz')['W'])
    opt = g._tiny_opt()
    B, T = len(lens), max(lens)
    gen = torch.Generator().manual_seed(100 + sum(lens))
    clean = torch.rand(B, T, 257, generator=gen) * 300
    mix = clean + torch.rand(B, T, 257, generator=gen) * 100
    mix_log = torch.randn(B, T, 257, generator=gen)
    for b, l in enumerate(lens):
        clean[b, l:], mix[b, l:], mix_log[b, l:] = 0, 0, 0
    targets = torch.randint(1, opt.odim - 1, (sum(tls),), generator=gen)
    cm = torch.from_numpy(fx['cmvn'])
    enh, asr, gan = _load(EnhanceModel(opt), fx, 'enh.'), _load(ShareE2E(opt), fx, 'asr.'), _load(GANModel(opt), fx, 'gan.')
    fb = FbankModel(opt)
    fb.load_state_dict({'fc': W})
    cfg = dict(enhance_layers=2, elayers=2, mtlalpha=0.5, enhance_loss_lambda=1.0, coral_loss_lambda=0.5, gan_loss_lambda=1.0, grad_clip=5.0,
               eps=1e-8, isGAN=True, enhance_loss_type='L2')
    sub = lambda pre: {k[len(pre):]: torch.from_numpy(v) for k, v in fx.items() if k.startswith(pre)}
    st = oj.JointState(sub('enh.'), sub('asr.'), sub('gan.'), W, cfg)
    ref = oj.joint_step(st, (clean, mix, mix_log, targets, lens, tls), cm)
    tr = JointTrainer(opt, enh, fb.to(DEV).train(), asr, gan)
    data = (None, None, clean, None, mix, mix_log, None, targets, torch.IntTensor(lens), torch.IntTensor(tls))
    out = JointTrainer.to_floats(tr.step(data, 0.0, cm))
    for k in ('loss', 'loss_ctc', 'loss_att', 'enhance_loss', 'coral_loss', 'loss_D'):
        a, b = out['train/' + k], float(ref[k])
        assert abs(a - b) <= 1e-3 * max(1.0, abs(b)), (k, a, b)
    assert abs(out['train/acc'] - ref['acc']) < 1e-6
    assert abs(out['grad_norm'] - ref['grad_norm_asr']) <= 3e-3 * ref['grad_norm_asr']
    rel('enhance_out', tr.last['enhance_out'], ref['enhance_out'].numpy())
    for pre, m, gd in (('enh', enh, ref['g_enh']), ('asr', asr, ref['g_asr']), ('gan', gan, ref['g_gan'])):
        for k, p in m.named_parameters():
            if k in gd:
                rel(pre + '.' + k, p.grad, gd[k].numpy(), tol=1.5e-3, atol=1e-7)


def test_instance_norm_discriminator_vs_torch():
    """--norm_D instance (gan_model.py:42-46,57): InstanceNorm2d(affine=False) between the convolutions, which then carry biases; the
    module against the same network built from torch.nn layers with the same weights (forward, input gradient, weight gradients)."""
    import torch.nn as nn
    import __graft_entry__ as g
    from robust_e2e_gan_amd.model.gan_model import GANModel
    opt = g._tiny_opt()
    opt.norm_D, opt.netD_type, opt.ndf = 'instance', 'basic', 8
    torch.manual_seed(3)
    gan = GANModel(opt).to(DEV).train()
    sd = gan.state_dict()
    assert not any('running' in k for k in sd), sd.keys()
    ndf = opt.ndf
    seq = [nn.Conv2d(1, ndf, 4, 2, 1), nn.LeakyReLU(0.2)]
    chans = [(ndf, 2 * ndf, 2), (2 * ndf, 4 * ndf, 2), (4 * ndf, 8 * ndf, 1)]
    for ci, co, st in chans:
        seq += [nn.Conv2d(ci, co, 4, st, 1, bias=True), nn.InstanceNorm2d(co, affine=False, track_running_stats=False), nn.LeakyReLU(0.2)]
    seq += [nn.Conv2d(8 * ndf, 1, 4, 1, 1)]
    ref = nn.Sequential(*seq)
    ref.load_state_dict({k.replace('model.', ''): v.cpu() for k, v in sd.items()})
    x = torch.randn(3, 64, 80)
    xr = x.clone().requires_grad_(True)
    yr = ref(xr.unsqueeze(1))
    w = torch.randn_like(yr)
    (yr * w).sum().backward()
    xg = x.to(DEV).requires_grad_(True)
    y = gan(xg)
    (y * w.to(DEV)).sum().backward()
    assert y.shape == yr.shape
    tol = lambda a, b, t: float((a.cpu() - b).abs().max()) <= t * float(b.abs().max()) + 1e-7
    assert tol(y.detach(), yr.detach(), 2e-4)
    assert tol(xg.grad, xr.grad, 1e-3)
    # (a bias in front of an InstanceNorm has an exactly-zero gradient: both sides hold rounding noise there, so the yardstick is the
    # largest gradient of the network, not the tensor's own)
    gscale = max(float(pr.grad.abs().max()) for pr in ref.parameters())
    for (k, p), (kr, pr) in zip(gan.model.named_parameters(), ref.named_parameters()):
        assert k == kr and float((p.grad.cpu() - pr.grad).abs().max()) <= 1e-3 * max(float(pr.grad.abs().max()), 1e-2 * gscale), k


def test_instance_norm_unet_vs_torch():
    """--enhance_norm instance (enhance_model.py:258-261): the pix2pix U-Net with InstanceNorm2d(affine=False) and biased convolutions,
    unet_128 on a (2, 64, 32) image, against the same network built from torch.nn layers with the same weights."""
    import argparse
    import torch.nn as nn
    import __graft_entry__ as g
    from robust_e2e_gan_amd.model.enhance_model import UnetGenerator

    class Block(nn.Module):                       # enhance_model.py:249-303, norm_layer = InstanceNorm2d
        def __init__(self, outer, inner, inp=None, sub=None, outermost=False, innermost=False):
            super().__init__()
            self.outermost = outermost
            inp = outer if inp is None else inp
            down = nn.Conv2d(inp, inner, 4, 2, 1, bias=True)
            IN = lambda c: nn.InstanceNorm2d(c, affine=False, track_running_stats=False)
            if outermost:
                m = [down, sub, nn.ReLU(), nn.ConvTranspose2d(inner * 2, outer, 4, 2, 1), nn.Sigmoid()]
            elif innermost:
                m = [nn.LeakyReLU(0.2), down, nn.ReLU(), nn.ConvTranspose2d(inner, outer, 4, 2, 1, bias=True), IN(outer)]
            else:
                m = [nn.LeakyReLU(0.2), down, IN(inner), sub, nn.ReLU(), nn.ConvTranspose2d(inner * 2, outer, 4, 2, 1, bias=True), IN(outer)]
            self.model = nn.Sequential(*m)

        def forward(self, x):
            return self.model(x) if self.outermost else torch.cat([x, self.model(x)], 1)
    ngf = 4
    torch.manual_seed(9)
    net = UnetGenerator(1, 1, 5, ngf, 0.0, norm='instance').to(DEV).train()
    for p_ in net.parameters():
        torch.nn.init.normal_(p_, 0.0, 0.2)
    b = Block(ngf * 8, ngf * 8, innermost=True)
    b = Block(ngf * 4, ngf * 8, sub=b)
    b = Block(ngf * 2, ngf * 4, sub=b)
    b = Block(ngf, ngf * 2, sub=b)
    ref = Block(1, ngf, inp=1, sub=b, outermost=True)
    ref.load_state_dict({k[len('model.'):]: v.cpu() for k, v in net.state_dict().items()})
    x = torch.randn(2, 1, 64, 32)
    xr = x.clone().requires_grad_(True)
    yr = ref(xr)
    w = torch.randn_like(yr)
    (yr * w).sum().backward()
    xg = x.permute(0, 2, 3, 1).contiguous().to(DEV).requires_grad_(True)
    y, _ = net(xg, None)
    (y * w.permute(0, 2, 3, 1).to(DEV)).sum().backward()
    err = lambda a, b_: float((a.cpu() - b_).abs().max())
    assert err(y.detach().permute(0, 3, 1, 2), yr.detach()) <= 2e-4 * float(yr.detach().abs().max())
    assert err(xg.grad.permute(0, 3, 1, 2), xr.grad) <= 1e-3 * float(xr.grad.abs().max())
    gscale = max(float(p_.grad.abs().max()) for p_ in ref.parameters())
    for (k, p_), (kr, pr) in zip(net.named_parameters(), ref.named_parameters()):
        assert k == 'model.' + kr and err(p_.grad, pr.grad) <= 1e-3 * max(float(pr.grad.abs().max()), 1e-2 * gscale), k
