"""Pin the CPU oracle (oracle/) to the golden vectors produced from the reference import
(tests/golden/make_fixtures.py).  CPU only."""
import os

import numpy as np
import torch

from oracle import nets, joint
from oracle.fbank_tables import mel_matrix

TOL = dict(rtol=2e-4, atol=2e-5)


def _load(golden_dir, name):
    return {k: v for k, v in np.load(os.path.join(golden_dir, name), allow_pickle=False).items()}


def _sub(fx, prefix, grad=False):
    out = {}
    for k, v in fx.items():
        if k.startswith(prefix):
            t = torch.from_numpy(v)
            if grad and t.dtype.is_floating_point and 'running_' not in k:
                t = t.clone().requires_grad_(True)
            out[k[len(prefix):]] = t
    return out


def test_mel_matrix_matches_reference_table(golden_dir):
    fx = _load(golden_dir, 'fbank_tiny.npz')
    W = mel_matrix()
    assert W.shape == fx['W'].shape
    assert np.abs(W - fx['W']).max() < 2e-5       # reference literals carry 5-6 digits
    assert ((W != 0) == (fx['W'] != 0)).all()


def test_enhance(golden_dir):
    fx = _load(golden_dir, 'enhance_tiny.npz')
    p = _sub(fx, 'p.', grad=True)
    mix, mix_log = torch.from_numpy(fx['mix']), torch.from_numpy(fx['mix_log'])
    lens = fx['lens'].tolist()
    out = nets.enhance_forward(p, mix, mix_log, lens, 2)
    np.testing.assert_allclose(out.detach().numpy(), fx['enhance_out'], **TOL)
    # padded rows exactly zero (Appendix A.2)
    for b, l in enumerate(lens):
        assert (out[b, l:] == 0).all()
    loss, out2 = nets.enhance_forward(p, mix, mix_log, lens, 2, torch.from_numpy(fx['clean']),
                                      torch.from_numpy(fx['cos']))
    np.testing.assert_allclose(loss.detach().numpy(), fx['l1_loss'], rtol=1e-5)
    (loss + (out2 * torch.linspace(0.5, 1.5, 257)).mean()).backward()
    for k, v in p.items():
        np.testing.assert_allclose(v.grad.numpy(), fx['g.' + k], rtol=2e-3, atol=2e-6, err_msg=k)


def test_fbank_and_cmvn(golden_dir):
    fx = _load(golden_dir, 'fbank_tiny.npz')
    x = torch.from_numpy(fx['x']).requires_grad_(True)
    W, cm = torch.from_numpy(fx['W']), torch.from_numpy(fx['cmvn'])
    y0 = nets.fbank_forward(x, W)
    y1 = nets.fbank_forward(x, W, cm)
    np.testing.assert_allclose(y0.detach().numpy(), fx['y_nocmvn'], **TOL)
    np.testing.assert_allclose(y1.detach().numpy(), fx['y_cmvn'], **TOL)
    assert np.allclose(y0.detach().numpy()[0, -1], np.log(1e-7)) or fx['lens'][0] == x.shape[1]
    (y1 * torch.linspace(-1, 1, 80)).sum().backward()
    np.testing.assert_allclose(x.grad.numpy(), fx['dx'], rtol=1e-4, atol=1e-6)
    acc = nets.CmvnAccumulator(80, 3)
    lens = fx['lens']
    assert acc.update(y0.detach(), lens) is None
    est = acc.update(y0.detach(), lens)
    np.testing.assert_allclose(est, fx['cmvn_est'], rtol=1e-4, atol=1e-5)


def test_e2e(golden_dir):
    fx = _load(golden_dir, 'e2e_tiny.npz')
    p = _sub(fx, 'p.', grad=True)
    feat = torch.from_numpy(fx['feat']).requires_grad_(True)
    lc, la, acc, hpad, hl = nets.e2e_forward(p, feat, torch.from_numpy(fx['targets']),
                                              fx['lens'].tolist(), fx['tlens'].tolist(), 2)
    np.testing.assert_allclose(hpad.detach().numpy(), fx['hpad'], **TOL)
    assert list(hl) == fx['hlens'].tolist()
    np.testing.assert_allclose(lc.detach().numpy(), fx['loss_ctc'], rtol=1e-4)
    np.testing.assert_allclose(la.detach().numpy(), fx['loss_att'], rtol=1e-4)
    assert abs(acc - float(fx['acc'])) < 1e-9
    (0.5 * lc + 0.5 * la).backward()
    np.testing.assert_allclose(feat.grad.numpy(), fx['dfeat'], rtol=2e-3, atol=1e-6)
    for k, v in p.items():
        if k.startswith('dec.att.'):
            continue       # shared module registered twice (e2e_model.py:137); grads live on att.*
        np.testing.assert_allclose(v.grad.numpy(), fx['g.' + k], rtol=5e-3, atol=2e-6, err_msg=k)


def _dec_case(fx, tag):
    """Parameters of a dec_persist_tiny case under the oracle's names (the attention module hangs off the decoder there)."""
    p = {}
    for k, v in fx.items():
        if k.startswith(tag + 'p.dec.'):
            n = k[len(tag) + 2:]
            n = n[4:] if n.startswith('dec.att.') else n
            p[n] = torch.from_numpy(v).clone().requires_grad_(True)
    tl = fx[tag + 'tlens'].tolist()
    ys, o = [], 0
    for n in tl:
        ys.append(torch.from_numpy(fx[tag + 'ys'][o:o + n]))
        o += n
    return p, ys


def test_decoder_at_persistent_widths(golden_dir):
    """Decoder + AttLoc alone at dunits=16 / eprojs=32 / adim=20 (make_fixtures_dec.py): the widths csrc/decloop.hip accepts, one case
    with T' = 300 > 256 frames.  Attention weights of every step, loss, accuracy, d(hpad), every parameter gradient."""
    fx = _load(golden_dir, 'dec_persist_tiny.npz')
    for tag in ('a.', 'b.'):
        p, ys = _dec_case(fx, tag)
        hpad = torch.from_numpy(fx[tag + 'hpad']).requires_grad_(True)
        loss, acc, att = nets.decoder_forward(p, hpad, fx[tag + 'hlens'].tolist(), ys, 11, return_att=True)
        np.testing.assert_allclose(att.detach().numpy(), fx[tag + 'att_w'], rtol=2e-4, atol=1e-7)
        np.testing.assert_allclose(loss.detach().numpy().reshape(-1), fx[tag + 'loss_att'], rtol=1e-5)
        assert abs(acc - float(fx[tag + 'acc'])) < 1e-9
        loss.backward()
        np.testing.assert_allclose(hpad.grad.numpy(), fx[tag + 'd_hpad'], rtol=2e-3, atol=1e-7)
        for k, v in p.items():
            ref = fx[tag + 'g.dec.' + (k[4:] if k.startswith('dec.') else k)]       # fixture keys: names inside the reference's Decoder module
            err = np.abs(v.grad.numpy() - ref).max()
            assert err <= 1e-4 * np.abs(ref).max() + 1e-7, (tag, k, err)


def test_gan(golden_dir):
    fx = _load(golden_dir, 'gan_tiny.npz')
    full = _sub(fx, 'p.')
    p = {k: v.clone().requires_grad_(True) for k, v in full.items() if v.dtype.is_floating_point and 'running' not in k}
    buf = {k: v.clone() for k, v in full.items() if 'running' in k or 'num_batches' in k}
    x = torch.from_numpy(fx['x']).requires_grad_(True)
    d = nets.discriminator_forward(p, buf, x)
    np.testing.assert_allclose(d.detach().numpy(), fx['d_out'], **TOL)
    lr = nets.gan_loss(d, True)
    lf = nets.gan_loss(nets.discriminator_forward(p, buf, x * 0.9 + 0.1), False)
    np.testing.assert_allclose(lr.detach().numpy(), fx['l_real'], rtol=1e-4)
    np.testing.assert_allclose(lf.detach().numpy(), fx['l_fake'], rtol=1e-4)
    ((lr + lf) * 0.5).backward()
    np.testing.assert_allclose(x.grad.numpy(), fx['dx'], rtol=2e-3, atol=1e-7)
    for k, v in p.items():
        np.testing.assert_allclose(v.grad.numpy(), fx['g.' + k], rtol=5e-3, atol=1e-6, err_msg=k)
    for k, v in buf.items():     # BN running stats after two train-mode forwards (Appendix A.12)
        np.testing.assert_allclose(v.numpy(), fx['after.' + k], rtol=1e-4, atol=1e-6, err_msg=k)


def joint_cfg():
    return dict(enhance_layers=2, elayers=2, mtlalpha=0.5, enhance_loss_lambda=1.0, coral_loss_lambda=0.5,
                gan_loss_lambda=1.0, grad_clip=5.0, eps=1e-8, isGAN=True, enhance_loss_type='L2')


def test_joint_step(golden_dir):
    fx = _load(golden_dir, 'joint_tiny.npz')
    st = joint.JointState(_sub(fx, 'enh.'), _sub(fx, 'asr.'), _sub(fx, 'gan.'),
                          torch.from_numpy(_load(golden_dir, 'fbank_tiny.npz')['W']), joint_cfg())
    batch = (torch.from_numpy(fx['clean']), torch.from_numpy(fx['mix']), torch.from_numpy(fx['mix_log']),
             torch.from_numpy(fx['targets']), fx['lens'].tolist(), fx['tlens'].tolist())
    out = joint.joint_step(st, batch, torch.from_numpy(fx['cmvn']))
    np.testing.assert_allclose(out['enhance_out'].numpy(), fx['enhance_out'], **TOL)
    for k in ('loss', 'loss_ctc', 'loss_att', 'enhance_loss', 'coral_loss', 'gan_loss', 'loss_D'):
        np.testing.assert_allclose(out[k].numpy().reshape(-1), fx[k], rtol=3e-4, err_msg=k)
    assert abs(out['acc'] - float(fx['acc'])) < 1e-9
    assert abs(out['grad_norm_asr'] - float(fx['grad_norm_asr'])) < 2e-3 * float(fx['grad_norm_asr'])
    assert abs(out['grad_norm_gan'] - float(fx['grad_norm_gan'])) < 2e-3 * float(fx['grad_norm_gan'])
    # gradients are compared relative to each tensor's scale (att.gvec.bias has a true gradient
    # of exactly 0 -- softmax shift invariance -- so only rounding noise ~1e-9 is left there)
    for pre, gd in (('genh.', out['g_enh']), ('gasr.', out['g_asr']), ('ggan.', out['g_gan'])):
        for k, v in gd.items():
            ref = fx[pre + k]
            err = np.abs(v.numpy() - ref).max()
            assert err <= 1e-3 * np.abs(ref).max() + 1e-7, (pre + k, err, np.abs(ref).max())
    # first Adadelta step: parameter deltas for three named tensors
    for net, d, names in (('enh', st.enh, ['enc1.l_last.weight']), ('asr', st.asr, ['enc.enc2.bt0.weight', 'dec.output.bias']),
                          ('gan', st.gan, ['model.0.weight'])):
        for n in names:
            np.testing.assert_allclose(d[n].detach().numpy(), fx['%s_after.%s' % (net, n)], rtol=1e-3, atol=2e-5,
                                       err_msg=n)
    for k, v in st.gan_buf.items():
        np.testing.assert_allclose(v.numpy(), fx['gan_after.' + k], rtol=1e-4, atol=1e-6, err_msg=k)


# ---- the other trainers' steps and the validation pass (tests/golden/make_fixtures_trainers.py) ----
def _after_close(params, fx, prefix, rtol=1e-3, atol=2e-5):
    for k, v in params.items():
        np.testing.assert_allclose(v.detach().numpy(), fx[prefix + k], rtol=rtol, atol=atol, err_msg=prefix + k)


def _tr_batch(fx):
    return (torch.from_numpy(fx['clean']), torch.from_numpy(fx['mix']), torch.from_numpy(fx['mix_log']), torch.from_numpy(fx['cos']),
            fx['lens'].tolist())


def test_enhance_base_step(golden_dir):
    from oracle import trainers
    fx, cfg = _load(golden_dir, 'trainers_tiny.npz'), joint_cfg()
    enh = trainers.leaf(_sub(fx, 'base.p.'))
    out = trainers.enhance_base_step(enh, joint.Adadelta(enh, eps=cfg['eps']), _tr_batch(fx), cfg)
    np.testing.assert_allclose(out['loss'].numpy().reshape(-1), fx['base.loss'], rtol=3e-4)
    assert abs(out['grad_norm'] - float(fx['base.grad_norm'])) < 2e-3 * float(fx['base.grad_norm'])
    _after_close(enh, fx, 'base.after.')


def test_enhance_fbank_step(golden_dir):
    from oracle import trainers
    fx = _load(golden_dir, 'trainers_tiny.npz')
    W = torch.from_numpy(fx['fbank_W'])
    for kind in ('L2', 'L1', 'smooth_L1'):
        cfg = dict(joint_cfg(), enhance_loss_type=kind)
        enh = trainers.leaf(_sub(fx, 'fbank.p.'))
        out = trainers.enhance_fbank_step(enh, joint.Adadelta(enh, eps=cfg['eps']), W, _tr_batch(fx), cfg)
        np.testing.assert_allclose(out['loss'].numpy().reshape(-1), fx['fbank.%s.loss' % kind], rtol=3e-4, err_msg=kind)
        assert abs(out['grad_norm'] - float(fx['fbank.%s.grad_norm' % kind])) < 2e-3 * float(fx['fbank.%s.grad_norm' % kind])
        _after_close(enh, fx, 'fbank.%s.after.' % kind)


def test_enhance_gan_step(golden_dir):
    from oracle import trainers
    fx, cfg = _load(golden_dir, 'trainers_tiny.npz'), joint_cfg()
    enh, gan = trainers.leaf(_sub(fx, 'gan.enh.p.')), trainers.leaf(_sub(fx, 'gan.d.p.'))
    buf = trainers.buffers(_sub(fx, 'gan.d.p.'))
    out = trainers.enhance_gan_step(enh, gan, buf, joint.Adadelta(enh, eps=cfg['eps']), joint.Adadelta(gan, eps=cfg['eps']),
                                    torch.from_numpy(fx['fbank_W']), _tr_batch(fx), torch.from_numpy(fx['cmvn']), cfg)
    for k in ('loss', 'gan_loss', 'enhance_loss', 'loss_D'):
        np.testing.assert_allclose(out[k].numpy().reshape(-1), fx['gan.' + k], rtol=3e-4, err_msg=k)
    assert abs(out['grad_norm'] - float(fx['gan.grad_norm'])) < 2e-3 * float(fx['gan.grad_norm'])
    assert abs(out['grad_norm_D'] - float(fx['gan.grad_norm_D'])) < 2e-3 * float(fx['gan.grad_norm_D'])
    _after_close(enh, fx, 'gan.enh.after.')
    _after_close(gan, fx, 'gan.d.after.')
    for k, v in buf.items():
        np.testing.assert_allclose(v.numpy(), fx['gan.d.after.' + k], rtol=1e-4, atol=1e-6, err_msg=k)


def test_asr_step(golden_dir):
    from oracle import trainers
    fx, cfg = _load(golden_dir, 'trainers_tiny.npz'), joint_cfg()
    asr = trainers.leaf({k: v for k, v in _sub(fx, 'asr.p.').items() if not k.startswith('dec.att.')})
    out = trainers.asr_step(asr, joint.Adadelta(asr, eps=cfg['eps']), torch.from_numpy(fx['asr.feats']), torch.from_numpy(fx['targets']),
                            fx['lens'].tolist(), fx['tlens'].tolist(), cfg)
    for k in ('loss', 'loss_ctc', 'loss_att'):
        np.testing.assert_allclose(out[k].numpy().reshape(-1), fx['asr.' + k], rtol=3e-4, err_msg=k)
    assert abs(out['acc'] - float(fx['asr.acc'])) < 1e-9
    assert abs(out['grad_norm'] - float(fx['asr.grad_norm'])) < 2e-3 * float(fx['asr.grad_norm'])
    for n in ('enc.enc2.bt0.weight', 'dec.output.bias', 'ctc.ctc_lo.weight', 'att.mlp_enc.weight'):
        np.testing.assert_allclose(asr[n].detach().numpy(), fx['asr.after.' + n], rtol=1e-3, atol=2e-5, err_msg=n)


def test_joint_validate(golden_dir):
    from oracle import trainers
    fx, cfg = _load(golden_dir, 'trainers_tiny.npz'), joint_cfg()
    enh = _sub(fx, 'val.enh.')
    asr = {k: v for k, v in _sub(fx, 'val.asr.').items() if not k.startswith('dec.att.')}
    gan, buf = _sub(fx, 'val.gan.'), trainers.buffers(_sub(fx, 'val.gan.'))
    batch = (torch.from_numpy(fx['clean']), torch.from_numpy(fx['mix']), torch.from_numpy(fx['mix_log']), torch.from_numpy(fx['targets']),
             fx['lens'].tolist(), fx['tlens'].tolist())
    out = trainers.joint_validate(enh, asr, gan, buf, torch.from_numpy(fx['fbank_W']), batch, torch.from_numpy(fx['cmvn']), cfg)
    for k in ('loss', 'loss_ctc', 'loss_att', 'enhance_loss', 'gan_loss'):
        np.testing.assert_allclose(out[k].numpy().reshape(-1), fx['val.' + k], rtol=3e-4, err_msg=k)
    assert abs(out['acc'] - float(fx['val.acc'])) < 1e-9
    np.testing.assert_allclose(out['att_ws'].numpy(), fx['val.att_ws'], rtol=1e-3, atol=1e-6)
    for k, v in buf.items():
        np.testing.assert_allclose(v.numpy(), fx['val.gan_after.' + k], rtol=1e-4, atol=1e-6, err_msg=k)


def test_scheduled_sampling_forward_backward(golden_dir):
    """Rate 1.0: every decoder step i > 0 feeds back its own arg-max (e2e_decoder.py:123-127)."""
    fx, cfg = _load(golden_dir, 'trainers_tiny.npz'), joint_cfg()
    asr = {k: v.clone().requires_grad_(True) for k, v in _sub(fx, 'asr.p.').items() if v.dtype.is_floating_point and not k.startswith('dec.att.')}
    L1 = int(fx['tlens'].max()) + 1
    loss_ctc, loss_att, acc, _, _ = nets.e2e_forward(asr, torch.from_numpy(fx['asr.feats']), torch.from_numpy(fx['targets']), fx['lens'].tolist(),
                                                     fx['tlens'].tolist(), cfg['elayers'], cfg['mtlalpha'], sample_steps=[i > 0 for i in range(L1)])
    (cfg['mtlalpha'] * loss_ctc + (1 - cfg['mtlalpha']) * loss_att).backward()
    np.testing.assert_allclose(loss_att.detach().numpy().reshape(-1), fx['ss.loss_att'], rtol=3e-4)
    assert abs(acc - float(fx['ss.acc'])) < 1e-9
    for n in ('dec.embed.weight', 'dec.decoder.0.weight_ih', 'att.mlp_dec.weight', 'dec.output.weight', 'enc.enc2.bt0.weight'):
        ref = fx['ss.g.' + n]
        err = np.abs(asr[n].grad.numpy() - ref).max()
        assert err <= 1e-3 * np.abs(ref).max() + 1e-7, (n, err)


# ---- beam search (tests/golden/make_fixtures_recog.py) ----
RECOG_CONFIGS = [('att_b3', 3, 0.0, 0.0, 0.0, 0.0, 3), ('joint_b4', 4, 0.1, 0.3, 0.0, 0.0, 4), ('joint_ratio', 3, 0.0, 0.5, 0.6, 0.2, 2),
                 ('ctc_heavy', 2, 0.2, 0.9, 0.0, 0.0, 2)]


def check_nbest(got, fx, name, u, tol=2e-3):
    seqs, scores = fx['%s.u%d.yseq' % (name, u)], fx['%s.u%d.score' % (name, u)]
    assert len(got) == len(scores), (name, u, len(got), len(scores))
    for i, h in enumerate(got):
        want = [int(v) for v in seqs[i] if v >= 0]
        assert h['yseq'] == want, (name, u, i, h['yseq'], want)
        assert abs(h['score'] - scores[i]) <= tol * max(1.0, abs(scores[i])), (name, u, i, h['score'], scores[i])


def test_beam_search(golden_dir):
    from oracle import decode
    fx = _load(golden_dir, 'recog_tiny.npz')
    p = {k: v for k, v in _sub(fx, 'p.').items() if not k.startswith('dec.att.')}
    feats = torch.from_numpy(fx['feats'])
    for name, beam, penalty, ctcw, maxr, minr, nbest in RECOG_CONFIGS:
        for u, T in enumerate(fx['lens'].tolist()):
            got = decode.recognize(p, feats[u:u + 1, :T], 2, beam, penalty, ctcw, maxr, minr, nbest)
            check_nbest(got, fx, name, u)


# ---- frame subsampling and label smoothing (tests/golden/make_fixtures_n4.py) ----
def _e2e_grads_close(p, fx, pre, names):
    for n in names:
        ref = fx[pre + 'g.' + n]
        err = np.abs(p[n].grad.numpy() - ref).max()
        assert err <= 1e-3 * np.abs(ref).max() + 1e-7, (n, err)


def test_blstmp_subsampling(golden_dir):
    fx = _load(golden_dir, 'n4_tiny.npz')
    p = {k: v.clone().requires_grad_(True) for k, v in _sub(fx, 'sub.p.').items() if v.dtype.is_floating_point and not k.startswith('dec.att.')}
    feats, lens = torch.from_numpy(fx['feats']), fx['lens'].tolist()
    ys = nets.split_targets(torch.from_numpy(fx['targets']), fx['tlens'].tolist())
    hpad, hlens = nets.blstmp_forward(p, feats, lens, 3, pre='enc.enc1.', subsample=[1, 2, 2, 1, 1])
    assert list(hlens) == fx['sub.hlens'].tolist()
    np.testing.assert_allclose(hpad.detach().numpy(), fx['sub.hpad'], **TOL)
    loss_ctc = nets.ctc_forward(p, hpad, hlens, ys)
    loss_att, acc = nets.decoder_forward(p, hpad, hlens, ys, 11)
    np.testing.assert_allclose(loss_ctc.detach().numpy().reshape(-1), fx['sub.loss_ctc'], rtol=3e-4)
    np.testing.assert_allclose(loss_att.detach().numpy().reshape(-1), fx['sub.loss_att'], rtol=3e-4)
    (0.5 * loss_ctc + 0.5 * loss_att).backward()
    _e2e_grads_close(p, fx, 'sub.', ['enc.enc1.bilstm0.weight_ih_l0', 'enc.enc1.bt1.weight', 'enc.enc1.bilstm2.weight_hh_l0_reverse',
                                      'dec.output.weight', 'ctc.ctc_lo.weight'])


def test_blstmp_maxpooling_subsampling(golden_dir):
    fx = _load(golden_dir, 'n4_tiny.npz')
    p = {k: v.clone().requires_grad_(True) for k, v in _sub(fx, 'sub.p.').items() if v.dtype.is_floating_point and not k.startswith('dec.att.')}
    ys = nets.split_targets(torch.from_numpy(fx['targets']), fx['tlens'].tolist())
    hpad, hlens = nets.blstmp_forward(p, torch.from_numpy(fx['feats']), fx['lens'].tolist(), 3, pre='enc.enc1.', subsample=[1, 2, 2, 1, 1],
                                      subsample_type='maxpooling')
    assert list(hlens) == fx['mp.hlens'].tolist()
    np.testing.assert_allclose(hpad.detach().numpy(), fx['mp.hpad'], **TOL)
    loss_ctc = nets.ctc_forward(p, hpad, hlens, ys)
    loss_att, acc = nets.decoder_forward(p, hpad, hlens, ys, 11)
    np.testing.assert_allclose(loss_att.detach().numpy().reshape(-1), fx['mp.loss_att'], rtol=3e-4)
    (0.5 * loss_ctc + 0.5 * loss_att).backward()
    _e2e_grads_close(p, fx, 'mp.', ['enc.enc1.bilstm0.weight_ih_l0', 'enc.enc1.bt1.weight', 'enc.enc1.bilstm2.weight_hh_l0_reverse',
                                     'dec.output.weight', 'ctc.ctc_lo.weight'])


def test_label_smoothing(golden_dir):
    fx = _load(golden_dir, 'n4_tiny.npz')
    p = {k: v.clone().requires_grad_(True) for k, v in _sub(fx, 'lsm.p.').items() if v.dtype.is_floating_point and not k.startswith('dec.att.')}
    feats, lens = torch.from_numpy(fx['feats']), fx['lens'].tolist()
    ys = nets.split_targets(torch.from_numpy(fx['targets']), fx['tlens'].tolist())
    hpad, hlens = nets.encoder_forward(p, feats, lens, 2)
    loss_ctc = nets.ctc_forward(p, hpad, hlens, ys)
    loss_att, acc = nets.decoder_forward(p, hpad, hlens, ys, 11, labeldist=torch.from_numpy(fx['lsm.labeldist']), lsm_weight=0.1)
    np.testing.assert_allclose(loss_att.detach().numpy().reshape(-1), fx['lsm.loss_att'], rtol=3e-4)
    (0.5 * loss_ctc + 0.5 * loss_att).backward()
    _e2e_grads_close(p, fx, 'lsm.', ['dec.output.weight', 'dec.output.bias', 'dec.embed.weight', 'enc.enc2.bt0.weight'])


def _fp64_step(golden_dir, dtype):
    fx = _load(golden_dir, 'joint_tiny.npz')
    st = joint.JointState(_sub(fx, 'enh.'), _sub(fx, 'asr.'), _sub(fx, 'gan.'),
                          torch.from_numpy(_load(golden_dir, 'fbank_tiny.npz')['W']), joint_cfg(), dtype=dtype)
    t = lambda k: torch.from_numpy(fx[k]).to(dtype)
    batch = (t('clean'), t('mix'), t('mix_log'), torch.from_numpy(fx['targets']), fx['lens'].tolist(), fx['tlens'].tolist())
    return joint.joint_step(st, batch, t('cmvn'), update=False)


def _grad_dev(out, f64):
    """worst |g - g64| / max|g64| over all parameter gradients (tensors whose true gradient is 0 -- att.gvec.bias, softmax
    shift invariance -- are held to an absolute 1e-7 instead)."""
    worst = (0.0, None)
    for pre, gd in (('genh.', out['g_enh']), ('gasr.', out['g_asr']), ('ggan.', out['g_gan'])):
        for k, v in gd.items():
            ref = f64[pre + k]
            err, scale = np.abs(v.double().numpy() - ref).max(), np.abs(ref).max()
            if scale < 1e-12:
                assert err < 1e-7, (k, err)
                continue
            worst = max(worst, (err / scale, pre + k))
    return worst


def test_oracle_fp64_equals_reference_fp64(golden_dir):
    """The arbiter is pinned too: the oracle in double precision reproduces the REFERENCE modules run in double precision
    (tests/golden/make_fixtures_fp64.py) to 1e-8 -- losses, grad norms and every gradient of the composed step."""
    f64 = _load(golden_dir, 'joint_tiny_fp64.npz')
    out = _fp64_step(golden_dir, torch.float64)
    for k in ('loss', 'loss_ctc', 'loss_att', 'enhance_loss', 'coral_loss', 'gan_loss', 'loss_D'):
        assert abs(float(out[k]) - float(f64[k])) <= 1e-10 * max(1.0, abs(float(f64[k]))), k
    assert abs(out['grad_norm_asr'] - float(f64['grad_norm_asr'])) < 1e-9 * float(f64['grad_norm_asr'])
    assert abs(out['grad_norm_gan'] - float(f64['grad_norm_gan'])) < 1e-9 * float(f64['grad_norm_gan'])
    dev, name = _grad_dev(out, f64)
    assert dev < 1e-8, (dev, name)


def test_fp32_sides_within_1e3_of_fp64(golden_dir):
    """north_star's 1e-3 is stated against the exact result: the fp32 oracle AND the fp32 reference run (joint_tiny.npz)
    are each within 1e-3 of the double-precision run, tensor by tensor (they sit at <= 6e-4: fp32 rounding of a
    37-frame recurrence).  The GPU tests hold the HIP path to the same 1e-3 against the same double-precision vectors."""
    f64 = _load(golden_dir, 'joint_tiny_fp64.npz')
    fx = _load(golden_dir, 'joint_tiny.npz')
    dev, name = _grad_dev(_fp64_step(golden_dir, torch.float32), f64)
    assert dev < 1e-3, ('oracle fp32', dev, name)
    ref32 = {'g_enh': {}, 'g_asr': {}, 'g_gan': {}}
    for k, v in fx.items():
        for pre, d in (('genh.', 'g_enh'), ('gasr.', 'g_asr'), ('ggan.', 'g_gan')):
            if k.startswith(pre):
                ref32[d][k[len(pre):]] = torch.from_numpy(v)
    dev, name = _grad_dev(ref32, f64)
    assert dev < 1e-3, ('reference fp32', dev, name)


# ---- SURVEY 8(f) N4, round 2 (tests/golden/make_fixtures_n4b.py): --no_lsgan, blstmp enhancer, trainable fbank, dropout ----
def test_philox_known_answers():
    """Random123's published known-answer vectors for philox4x32-10: pins the generator the dropout masks are built on."""
    from oracle.philox import philox4x32_10, dropout_mask
    kat = [((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
           ((0xffffffff,) * 4, (0xffffffff, 0xffffffff), (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
           ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0), (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1))]
    for c, k, want in kat:
        got = philox4x32_10(np.array([c], np.uint32), np.array(k, np.uint32))[0]
        assert tuple(int(x) for x in got) == want
    m = dropout_mask(200001, 0.3, 99, 4)
    assert m.shape == (200001,) and set(np.unique(m).tolist()) == {0.0, float(np.float32(1.0) / (np.float32(1.0) - np.float32(0.3)))}
    assert abs((m > 0).mean() - 0.7) < 5e-3
    assert not np.array_equal(m, dropout_mask(200001, 0.3, 99, 5))            # another mask index, another mask


def test_bce_gan_loss(golden_dir):
    fx = _load(golden_dir, 'n4b_tiny.npz')
    full = _sub(fx, 'bce.p.')
    p = {k: v.clone().requires_grad_(True) for k, v in full.items() if v.dtype.is_floating_point and 'running' not in k}
    buf = {k: v.clone() for k, v in full.items() if 'running' in k or 'num_batches' in k}
    x = torch.from_numpy(fx['feats']).requires_grad_(True)
    d = nets.discriminator_forward(p, buf, x, use_sigmoid=True)
    np.testing.assert_allclose(d.detach().numpy(), fx['bce.d_out'], **TOL)
    lr = nets.gan_loss(d, True, use_lsgan=False)
    lf = nets.gan_loss(nets.discriminator_forward(p, buf, x * 0.9 + 0.1, use_sigmoid=True), False, use_lsgan=False)
    np.testing.assert_allclose(lr.detach().numpy().reshape(-1), fx['bce.l_real'], rtol=1e-4)
    np.testing.assert_allclose(lf.detach().numpy().reshape(-1), fx['bce.l_fake'], rtol=1e-4)
    ((lr + lf) * 0.5).backward()
    np.testing.assert_allclose(x.grad.numpy(), fx['bce.dx'], rtol=2e-3, atol=1e-7)
    for k, v in p.items():
        ref = fx['bce.g.' + k]
        assert np.abs(v.grad.numpy() - ref).max() <= 2e-3 * np.abs(ref).max() + 1e-7, k


def test_blstmp_enhancer(golden_dir):
    fx = _load(golden_dir, 'n4b_tiny.npz')
    p = {k: v.clone().requires_grad_(True) for k, v in _sub(fx, 'enhb.p.').items()}
    t = lambda k: torch.from_numpy(fx[k])
    lens = fx['lens'].tolist()
    out = nets.enhance_forward(p, t('mix'), t('mix_log'), lens, 2, kind='blstmp')
    np.testing.assert_allclose(out.detach().numpy(), fx['enhb.enhance_out'], **TOL)
    loss, out2 = nets.enhance_forward(p, t('mix'), t('mix_log'), lens, 2, t('clean'), t('cos'), kind='blstmp')
    np.testing.assert_allclose(loss.detach().numpy().reshape(-1), fx['enhb.l1_loss'], rtol=1e-4)
    (loss + (out2 * torch.linspace(0.5, 1.5, 257)).mean()).backward()
    for k, v in p.items():
        ref = fx['enhb.g.' + k]
        assert np.abs(v.grad.numpy() - ref).max() <= 2e-3 * np.abs(ref).max() + 1e-7, k


def test_trainable_fbank(golden_dir):
    fx = _load(golden_dir, 'n4b_tiny.npz')
    W = torch.from_numpy(fx['fbt.W']).requires_grad_(True)
    x = torch.from_numpy(fx['fbt.x']).requires_grad_(True)
    cm = torch.from_numpy(fx['cmvn'])
    np.testing.assert_allclose(nets.fbank_forward(x, W).detach().numpy(), fx['fbt.y_nocmvn'], rtol=1e-5, atol=1e-5)
    y1 = nets.fbank_forward(x, W, cm)
    np.testing.assert_allclose(y1.detach().numpy(), fx['fbt.y_cmvn'], rtol=1e-5, atol=1e-5)
    (y1 * torch.linspace(-1, 1, 80)).sum().backward()
    np.testing.assert_allclose(x.grad.numpy(), fx['fbt.dx'], rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(W.grad.numpy(), fx['fbt.dW'], rtol=1e-4, atol=1e-3 * np.abs(fx['fbt.dW']).max())


def test_ctc_dropout_matches_reference_run(golden_dir):
    """The reference's CTC head with F.dropout replaced by the recorded counter-based mask (make_fixtures_n4b.py)."""
    fx = _load(golden_dir, 'n4b_tiny.npz')
    p = {k: v.clone().requires_grad_(True) for k, v in _sub(fx, 'drop.p.').items() if v.dtype.is_floating_point and not k.startswith('dec.att.')}
    lc, la, acc, _, _ = nets.e2e_forward(p, torch.from_numpy(fx['feats']), torch.from_numpy(fx['drop.targets']), fx['lens'].tolist(),
                                         fx['drop.tlens'].tolist(), 2, ctc_dropout=(0.3, int(fx['drop.seed']), 0))
    np.testing.assert_allclose(lc.detach().numpy().reshape(-1), fx['drop.loss_ctc'], rtol=3e-4)
    np.testing.assert_allclose(la.detach().numpy().reshape(-1), fx['drop.loss_att'], rtol=3e-4)
    (0.5 * lc + 0.5 * la).backward()
    _e2e_grads_close(p, fx, 'drop.', ['ctc.ctc_lo.weight', 'ctc.ctc_lo.bias', 'enc.enc2.bt1.weight', 'enc.enc1.conv1_1.weight'])


def test_unet_enhancer(golden_dir):
    fx = _load(golden_dir, 'n4b_tiny.npz')
    full = _sub(fx, 'unet.p.')
    p = {k: v.clone().requires_grad_(True) for k, v in full.items() if v.dtype.is_floating_point and 'running' not in k}
    buf = {k: v.clone() for k, v in full.items() if 'running' in k}
    t = lambda k: torch.from_numpy(fx['unet.' + k])
    lens = fx['unet.lens'].tolist()
    out = nets.unet_enhance_forward(p, buf, t('mix'), t('mix_log'), lens)
    np.testing.assert_allclose(out.detach().numpy(), fx['unet.enhance_out'], rtol=1e-4, atol=1e-4 * np.abs(fx['unet.enhance_out']).max())
    loss, out2 = nets.unet_enhance_forward(p, buf, t('mix'), t('mix_log'), lens, clean=t('clean'), cos=t('cos'))
    np.testing.assert_allclose(loss.detach().numpy().reshape(-1), fx['unet.l1_loss'], rtol=1e-4)
    (loss + (out2 * torch.linspace(0.5, 1.5, 32)).mean()).backward()
    for k, v in p.items():
        if ('unet.g.' + k) in fx:
            ref = fx['unet.g.' + k]
            assert np.abs(v.grad.numpy() - ref).max() <= 2e-3 * np.abs(ref).max() + 1e-7, k
    for k, v in buf.items():            # two train-mode forwards moved the running statistics
        np.testing.assert_allclose(v.numpy(), fx['unet.after.' + k], rtol=1e-4, atol=1e-6, err_msg=k)
