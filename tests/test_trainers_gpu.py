"""SURVEY 8(f) N1 on the GPU: the other four trainers' step functions, the validation pass (incl. the greedy
calculate_all_attentions) and scheduled sampling, against the vectors generated from the reference import
(tests/golden/trainers_tiny.npz, make_fixtures_trainers.py).  Through the C ABI; tolerance 1e-3."""
import random

import numpy as np
import pytest
import torch

from test_modules_gpu import DEV, _fx, _load, _opt, rel

pytestmark = pytest.mark.gpu


def _data(fx):
    t = lambda k: torch.from_numpy(fx[k])
    return (None, None, t('clean'), None, t('mix'), t('mix_log'), t('cos'), t('targets'), torch.IntTensor(fx['lens']), torch.IntTensor(fx['tlens']))


def _fbank(fx):
    from robust_e2e_gan_amd.model.feat_model import FbankModel
    fb = FbankModel(_opt())
    fb.load_state_dict({'fc': torch.from_numpy(fx['fbank_W'])})
    return fb.to(DEV).train()


def _after(m, fx, prefix, tol=1e-3, atol=2e-5):
    for k, v in m.state_dict().items():
        if v.dtype.is_floating_point:
            rel(prefix + k, v, fx[prefix + k], tol=tol, atol=atol)


def _scalar(out, key, ref, tol=1e-3):
    ref = float(np.asarray(ref).reshape(-1)[0])
    assert abs(out[key] - ref) <= tol * max(1.0, abs(ref)), (key, out[key], ref)


def test_enhance_base_trainer(golden_dir):
    from robust_e2e_gan_amd.joint_train import JointTrainer
    from robust_e2e_gan_amd.model.enhance_model import EnhanceModel
    from robust_e2e_gan_amd.trainers import EnhanceBaseTrainer
    fx = _fx(golden_dir, 'trainers_tiny.npz')
    enh = _load(EnhanceModel(_opt()), fx, 'base.p.')
    out = JointTrainer.to_floats(EnhanceBaseTrainer(_opt(), enh).step(_data(fx)))
    _scalar(out, 'train/loss', fx['base.loss'])
    _scalar(out, 'grad_norm', fx['base.grad_norm'], 2e-3)
    _after(enh, fx, 'base.after.')


@pytest.mark.parametrize('kind', ['L2', 'L1', 'smooth_L1'])
def test_enhance_fbank_trainer(golden_dir, kind):
    from robust_e2e_gan_amd.joint_train import JointTrainer
    from robust_e2e_gan_amd.model.enhance_model import EnhanceModel
    from robust_e2e_gan_amd.trainers import EnhanceFbankTrainer
    fx = _fx(golden_dir, 'trainers_tiny.npz')
    opt = _opt()
    opt.enhance_loss_type = kind
    enh = _load(EnhanceModel(opt), fx, 'fbank.p.')
    out = JointTrainer.to_floats(EnhanceFbankTrainer(opt, enh, _fbank(fx)).step(_data(fx)))
    _scalar(out, 'train/loss', fx['fbank.%s.loss' % kind])
    _scalar(out, 'grad_norm', fx['fbank.%s.grad_norm' % kind], 2e-3)
    _after(enh, fx, 'fbank.%s.after.' % kind)


def test_enhance_gan_trainer(golden_dir):
    from robust_e2e_gan_amd.joint_train import JointTrainer
    from robust_e2e_gan_amd.model.enhance_model import EnhanceModel
    from robust_e2e_gan_amd.model.gan_model import GANModel
    from robust_e2e_gan_amd.trainers import EnhanceGanTrainer
    fx = _fx(golden_dir, 'trainers_tiny.npz')
    opt = _opt()
    enh, gan = _load(EnhanceModel(opt), fx, 'gan.enh.p.'), _load(GANModel(opt), fx, 'gan.d.p.')
    out = JointTrainer.to_floats(EnhanceGanTrainer(opt, enh, _fbank(fx), gan).step(_data(fx), torch.from_numpy(fx['cmvn'])))
    for k in ('loss', 'gan_loss', 'enhance_loss', 'loss_D'):
        _scalar(out, 'train/' + k, fx['gan.' + k])
    _scalar(out, 'grad_norm', fx['gan.grad_norm'], 2e-3)
    _scalar(out, 'grad_norm_D', fx['gan.grad_norm_D'], 2e-3)
    _after(enh, fx, 'gan.enh.after.')
    _after(gan, fx, 'gan.d.after.')


def test_asr_trainer(golden_dir):
    from robust_e2e_gan_amd.joint_train import JointTrainer
    from robust_e2e_gan_amd.model.e2e_model import E2E
    from robust_e2e_gan_amd.trainers import AsrTrainer
    fx = _fx(golden_dir, 'trainers_tiny.npz')
    opt = _opt()
    asr = _load(E2E(opt), fx, 'asr.p.')
    data = (None, None, torch.from_numpy(fx['asr.feats']), torch.from_numpy(fx['targets']), torch.IntTensor(fx['lens']), torch.IntTensor(fx['tlens']))
    out = JointTrainer.to_floats(AsrTrainer(opt, asr).step(data, 0.0))
    for k in ('loss', 'loss_ctc', 'loss_att'):
        _scalar(out, 'train/' + k, fx['asr.' + k])
    assert abs(out['train/acc'] - float(fx['asr.acc'])) < 1e-6
    _scalar(out, 'grad_norm', fx['asr.grad_norm'], 2e-3)
    _after(asr, fx, 'asr.after.')


@pytest.mark.parametrize('which', ['base', 'fbank', 'gan', 'asr'])
def test_n1_trainers_streams_equal_single_stream(golden_dir, which):
    """Round 6: the four N1 trainers run their step on a stream of their own with the weight gradients (and EnhanceGanTrainer's whole D-step) on
    filler streams (trainers.StepStreams).  Same kernels in the same per-tensor order: the meters, every gradient and every updated parameter /
    buffer of the multi-stream step must equal the single-stream step's, and the routing must not leak out of the step."""
    from robust_e2e_gan_amd import ops
    from robust_e2e_gan_amd.joint_train import JointTrainer
    from robust_e2e_gan_amd.model.e2e_model import E2E
    from robust_e2e_gan_amd.model.enhance_model import EnhanceModel
    from robust_e2e_gan_amd.model.gan_model import GANModel
    from robust_e2e_gan_amd import trainers as T
    fx = _fx(golden_dir, 'trainers_tiny.npz')
    opt = _opt()
    res = {}
    for on in (True, False):
        if which == 'base':
            nets = [_load(EnhanceModel(opt), fx, 'base.p.')]
            tr, run = T.EnhanceBaseTrainer(opt, nets[0]), (lambda tr: tr.step(_data(fx)))
        elif which == 'fbank':
            nets = [_load(EnhanceModel(opt), fx, 'fbank.p.')]
            tr, run = T.EnhanceFbankTrainer(opt, nets[0], _fbank(fx)), (lambda tr: tr.step(_data(fx)))
        elif which == 'gan':
            nets = [_load(EnhanceModel(opt), fx, 'gan.enh.p.'), _load(GANModel(opt), fx, 'gan.d.p.')]
            tr, run = T.EnhanceGanTrainer(opt, nets[0], _fbank(fx), nets[1]), (lambda tr: tr.step(_data(fx), torch.from_numpy(fx['cmvn'])))
        else:
            nets = [_load(E2E(opt), fx, 'asr.p.')]
            data = (None, None, torch.from_numpy(fx['asr.feats']), torch.from_numpy(fx['targets']), torch.IntTensor(fx['lens']), torch.IntTensor(fx['tlens']))
            tr, run = T.AsrTrainer(opt, nets[0]), (lambda tr: tr.step(data, 0.0))
        assert tr.streams.on                       # the schedule is what runs by default on a GPU
        tr.streams.on = on
        outs = [JointTrainer.to_floats(run(tr)) for _ in range(2)]          # two steps: the second starts from the first one's update
        torch.cuda.synchronize()
        assert ops.MULTI_STREAM is False and ops.WGRAD_STREAM is None and ops.AUX_STREAM is None
        res[on] = (outs, {'%d.%s' % (i, k): v.detach().clone() for i, m in enumerate(nets) for k, v in m.state_dict().items()},
                   {'%d.%s' % (i, k): p.grad.detach().clone() for i, m in enumerate(nets) for k, p in m.named_parameters() if p.grad is not None})
    for a, b in zip(res[True][0], res[False][0]):
        for k in b:
            assert abs(a[k] - b[k]) <= 1e-6 * max(1.0, abs(b[k])), (k, a[k], b[k])
    for part in (1, 2):
        for k, ref in res[False][part].items():
            if ref.dtype.is_floating_point:
                rel(k, res[True][part][k], ref.cpu().numpy(), tol=2e-5, atol=1e-9)


def test_scheduled_sampling(golden_dir):
    """Rate 1.0: every step i > 0 feeds back the arg-max of its own previous output (e2e_decoder.py:123-127)."""
    from robust_e2e_gan_amd.model.e2e_model import E2E
    fx = _fx(golden_dir, 'trainers_tiny.npz')
    asr = _load(E2E(_opt()), fx, 'asr.p.')
    random.seed(0)
    lc, la, acc = asr(torch.from_numpy(fx['asr.feats']), torch.from_numpy(fx['targets']), torch.IntTensor(fx['lens']), torch.IntTensor(fx['tlens']), 1.0)
    rel('ss.loss_att', la.view(1), fx['ss.loss_att'])
    assert abs(float(acc) - float(fx['ss.acc'])) < 1e-6
    (0.5 * lc.view(()) + 0.5 * la).backward()
    named = dict(asr.named_parameters())
    for n in ('dec.embed.weight', 'dec.decoder.0.weight_ih', 'att.mlp_dec.weight', 'dec.output.weight', 'enc.enc2.bt0.weight'):
        rel('ss.g.' + n, named[n].grad, fx['ss.g.' + n], tol=1.5e-3)


def test_joint_validate(golden_dir):
    from robust_e2e_gan_amd.joint_train import JointTrainer
    from robust_e2e_gan_amd.model.enhance_model import EnhanceModel
    from robust_e2e_gan_amd.model.e2e_model import ShareE2E
    from robust_e2e_gan_amd.model.gan_model import GANModel
    fx = _fx(golden_dir, 'trainers_tiny.npz')
    opt = _opt()
    enh, asr, gan = _load(EnhanceModel(opt), fx, 'val.enh.'), _load(ShareE2E(opt), fx, 'val.asr.'), _load(GANModel(opt), fx, 'val.gan.')
    tr = JointTrainer(opt, enh, _fbank(fx), asr, gan)
    p0 = {k: v.clone() for k, v in asr.state_dict().items()}
    errs = tr.validate(_data(fx), torch.from_numpy(fx['cmvn']), want_attention=True)
    out = JointTrainer.to_floats(errs)
    for k in ('loss', 'loss_ctc', 'loss_att', 'enhance_loss', 'gan_loss'):
        _scalar(out, 'val/' + k, fx['val.' + k])
    assert abs(out['val/acc'] - float(fx['val.acc'])) < 1e-6
    rel('att_ws', errs['att_ws'], fx['val.att_ws'], tol=2e-3, atol=1e-6)
    assert enh.training and asr.training                        # modes restored
    assert all(p.grad is None or float(p.grad.abs().sum()) == 0.0 for p in asr.parameters())
    for k, v in asr.state_dict().items():
        assert torch.equal(v, p0[k])                            # validation does not touch parameters ...
    for k, v in gan.state_dict().items():                       # ... but D's BatchNorm statistics move (D stays in train mode)
        if 'running' in k or 'num_batches' in k:
            rel('val.gan_after.' + k, v, fx['val.gan_after.' + k], tol=1e-4)


def test_fit_loop_cadence(golden_dir, tmp_path):
    """JointTrainer.fit keeps the reference loop's cadence (joint_train.py:145-329): CMVN before training and after
    every validation, 'latest' checkpoints at print_freq, validation + model selection at validate_freq, the
    scheduled-sampling rate refreshed at validation time only, checkpoint keys as upstream."""
    from robust_e2e_gan_amd.joint_train import JointTrainer
    from robust_e2e_gan_amd.data.synthetic import make_batch
    from robust_e2e_gan_amd.model.enhance_model import EnhanceModel
    from robust_e2e_gan_amd.model.feat_model import FbankModel
    from robust_e2e_gan_amd.model.e2e_model import ShareE2E
    from robust_e2e_gan_amd.model.gan_model import GANModel
    opt = _opt()
    for k, v in dict(exp_path=str(tmp_path), epochs=2, shuffle_epoch=-1, print_freq=2, validate_freq=3, num_save_attention=2, criterion='acc',
                     eps_decay=0.01, sche_samp_start_iter=3, sche_samp_final_iter=6, sche_samp_final_rate=0.5, train_dataset_len=3,
                     num_utt_cmvn=3).items():
        setattr(opt, k, v)
    torch.manual_seed(7)
    enh, fb, asr, gan = (m.to(DEV).train() for m in (EnhanceModel(opt), FbankModel(opt), ShareE2E(opt), GANModel(opt)))

    def batch(seed):
        clean, mix, mix_log, targets, il, tl = make_batch(3, 40, 4, opt.odim, seed=seed)
        return (['u%d_%d' % (seed, i) for i in range(3)], None, clean, None, mix, mix_log, None, targets, il, tl)

    train_loader, val_loader = [batch(s) for s in (1, 2, 3)], [batch(9)]

    class Rec(object):
        def __init__(self):
            self.sets, self.prints, self.epochs, self.atts, self.resets = [], 0, 0, [], 0
            self.meters = {}

        def set_current_errors(self, e):
            self.sets.append(dict(e))
            for k, v in e.items():
                self.meters.setdefault(k, []).append(float(v))

        def get_current_errors(self, k):
            v = self.meters.get(k, [])
            return sum(v) / len(v) if v else 0.0

        def print_current_errors(self, epoch, iters):
            self.prints += 1

        def print_epoch_errors(self, epoch, iters):
            self.epochs += 1

        def plot_epoch_errors(self, epoch, iters, name):
            return {'file': name}

        def plot_attention(self, att_w, dec_len, enc_len, name):
            self.atts.append((att_w.shape, dec_len, enc_len, name))

        def reset(self):
            self.resets += 1
            self.meters = {}

    rec = Rec()
    tr = JointTrainer(opt, enh, fb, asr, gan)
    rates = []
    orig_step = tr.step
    tr.step = lambda data, rate, cmvn: (rates.append(rate), orig_step(data, rate, cmvn))[1]
    eps0 = tr.asr_optimizer.param_groups[0]['eps']
    iters, best_loss, best_acc = tr.fit(train_loader, val_loader, rec)
    assert iters == 6 and rec.prints == 3 and rec.epochs == 2 and rec.resets == 2
    # rate is refreshed only at validation time: iters 0-2 use update(0)=0, iters 3-5 use update(3)=0, afterwards update(6)=0.5
    assert rates == [0.0] * 6
    train_sets = [e for e in rec.sets if any(k.startswith('train/') for k in e)]
    assert len(train_sets) == 6 and set(train_sets[0]) == {'train/loss', 'train/loss_ctc', 'train/acc', 'train/loss_att', 'train/enhance_loss',
                                                             'train/coral_loss', 'train/loss_D', 'train/gan_loss'}
    val_sets = [e for e in rec.sets if any(k.startswith('val/') for k in e)]
    assert len(val_sets) == 2 and set(val_sets[0]) == {'val/loss', 'val/loss_ctc', 'val/acc', 'val/loss_att', 'val/enhance_loss', 'val/gan_loss'}
    assert len(rec.atts) == 4 and rec.atts[0][3] == 'u9_0_ep0_it3.png' and rec.atts[0][0][0] == 5      # Lmax + 1 decoder steps
    import os
    assert os.path.isfile(os.path.join(str(tmp_path), 'latest')) and os.path.isfile(os.path.join(str(tmp_path), 'enhance_cmvn.npy'))
    st = torch.load(os.path.join(str(tmp_path), 'latest'), weights_only=False)
    assert set(st) == {'asr_state_dict', 'fbank_state_dict', 'enhance_state_dict', 'gan_state_dict', 'opt', 'epoch', 'iters', 'eps', 'lr',
                       'best_loss', 'best_acc', 'acc_report', 'loss_report', 'dropout_state'}      # upstream's keys + the dropout mask stream
    assert st['iters'] == 6 and best_acc >= 0.0
    # either the first validation set the best accuracy (saved) or eps was decayed -- never both for one validation
    assert os.path.isfile(os.path.join(str(tmp_path), 'model.acc.best')) or tr.asr_optimizer.param_groups[0]['eps'] < eps0


# ---- beam search ----
def test_recognize_matches_reference_nbest(golden_dir):
    """E2E.recognize (all hypotheses of a position batched on the GPU) reproduces the n-best lists of the reference's
    one-hypothesis-at-a-time search for attention-only, joint CTC/attention, length-ratio and CTC-heavy configurations."""
    import argparse
    from test_oracle_golden import RECOG_CONFIGS, check_nbest
    from robust_e2e_gan_amd.model.e2e_model import E2E
    fx = _fx(golden_dir, 'recog_tiny.npz')
    asr = _load(E2E(_opt()), fx, 'p.')
    feats = torch.from_numpy(fx['feats'])
    for name, beam, penalty, ctcw, maxr, minr, nbest in RECOG_CONFIGS:
        args = argparse.Namespace(beam_size=beam, penalty=penalty, ctc_weight=ctcw, maxlenratio=maxr, minlenratio=minr, nbest=nbest, lm_weight=0.0)
        for u, T in enumerate(fx['lens'].tolist()):
            got = asr.recognize(feats[u:u + 1, :T], args, [str(i) for i in range(12)])
            check_nbest(got, fx, name, u)
    assert asr.training                                                  # mode restored


# ---- frame subsampling and label smoothing ----
def _e2e_case(golden_dir, pre, overrides, names, name='n4_tiny.npz'):
    import argparse
    from robust_e2e_gan_amd.model.e2e_model import E2E
    fx = _fx(golden_dir, name)
    opt = argparse.Namespace(**{**vars(_opt()), **overrides})
    asr = _load(E2E(opt), fx, pre + 'p.')
    feats = torch.from_numpy(fx['feats'])
    lens, tl = torch.IntTensor(fx['lens']), torch.IntTensor(fx['tlens'])
    hpad, hl = asr.enc(feats.to(DEV), lens)
    assert list(hl) == fx[pre + 'hlens'].tolist()
    rel(pre + 'hpad', hpad, fx[pre + 'hpad'])
    lc, la, acc = asr(feats, torch.from_numpy(fx['targets']), lens, tl, 0.0)
    rel(pre + 'loss_ctc', lc.view(1), fx[pre + 'loss_ctc'])
    rel(pre + 'loss_att', la.view(1), fx[pre + 'loss_att'])
    assert abs(float(acc) - float(fx[pre + 'acc'])) < 1e-6
    (0.5 * lc.view(()) + 0.5 * la.view(())).backward()
    named = dict(asr.named_parameters())
    for n in names:
        rel(pre + 'g.' + n, named[n].grad, fx[pre + 'g.' + n], tol=1.5e-3)


def test_blstmp_frame_subsampling(golden_dir):
    _e2e_case(golden_dir, 'sub.', dict(etype='blstmp', elayers=3, subsample='1_2_2_1_1'),
              ['enc.enc1.bilstm0.weight_ih_l0', 'enc.enc1.bt1.weight', 'enc.enc1.bilstm2.weight_hh_l0_reverse', 'dec.output.weight', 'ctc.ctc_lo.weight'])


def test_blstmp_maxpooling_subsampling(golden_dir):
    import numpy as _np
    fx = _fx(golden_dir, 'n4_tiny.npz')
    fx2 = {('mp.p.' + k[len('sub.p.'):] if k.startswith('sub.p.') else k): v for k, v in fx.items()}
    _np.savez('/tmp/n4_mp.npz', **fx2)
    _e2e_case('/tmp', 'mp.', dict(etype='blstmp', elayers=3, subsample='1_2_2_1_1', subsample_type='maxpooling'),
              ['enc.enc1.bilstm0.weight_ih_l0', 'enc.enc1.bt1.weight', 'enc.enc1.bilstm2.weight_hh_l0_reverse', 'dec.output.weight', 'ctc.ctc_lo.weight'],
              name='n4_mp.npz')


def test_label_smoothing(golden_dir):
    fx = _fx(golden_dir, 'n4_tiny.npz')
    _e2e_case(golden_dir, 'lsm.', dict(lsm_type='unigram', lsm_weight=0.1, labeldist=fx['lsm.labeldist']),
              ['dec.output.weight', 'dec.output.bias', 'dec.embed.weight', 'enc.enc2.bt0.weight'])


def test_pixel_discriminator(golden_dir):
    """GANModel(netD_type='pixel') (1x1 convs + BatchNorm, gan_model.py:98-116): logits, LSGAN loss, all gradients and the
    BatchNorm running statistics after two forwards."""
    import argparse
    from robust_e2e_gan_amd.model.gan_model import GANModel, GANLoss
    fx = _fx(golden_dir, 'n4_tiny.npz')
    opt = argparse.Namespace(**{**vars(_opt()), 'netD_type': 'pixel'})
    gan = _load(GANModel(opt), fx, 'pix.p.')
    crit = GANLoss(use_lsgan=True)
    x = torch.from_numpy(fx['pix.x']).to(DEV).requires_grad_(True)
    d = gan(x)
    rel('pix.d_out', d, fx['pix.d_out'])
    loss = (crit(d, True) + crit(gan(x * 0.9), False)) * 0.5
    rel('pix.loss', loss.view(1), fx['pix.loss'])
    loss.backward()
    rel('pix.dx', x.grad, fx['pix.dx'], tol=1.5e-3)
    for k, p in gan.named_parameters():
        rel('pix.g.' + k, p.grad, fx['pix.g.' + k], tol=1.5e-3)
    for k, v in gan.state_dict().items():
        if 'running' in k or 'num_batches' in k:
            rel('pix.after.' + k, v, fx['pix.after.' + k], tol=1e-4)


# ---- SURVEY 8(f) N4, round 2 (tests/golden/make_fixtures_n4b.py): --no_lsgan, blstmp enhancer, trainable fbank, dropout ----
def test_bce_gan_loss_no_lsgan(golden_dir):
    """--no_lsgan: Sigmoid-headed discriminator (gan_model.py:90-91,126) + nn.BCELoss (gan_model.py:157-160) against the
    reference: probabilities, both losses, every gradient, BatchNorm running statistics."""
    import argparse
    from robust_e2e_gan_amd.model.gan_model import GANModel, GANLoss
    fx = _fx(golden_dir, 'n4b_tiny.npz')
    opt = argparse.Namespace(**{**vars(_opt()), 'no_lsgan': True})
    gan = _load(GANModel(opt), fx, 'bce.p.')
    crit = GANLoss(use_lsgan=False)
    x = torch.from_numpy(fx['feats']).to(DEV).requires_grad_(True)
    d = gan(x)
    rel('bce.d_out', d, fx['bce.d_out'])
    lr = crit(d, True)
    lf = crit(gan(x * 0.9 + 0.1), False)
    rel('bce.l_real', lr.view(1), fx['bce.l_real'])
    rel('bce.l_fake', lf.view(1), fx['bce.l_fake'])
    ((lr + lf) * 0.5).backward()
    rel('bce.dx', x.grad, fx['bce.dx'], tol=1.5e-3)
    for k, p in gan.named_parameters():
        rel('bce.g.' + k, p.grad, fx['bce.g.' + k], tol=1.5e-3)
    for k, v in gan.state_dict().items():
        if 'running' in k or 'num_batches' in k:
            rel('bce.after.' + k, v, fx['bce.after.' + k], tol=1e-4)


def test_blstmp_enhancer(golden_dir):
    """EnhanceModel(enhance_type='blstmp') (enhance_model.py:90-93) against the reference: mask product, mask-L1 loss, grads."""
    import argparse
    from robust_e2e_gan_amd.model.enhance_model import EnhanceModel
    fx = _fx(golden_dir, 'n4b_tiny.npz')
    opt = argparse.Namespace(**{**vars(_opt()), 'enhance_type': 'blstmp', 'enhance_layers': 2, 'subsample': '1_1_1'})
    enh = _load(EnhanceModel(opt), fx, 'enhb.p.')
    t = lambda k: torch.from_numpy(fx[k])
    lens = torch.IntTensor(fx['lens'])
    out = enh(t('mix'), t('mix_log'), lens)
    rel('enhb.enhance_out', out, fx['enhb.enhance_out'])
    for b, l in enumerate(fx['lens']):
        assert (out[b, l:] == 0).all()
    loss, out2 = enh(t('mix'), t('mix_log'), lens, t('clean'), t('cos'))
    rel('enhb.l1_loss', loss.view(1), fx['enhb.l1_loss'])
    (loss + (out2 * torch.linspace(0.5, 1.5, 257).to(DEV)).mean()).backward()
    for k, p in enh.named_parameters():
        rel('enhb.g.' + k, p.grad, fx['enhb.g.' + k], tol=1.5e-3)
    spec = enh.calculate_all_specgram(t('mix'), t('mix_log'), lens)           # no length mask: padded rows = sigmoid(fc(tanh(bias)))
    assert spec.shape == out.shape and torch.isfinite(spec).all()
    assert torch.equal(spec[0, :int(fx['lens'][0])], out.detach()[0, :int(fx['lens'][0])])


def test_trainable_fbank(golden_dir):
    """FbankModel(fbank_opti_type='train') (feat_model.py:105-109): dense (257,80) parameter -- features, dx and dW."""
    import argparse
    from robust_e2e_gan_amd.model.feat_model import FbankModel
    fx = _fx(golden_dir, 'n4b_tiny.npz')
    opt = argparse.Namespace(**{**vars(_opt()), 'fbank_opti_type': 'train'})
    fb = FbankModel(opt)
    assert fb.fc.requires_grad
    fb.load_state_dict({'fc': torch.from_numpy(fx['fbt.W'])})
    fb = fb.to(DEV)
    x = torch.from_numpy(fx['fbt.x']).to(DEV).requires_grad_(True)
    cm = torch.from_numpy(fx['cmvn'])
    rel('fbt.y_nocmvn', fb(x), fx['fbt.y_nocmvn'], tol=1e-5)
    y1 = fb(x, cm)
    rel('fbt.y_cmvn', y1, fx['fbt.y_cmvn'], tol=1e-5)
    (y1 * torch.linspace(-1, 1, 80).to(DEV)).sum().backward()
    rel('fbt.dx', x.grad, fx['fbt.dx'], tol=1e-4)
    rel('fbt.dW', fb.fc.grad, fx['fbt.dW'], tol=1e-4)


def test_dropout_kernel_is_the_recorded_mask():
    """re2e_dropout against oracle/philox.py bit for bit (odd sizes, several rates / seeds / mask indices), and its backward
    (the same mask regenerated, nothing stored)."""
    from oracle.philox import dropout_mask
    from robust_e2e_gan_amd import ops
    for n, p, seed, call in ((1, 0.5, 1, 0), (7, 0.1, 0xFFFFFFFFFFFF, 3), (100003, 0.3, 20261003, 0), (4096, 0.9, 42, 4000000000)):
        x = torch.randn(n, generator=torch.Generator().manual_seed(n)).to(DEV).requires_grad_(True)
        ops.dropout_seed(seed, call)
        y = ops.dropout(x, p)
        m = torch.from_numpy(dropout_mask(n, p, seed, call)).to(DEV)
        assert torch.equal(y.detach(), x.detach() * m), (n, p)
        g = torch.ones(n, device=DEV) * 3.0
        y.backward(g)
        assert torch.equal(x.grad, g * m)
        assert ops.dropout_state() == (seed & 0xFFFFFFFFFFFFFFFF, (call + 1) & 0xFFFFFFFF)
    assert ops.dropout(x, 0.0) is x


def test_ctc_dropout_matches_reference_run(golden_dir):
    """dropout_rate = 0.3 through E2E: the CTC head's always-on F.dropout (e2e_ctc.py:51) with the recorded mask -- losses and
    gradients equal the reference's run with that mask; BLSTMP's per-layer nn.LSTM(dropout=) stays a no-op."""
    import argparse
    from robust_e2e_gan_amd import ops
    from robust_e2e_gan_amd.model.e2e_model import E2E
    fx = _fx(golden_dir, 'n4b_tiny.npz')
    opt = argparse.Namespace(**{**vars(_opt()), 'dropout_rate': 0.3})
    asr = _load(E2E(opt), fx, 'drop.p.')
    ops.dropout_seed(int(fx['drop.seed']), 0)
    lc, la, acc = asr(torch.from_numpy(fx['feats']), torch.from_numpy(fx['drop.targets']), torch.IntTensor(fx['lens']),
                      torch.IntTensor(fx['drop.tlens']), 0.0)
    assert ops.dropout_state()[1] == 1                                  # exactly one mask was drawn
    rel('drop.loss_ctc', lc.view(1), fx['drop.loss_ctc'])
    rel('drop.loss_att', la.view(1), fx['drop.loss_att'])
    (0.5 * lc.view(()) + 0.5 * la.view(())).backward()
    named = dict(asr.named_parameters())
    for n in ('ctc.ctc_lo.weight', 'ctc.ctc_lo.bias', 'enc.enc2.bt1.weight', 'enc.enc1.conv1_1.weight'):
        rel('drop.g.' + n, named[n].grad, fx['drop.g.' + n], tol=1.5e-3)
    asr.eval()                                                          # upstream's F.dropout ignores eval mode
    ops.dropout_seed(int(fx['drop.seed']), 0)
    with torch.no_grad():
        lc2, _, _ = asr(torch.from_numpy(fx['feats']), torch.from_numpy(fx['drop.targets']), torch.IntTensor(fx['lens']),
                        torch.IntTensor(fx['drop.tlens']), 0.0)
    assert torch.equal(lc2.view(1), lc.detach().view(1))


def test_blstm_interlayer_dropout_vs_oracle(golden_dir):
    """nn.LSTM(num_layers=2, dropout=p) of the enhancer's BLSTM (e2e_encoder.py:156-157): the mask falls on layer 0's output in
    training mode only.  torch's generator cannot be replayed inside nn.LSTM, so this site is pinned through the oracle with
    the recorded mask; eval mode must equal the dropout-free reference output."""
    import argparse
    from oracle import nets as on
    from oracle.philox import dropout_mask
    from robust_e2e_gan_amd import ops
    from robust_e2e_gan_amd.model.enhance_model import EnhanceModel
    fx = _fx(golden_dir, 'enhance_tiny.npz')
    opt = argparse.Namespace(**{**vars(_opt()), 'dropout_rate': 0.25})
    enh = _load(EnhanceModel(opt), fx, 'p.')
    t = lambda k: torch.from_numpy(fx[k])
    lens = fx['lens'].tolist()
    B, T, H2 = len(lens), max(lens), 2 * opt.enhance_units
    ops.dropout_seed(77, 5)
    out = enh(t('mix'), t('mix_log'), torch.IntTensor(lens))
    p = {k[2:]: torch.from_numpy(v).clone().requires_grad_(True) for k, v in fx.items() if k.startswith('p.')}
    mask = on.tm_mask(dropout_mask(T * B * H2, 0.25, 77, 5), B, T, H2)
    ref = on.enhance_forward(p, t('mix'), t('mix_log'), lens, 2, inter_masks=[mask])
    rel('enhance_out (dropout)', out, ref.detach().numpy())
    assert (out.detach().cpu() - t('enhance_out')).abs().max() > 1e-2 * t('enhance_out').abs().max()      # the mask did something
    (out * torch.linspace(0.5, 1.5, 257).to(DEV)).mean().backward()
    (ref * torch.linspace(0.5, 1.5, 257)).mean().backward()
    for k, q in enh.named_parameters():
        rel('g.' + k, q.grad, p[k].grad.numpy(), tol=1.5e-3)
    enh.eval()
    with torch.no_grad():
        rel('eval: no dropout', enh(t('mix'), t('mix_log'), torch.IntTensor(lens)), fx['enhance_out'])


def test_unet_enhancer(golden_dir):
    """EnhanceModel(enhance_type='unet_128') (pix2pix U-Net, enhance_model.py:224-303): stride-2 convolutions, transposed
    convolutions (fwd = the stride-2 data-gradient kernel, its gradients = the convolution's other two kernels), plain
    BatchNorm, channel concatenation, the double sigmoid -- mask product, mask-L1 loss, every gradient and the BatchNorm
    running statistics against the reference."""
    import argparse
    from robust_e2e_gan_amd.model.enhance_model import EnhanceModel
    fx = _fx(golden_dir, 'n4b_tiny.npz')
    opt = argparse.Namespace(**{**vars(_opt()), 'enhance_type': 'unet_128', 'idim': 32, 'enhance_input_nc': 1, 'enhance_output_nc': 1,
                                'enhance_ngf': 4, 'enhance_norm': 'batch'})
    enh = _load(EnhanceModel(opt), fx, 'unet.p.')
    t = lambda k: torch.from_numpy(fx['unet.' + k])
    lens = torch.IntTensor(fx['unet.lens'])
    out = enh(t('mix'), t('mix_log').unsqueeze(1), lens)                      # (B,1,T,F) as enhance_fbank_train.py:117 passes it
    rel('unet.enhance_out', out, fx['unet.enhance_out'])
    assert (out[1, int(fx['unet.lens'][1]):] == 0).all()
    loss, out2 = enh(t('mix'), t('mix_log').unsqueeze(1), lens, t('clean'), t('cos'))
    rel('unet.l1_loss', loss.view(1), fx['unet.l1_loss'])
    (loss + (out2 * torch.linspace(0.5, 1.5, 32).to(DEV)).mean()).backward()
    named = dict(enh.named_parameters())
    for k in fx:
        if k.startswith('unet.g.'):
            rel(k, named[k[len('unet.g.'):]].grad, fx[k], tol=1.5e-3)
    for k, v in enh.state_dict().items():
        if 'running' in k or 'num_batches' in k:
            rel('unet.after.' + k, v, fx['unet.after.' + k], tol=1e-4)
    with pytest.raises(Exception):
        enh(t('mix')[:, :50], t('mix_log')[:, :50].unsqueeze(1), lens)        # T not a multiple of 32: refused, as upstream's cat would fail


def test_fit_with_device_prefetcher_equals_collated_loader(tmp_path):
    """JointTrainer.fit(prefetch=True) on UN-collated batches (lists of ragged samples through data.prefetch.DevicePrefetcher: pinned
    staging, H2D + re2e_pack_pad on a copy stream, event hand-off) ends with exactly the parameters of the same run on the
    host-collated 10-tuples of mix_data_loader._collate_fn -- the input side moved onto the device changes no value."""
    from robust_e2e_gan_amd.joint_train import JointTrainer
    from robust_e2e_gan_amd.data.synthetic import make_batch
    from robust_e2e_gan_amd.data.mix_data_loader import _collate_fn
    from robust_e2e_gan_amd.model.enhance_model import EnhanceModel
    from robust_e2e_gan_amd.model.feat_model import FbankModel
    from robust_e2e_gan_amd.model.e2e_model import ShareE2E
    from robust_e2e_gan_amd.model.gan_model import GANModel

    class Quiet(object):
        def __getattr__(self, name):
            return (lambda *a, **k: 0.0) if name.startswith('get') else (lambda *a, **k: {'file': 'x'} if name.startswith('plot_epoch') else None)

    def samples(seed):
        clean, mix, mix_log, targets, il, tl = make_batch(3, 40, 4, 30, seed=seed)
        return [('u%d_%d' % (seed, i), 's', clean[i, :l].clone(), clean[i, :l].clone(), mix[i, :l].clone(), mix_log[i, :l].clone(),
                 mix[i, :l].clone(), targets[4 * i:4 * i + 4].clone()) for i, l in enumerate(il.tolist())]
    raw = [samples(s) for s in (1, 2, 3, 4)]
    finals = []
    for prefetch in (False, True):
        opt = _opt()
        for k, v in dict(exp_path=str(tmp_path / ('p%d' % prefetch)), epochs=1, shuffle_epoch=-1, print_freq=100, validate_freq=100,
                         num_save_attention=0, criterion='acc', eps_decay=0.01, sche_samp_start_iter=10 ** 9, sche_samp_final_iter=2 * 10 ** 9,
                         sche_samp_final_rate=0.5, train_dataset_len=3, num_utt_cmvn=3, odim=30).items():
            setattr(opt, k, v)
        opt.char_list = [str(i) for i in range(30)]
        torch.manual_seed(7)
        enh, fb, asr, gan = (m.to(DEV).train() for m in (EnhanceModel(opt), FbankModel(opt), ShareE2E(opt), GANModel(opt)))
        tr = JointTrainer(opt, enh, fb, asr, gan)
        loader = raw if prefetch else [_collate_fn(list(b)) for b in raw]
        iters, _, _ = tr.fit(loader, [], Quiet(), prefetch=prefetch)
        assert iters == 4
        torch.cuda.synchronize()
        finals.append({k: v.clone() for m in (enh, asr, gan) for k, v in m.state_dict().items()})
    for k, v in finals[0].items():
        assert torch.equal(v, finals[1][k]), k


def test_fit_repeats_a_step_whose_recurrence_was_aborted(tmp_path):
    """A persistent recurrence that gives up poisons its outputs; the device-side step gate refuses the update (D's included) and HOLDS the
    step that was enqueued behind it.  JointTrainer.fit must then repeat the aborted step with the launch-per-step recurrences on the state it
    first ran on, run the held step again, count the repeat, and end where an undisturbed run ends -- BatchNorm running statistics included:
    four steps on four DIFFERENT batches (an update applied out of order would show), the second one aborted by the test hook
    re2e_debug_force_abort; then the same with the abort in the LAST step (nothing enqueued behind it) and with the step behind a repeat aborted too."""
    from robust_e2e_gan_amd import lib
    from robust_e2e_gan_amd.data.synthetic import make_batch
    from robust_e2e_gan_amd.joint_train import JointTrainer
    from robust_e2e_gan_amd.model.enhance_model import EnhanceModel
    from robust_e2e_gan_amd.model.feat_model import FbankModel
    from robust_e2e_gan_amd.model.e2e_model import ShareE2E
    from robust_e2e_gan_amd.model.gan_model import GANModel
    import __graft_entry__ as g

    class Quiet(object):
        def __getattr__(self, name):
            return lambda *a, **k: 0.0

    def run(abort_at):
        opt = g._tiny_opt()
        for k, v in dict(exp_path=str(tmp_path), print_freq=100, validate_freq=100, epochs=1, shuffle_epoch=100, criterion='acc', eps_decay=0.01,
                         sche_samp_start_iter=300, sche_samp_final_iter=600, sche_samp_final_rate=0.0, train_dataset_len=3, num_utt_cmvn=3).items():
            setattr(opt, k, v)
        torch.manual_seed(11)
        enh, fb, asr, gan = (m.to(DEV).train() for m in (EnhanceModel(opt), FbankModel(opt), ShareE2E(opt), GANModel(opt)))
        batches = []
        for i in range(4):
            clean, mix, mix_log, targets, il, tl = make_batch(3, 40, 4, opt.odim, seed=5 + i)
            batches.append((['u%d' % j for j in range(3)], None, clean, None, mix, mix_log, None, targets, il, tl))
        tr = JointTrainer(opt, enh, fb, asr, gan)
        base = lib.query('re2e_lstm_abort_count')             # (the counter is per process; the trainer acknowledges what it finds at its first step)
        n = [0]
        orig = tr.step

        def step(data, rate, cmvn):
            n[0] += 1
            if n[0] in abort_at:
                lib.query('re2e_debug_force_abort', 1)         # the next persistent forward sequence "gives up"
            return orig(data, rate, cmvn)
        tr.step = step
        iters, _, _ = tr.fit(batches, [], Quiet())
        torch.cuda.synchronize()
        assert iters == 4
        state = {k: v.detach().clone() for m in (enh, asr, gan) for k, v in m.state_dict().items()}
        return tr, state, n[0], lib.query('re2e_lstm_abort_count') - base
    try:
        _, ref, calls, aborts = run(())
        assert calls == 4 and aborts == 0
        # (calls: the aborted step, the held one behind it, the repeat, the held one again)
        for abort_at, want_calls, want_aborts, want_rec in (((2,), 6, 1, 1), ((4,), 5, 1, 1), ((2, 5), 8, 2, 2)):
            tr, got, calls, aborts = run(abort_at)
            assert aborts == want_aborts and tr.recovered_steps == want_rec and calls == want_calls, (abort_at, aborts, tr.recovered_steps, calls)
            assert tr.unexplained_aborts == 0
            for k in ref:
                assert torch.isfinite(got[k].float()).all(), k
                rel('%r %s' % (abort_at, k), got[k].float(), ref[k].float().cpu().numpy(), tol=2e-4, atol=1e-6)
    finally:
        lib.query('re2e_debug_force_abort', 0)


def test_fit_repeats_a_validation_pass_or_cmvn_estimate_that_was_aborted(tmp_path):
    """The validation pass and the CMVN estimate run the persistent recurrences behind no step gate: a give-up there must not reach the model
    selection as NaN scores, nor sit in the device counter until the next training step's gate refuses that step on one replica only.
    JointTrainer.run_guarded reads the counter after the pass, repeats the pass with the launch-per-step kernels on the state it first ran on
    (D's BatchNorm buffers, the CMVN accumulators) and acknowledges: scores, CMVN and final weights equal an undisturbed run's."""
    from robust_e2e_gan_amd import lib
    from robust_e2e_gan_amd.data.synthetic import make_batch
    from robust_e2e_gan_amd.joint_train import JointTrainer
    from robust_e2e_gan_amd.model.enhance_model import EnhanceModel
    from robust_e2e_gan_amd.model.feat_model import FbankModel
    from robust_e2e_gan_amd.model.e2e_model import ShareE2E
    from robust_e2e_gan_amd.model.gan_model import GANModel
    import __graft_entry__ as g

    class Rec(object):
        def __init__(self):
            self.vals = []

        def set_current_errors(self, e):
            self.vals.append(dict(e))

        def __getattr__(self, name):
            return lambda *a, **k: 0.0

    def run(abort_validate_call, abort_enhancer_call):
        opt = g._tiny_opt()
        for k, v in dict(exp_path=str(tmp_path), print_freq=100, validate_freq=2, epochs=1, shuffle_epoch=100, criterion='acc', eps_decay=0.01,
                         sche_samp_start_iter=300, sche_samp_final_iter=600, sche_samp_final_rate=0.0, train_dataset_len=3, num_utt_cmvn=3,
                         num_save_attention=0).items():
            setattr(opt, k, v)
        torch.manual_seed(11)
        enh, fb, asr, gan = (m.to(DEV).train() for m in (EnhanceModel(opt), FbankModel(opt), ShareE2E(opt), GANModel(opt)))
        batches = []
        for i in range(5):
            clean, mix, mix_log, targets, il, tl = make_batch(3, 40, 4, opt.odim, seed=5 + i)
            batches.append((['u%d' % j for j in range(3)], None, clean, None, mix, mix_log, None, targets, il, tl))
        tr = JointTrainer(opt, enh, fb, asr, gan)
        nv, orig_v = [0], tr.validate

        def validate(data, cmvn, want_attention=False):
            nv[0] += 1
            if nv[0] == abort_validate_call:
                lib.query('re2e_debug_force_abort', 1)
            return orig_v(data, cmvn, want_attention=want_attention)
        tr.validate = validate
        ne, orig_e = [0], enh.forward

        def enh_forward(*a, **k):
            ne[0] += 1
            if ne[0] == abort_enhancer_call:
                lib.query('re2e_debug_force_abort', 1)
            return orig_e(*a, **k)
        enh.forward = enh_forward
        vis = Rec()
        iters, _, best_acc = tr.fit(batches[:4], batches[4:], vis)
        torch.cuda.synchronize()
        assert iters == 4 and tr.recovered_steps == 0 and tr.unexplained_aborts == 0
        state = {k: v.detach().clone() for m in (enh, asr, gan) for k, v in m.state_dict().items()}
        vals = [v for v in vis.vals if any(k.startswith('val/') for k in v)]
        return tr, state, vals, fb.fbank_cmvn.copy()
    try:
        _, ref, vref, cref = run(0, 0)
        assert len(vref) == 2 and all(np.isfinite(list(v.values())).all() for v in vref)
        # (a) the first validation pass (its enhancer recurrence) gives up; (b) the very first CMVN estimate does (enhancer call 1, before any step)
        for av, ae in ((1, 0), (0, 1)):
            tr, got, vgot, cgot = run(av, ae)
            assert tr.guarded_repeats == 1, (av, ae, tr.guarded_repeats)
            assert len(vgot) == len(vref)
            for a, b in zip(vgot, vref):
                for k in b:
                    assert np.isfinite(a[k]) and abs(a[k] - b[k]) <= 2e-4 * max(1.0, abs(b[k])), (av, ae, k, a[k], b[k])
            assert np.isfinite(cgot).all() and np.abs(cgot - cref).max() <= 2e-4 * np.abs(cref).max()
            for k in ref:
                assert torch.isfinite(got[k].float()).all(), k
                rel('%r %s' % ((av, ae), k), got[k].float(), ref[k].float().cpu().numpy(), tol=2e-4, atol=1e-6)
    finally:
        lib.query('re2e_debug_force_abort', 0)


# ---- InstanceNorm variants against vectors from the reference import (tests/golden/make_fixtures_n4c.py, round 4) ----
def test_instance_norm_discriminator_vs_reference(golden_dir):
    """--norm_D instance (gan_model.py:42-46,57): InstanceNorm2d(affine=False) between biased convolutions, against the reference's own run:
    output, LSGAN real / fake losses, input gradient and every parameter gradient.  (A bias in front of an InstanceNorm has an exactly-zero
    gradient: both sides hold rounding noise there, so those are held to the network's largest gradient, not their own.)"""
    import argparse
    from robust_e2e_gan_amd.model.gan_model import GANModel, GANLoss
    fx = _fx(golden_dir, 'n4c_tiny.npz')
    opt = argparse.Namespace(**{**vars(_opt()), 'norm_D': 'instance'})
    gan = _load(GANModel(opt), fx, 'ind.p.')
    assert not any('running' in k for k in gan.state_dict())
    crit = GANLoss(use_lsgan=True)
    x = torch.from_numpy(fx['feats']).to(DEV).requires_grad_(True)
    d = gan(x)
    rel('ind.d_out', d, fx['ind.d_out'])
    lr = crit(d, True)
    lf = crit(gan(x * 0.9 + 0.1), False)
    rel('ind.l_real', lr.view(1), fx['ind.l_real'])
    rel('ind.l_fake', lf.view(1), fx['ind.l_fake'])
    ((lr + lf) * 0.5).backward()
    rel('ind.dx', x.grad, fx['ind.dx'], tol=1.5e-3)
    gscale = max(float(np.abs(fx[k]).max()) for k in fx if k.startswith('ind.g.'))
    for k, p in gan.named_parameters():
        ref = fx['ind.g.' + k]
        err = float(np.abs(p.grad.detach().cpu().numpy() - ref).max())
        assert err <= 1.5e-3 * max(float(np.abs(ref).max()), 1e-2 * gscale), (k, err)


def test_instance_norm_unet_vs_reference(golden_dir):
    """--enhance_norm instance (enhance_model.py:258-261): the pix2pix U-Net with InstanceNorm2d(affine=False) and biased convolutions against
    the reference's own run: mask product, mask-L1 loss, every gradient."""
    import argparse
    from robust_e2e_gan_amd.model.enhance_model import EnhanceModel
    fx = _fx(golden_dir, 'n4c_tiny.npz')
    opt = argparse.Namespace(**{**vars(_opt()), 'enhance_type': 'unet_128', 'idim': 32, 'enhance_input_nc': 1, 'enhance_output_nc': 1,
                                'enhance_ngf': 4, 'enhance_norm': 'instance'})
    enh = _load(EnhanceModel(opt), fx, 'inu.p.')
    assert not any('running' in k for k in enh.state_dict())
    t = lambda k: torch.from_numpy(fx['inu.' + k])
    lens = torch.IntTensor(fx['inu.lens'])
    out = enh(t('mix'), t('mix_log').unsqueeze(1), lens)
    rel('inu.enhance_out', out, fx['inu.enhance_out'])
    loss, out2 = enh(t('mix'), t('mix_log').unsqueeze(1), lens, t('clean'), t('cos'))
    rel('inu.l1_loss', loss.view(1), fx['inu.l1_loss'])
    (loss + (out2 * torch.linspace(0.5, 1.5, 32).to(DEV)).mean()).backward()
    named = dict(enh.named_parameters())
    gscale = max(float(np.abs(fx[k]).max()) for k in fx if k.startswith('inu.g.'))
    n = 0
    for k in fx:
        if k.startswith('inu.g.'):
            g = named[k[len('inu.g.'):]].grad
            err = float(np.abs(g.detach().cpu().numpy() - fx[k]).max())
            assert err <= 1.5e-3 * max(float(np.abs(fx[k]).max()), 1e-2 * gscale), (k, err)
            n += 1
    assert n >= len(named) - 1          # (the reference leaves one parameter without a gradient: not in the fixture)
