"""The data-parallel branches of JointTrainer.step on ONE GPU: a 1-rank RCCL group with ``dist.world_size`` reporting 2,
so every all-reduce of the three flat gradient buffers really executes (over one rank: AVG = identity) on RCCL's own
stream, issued from the streams the trainer issues them from.  Whatever ordering bug lets a collective start before the
gradient kernels it reduces have finished shows up as a difference against the non-DP step."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


@pytest.fixture(scope='module')
def one_rank_group():
    import torch.distributed as dist
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    torch.cuda.set_device(0)
    dist.init_process_group('nccl', init_method='tcp://127.0.0.1:%d' % (29700 + os.getpid() % 200), rank=0, world_size=1)
    yield dist
    dist.destroy_process_group()


def _nets(opt, fx, W):
    from robust_e2e_gan_amd.model.enhance_model import EnhanceModel
    from robust_e2e_gan_amd.model.feat_model import FbankModel
    from robust_e2e_gan_amd.model.e2e_model import ShareE2E
    from robust_e2e_gan_amd.model.gan_model import GANModel
    torch.manual_seed(7)
    enh, asr, gan, fb = EnhanceModel(opt), ShareE2E(opt), GANModel(opt), FbankModel(opt)
    fb.load_state_dict({'fc': torch.from_numpy(W)})
    return [m.to(DEV).train() for m in (enh, fb, asr, gan)]


@pytest.mark.parametrize('etype,isgan', [('vggblstmp', True), ('blstmp', True), ('vggblstmp', False), ('blstmp', False)])
def test_dp_branches_equal_single_process_step(golden_dir, one_rank_group, monkeypatch, etype, isgan):
    import __graft_entry__ as g
    from robust_e2e_gan_amd import dist as rdist
    from robust_e2e_gan_amd.joint_train import JointTrainer
    fx = dict(np.load(os.path.join(golden_dir, 'joint_tiny.npz')))
    W = dict(np.load(os.path.join(golden_dir, 'fbank_tiny.npz')))['W']
    t = lambda k: torch.from_numpy(fx[k])
    data = (None, None, t('clean'), None, t('mix'), t('mix_log'), None, t('targets'), torch.IntTensor(fx['lens']), torch.IntTensor(fx['tlens']))
    res = {}
    calls = []
    real = rdist.allreduce_mean_

    def counting(flat, async_op=False):
        calls.append(flat.numel())
        return real(flat, async_op=async_op)
    for dp in (False, True):
        opt = g._tiny_opt()
        opt.etype, opt.isGAN = etype, isgan
        opt.coral_loss_lambda = 2e6
        enh, fb, asr, gan = _nets(opt, fx, W)
        tr = JointTrainer(opt, enh, fb, asr, gan if isgan else None)
        with monkeypatch.context() as mp:
            if dp:
                mp.setattr(rdist, 'world_size', lambda: 2)          # take the DP branches; the group itself has one rank
                mp.setattr(rdist, 'allreduce_mean_', counting)
            outs = [JointTrainer.to_floats(tr.step(data, 0.0, t('cmvn'))) for _ in range(3)]
        torch.cuda.synchronize()
        res[dp] = (outs, {n + '.' + k: v.clone() for n, m in (('enh', enh), ('asr', asr), ('gan', gan)) for k, v in m.state_dict().items()})
    # 3 steps x (ASR + enhancer [+ D]) all-reduces really ran
    nets = 3 if isgan else 2
    assert len(calls) == 3 * nets, calls
    for a, b in zip(res[False][0], res[True][0]):
        assert a == b, (a, b)                                        # bitwise: AVG over one rank is the identity
    for k, v in res[False][1].items():
        assert torch.equal(v, res[True][1][k]), k
