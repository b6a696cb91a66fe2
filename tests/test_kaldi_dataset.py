"""MixKaldiDataset: directory layout, dictionary / label parsing, length buckets, dataset CMVN and the
two sample forms -- host tensors as upstream (data/mix_data_loader.py:198-237) and raw records decoded on the GPU."""
import argparse
import os

import numpy as np
import pytest
import torch

from robust_e2e_gan_amd.data import kaldi_io as kio
from robust_e2e_gan_amd.data.kaldi_dataset import MixKaldiDataset, read_dictionary, read_targets
from robust_e2e_gan_amd.data.mix_data_loader import BucketingSampler, _collate_fn


@pytest.fixture()
def data_dir(tmp_path):
    rng = np.random.default_rng(5)
    d = tmp_path / 'data'
    d.mkdir()
    lens = {'c1': 23, 'c2': 41, 'c3': 30}
    mixes = [('c1__n1', 'c1'), ('c2__n1', 'c2'), ('c3__n2', 'c3'), ('c1__n2', 'c1')]
    tables = {n: open(str(d / (n + '.ark')), 'wb') for n in ('clean_feats', 'clean_angles', 'mix_feats', 'mix_angles')}
    scp = {n: [] for n in tables}

    def put(name, key, mat, compressed):
        f = tables[name]
        off = f.tell() + len(key) + 1
        (kio.write_mat_compressed if compressed else kio.write_mat)(f, mat, key)
        scp[name].append('%s %s:%d' % (key, str(d / (name + '.ark')), off))

    for c, T in lens.items():
        put('clean_feats', c, (np.abs(rng.standard_normal((T, 9))) * 200).astype(np.float32), True)
        put('clean_angles', c, rng.uniform(-3.1, 3.1, (T, 9)).astype(np.float32), False)
    for m, c in mixes:
        spect = (np.abs(rng.standard_normal((lens[c], 9))) * 250).astype(np.float32)
        spect[0, 0] = 0.0
        put('mix_feats', m, spect, m.endswith('n1'))
        put('mix_angles', m, rng.uniform(-3.1, 3.1, (lens[c], 9)).astype(np.float32), False)
    for n, f in tables.items():
        f.close()
        (d / (n + '.scp')).write_text('\n'.join(scp[n]) + '\n')
    (d / 'utt2spk').write_text('c1 spkA\nc2 spkB\nc3 spkA\n')
    (d / 'text_char').write_text('c1 a b b\nc2 c a\nc3 a z c\n', encoding='utf-8')
    (tmp_path / 'dict.txt').write_text('a 1\nb 2\nc 3\n')
    args = argparse.Namespace(feat_type='kaldi_magspec', exp_path=str(tmp_path / 'exp'), normalize_type=1, num_utt_cmvn=10, model_unit='char')
    return args, str(d), str(tmp_path / 'dict.txt'), lens, mixes


def test_dictionary_and_targets(data_dir):
    args, d, dict_file, lens, mixes = data_dir
    cl = read_dictionary(dict_file)
    assert cl == ['<blank>', 'a', 'b', 'c', '<eos>']
    tg, dist = read_targets(os.path.join(d, 'text_char'), cl)
    assert tg == {'c1': [1, 2, 2], 'c2': [3, 1], 'c3': [1, 1, 3]}             # 'z' is out of vocabulary -> index 1
    assert dist[0] == 0 and abs(dist.sum() - 1) < 1e-6 and dist[4] == 3 / (4 + 2 + 2 + 3)


def test_dataset_host_samples_and_cmvn(data_dir):
    args, d, dict_file, lens, mixes = data_dir
    np.random.seed(0)
    ds = MixKaldiDataset(args, d, dict_file)
    assert len(ds) == 4 and ds.feat_size == 9 and ds.num_classes == 5
    assert os.path.exists(os.path.join(d, 'mix_kaldi_feat_len.scp')) and os.path.exists(os.path.join(args.exp_path, 'cmvn.npy'))
    assert sorted(i for v in ds.bins_to_samples.values() for i in v) == [0, 1, 2, 3]
    logs = []
    for _, p in ds.mix_feat_ids:
        s = kio.read_mat(p)
        logs.append(10 * np.log10(np.maximum(s, 1e-7)))
    allf = np.concatenate(logs, 0)
    np.testing.assert_allclose(ds.cmvn[0], -allf.mean(0), rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(ds.cmvn[1], 1 / allf.std(0), rtol=2e-3)
    utt, spk, clean, clean_log, mix, mix_log, cos, target = ds[1]
    assert (utt, spk) == ('c2__n1', 'spkB') and target.tolist() == [3, 1]
    assert clean.shape == (41, 9) and mix.shape == (41, 9) and cos.shape == (41, 9)
    assert float(mix.min()) == pytest.approx(1e-7)                             # in-place clamp reaches the linear stream
    np.testing.assert_allclose(mix_log.numpy(), (10 * np.log10(mix.numpy()) + ds.cmvn[0]) * ds.cmvn[1], rtol=1e-5, atol=1e-5)
    batches = list(BucketingSampler(ds.bins_to_samples, batch_size=2))
    assert sorted(i for b in batches for i in b) == [0, 1, 2, 3]


@pytest.mark.gpu
def test_raw_samples_collated_on_device_match_host_collate(data_dir):
    from robust_e2e_gan_amd.data.mix_data_loader import collate_kaldi_device
    args, d, dict_file, lens, mixes = data_dir
    np.random.seed(0)
    host = MixKaldiDataset(args, d, dict_file)
    raw = MixKaldiDataset(args, d, dict_file, raw=True)
    ref = _collate_fn([host[i] for i in range(4)])
    got = collate_kaldi_device([raw[i] for i in range(4)], 'cuda:0', torch.from_numpy(host.cmvn))
    assert got[0] == ref[0] and got[1] == ref[1]
    assert got[8].tolist() == ref[8].tolist() and got[9].tolist() == ref[9].tolist() and got[7].tolist() == ref[7].tolist()
    for k in (2, 3, 4, 5, 6):
        np.testing.assert_allclose(got[k].cpu().numpy(), ref[k].numpy(), rtol=1e-5, atol=3e-4, err_msg=str(k))
