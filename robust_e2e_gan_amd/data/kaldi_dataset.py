"""Kaldi-table dataset of the joint / enhancement trainers (mirror of MixSequentialDataset,
data/mix_data_loader.py:14-262, for ``feat_type`` 'kaldi_magspec' / 'kaldi_powspec' -- the wav front-ends need
librosa / python_speech_features and are out of scope).

Directory layout as upstream (:31-35, :127-133): ``clean_feats.scp``, ``clean_angles.scp``, ``mix_feats.scp``,
``mix_angles.scp``, ``utt2spk``, ``text_char`` | ``text_word`` and a dictionary file with one symbol per line;
``mix_kaldi_feat_len.scp`` (frame counts) is created on first use (:63-66).  A mixture ``<clean-id>__<noise...>`` maps
to its clean utterance by the part before ``__`` (:219).

``__getitem__`` returns either the reference's 8-tuple of host tensors (``raw=False``; decoded, clamped, log + CMVN on
the host exactly as upstream) or the undecoded records for ``mix_data_loader.collate_kaldi_device`` (``raw=True``), which
decodes, clamps, takes the log, normalises and pads on the GPU."""
import os
from collections import defaultdict

import numpy as np
import torch

from . import kaldi_io

MAP_OOV = 1          # data/audioparse.py: out-of-vocabulary symbols map to index 1


def read_dictionary(dict_file):
    """char_list = ['<blank>'] + first column of the dictionary + ['<eos>']  (mix_data_loader.py:134-140)."""
    with open(dict_file, 'r', encoding='utf-8') as f:
        chars = [line.split(' ')[0].strip() for line in f if line.strip()]
    return ['<blank>'] + chars + ['<eos>']


def read_targets(label_file, char_list, model_unit='char'):
    """utt -> label ids (data/audioparse.py:88-124) and the smoothed label distribution used by label smoothing."""
    idx = {c: i for i, c in enumerate(char_list)}
    odim = len(char_list)
    count = np.zeros(odim)
    targets, n_utt = {}, 0
    with open(label_file, 'r', encoding='utf-8') as f:
        for line in f:
            parts = line.strip().replace('\t', ' ').split(' ')
            if not parts or not parts[0]:
                continue
            text = ''.join(parts[1:])          # 'word' units tokenise to the same symbol sequence, word by word
            tok = [idx.get(ch, MAP_OOV) for ch in text]
            if tok:
                for t in tok:
                    count[t] += 1
                n_utt += 1
            targets[parts[0]] = tok
    count[odim - 1] = n_utt
    count[count == 0] = 1
    count[0] = 0
    return targets, (count / count.sum()).astype(np.float32)


def _read_table(path):
    with open(path, encoding='utf-8') as f:
        return [line.strip().split(' ', 1) for line in f if line.strip()]


class MixKaldiDataset(torch.utils.data.Dataset):
    def __init__(self, args, data_dir, dict_file, raw=False):
        self.args, self.raw = args, raw
        self.feat_type = getattr(args, 'feat_type', 'kaldi_magspec')
        if not self.feat_type.startswith('kaldi'):
            raise ValueError('MixKaldiDataset reads Kaldi tables only (feat_type kaldi_magspec / kaldi_powspec)')
        self.clean_feats = dict(_read_table(os.path.join(data_dir, 'clean_feats.scp')))
        self.clean_angles = dict(_read_table(os.path.join(data_dir, 'clean_angles.scp')))
        self.mix_feat_ids = _read_table(os.path.join(data_dir, 'mix_feats.scp'))
        self.mix_angles = dict(_read_table(os.path.join(data_dir, 'mix_angles.scp')))
        len_scp = os.path.join(data_dir, 'mix_kaldi_feat_len.scp')
        if not os.path.exists(len_scp):                                    # :63-66, :264-281
            with open(len_scp, 'w') as f:
                for utt, path in self.mix_feat_ids:
                    f.write('%s %d\n' % (utt, kaldi_io.read_mat_raw(path).rows))
        self.feat_len = {k: int(v) for k, v in _read_table(len_scp)}
        lengths = [self.feat_len[utt] for utt, _ in self.mix_feat_ids]
        _, edges = np.histogram(lengths, bins='auto')                       # :68-73 length buckets for BucketingSampler
        self.bins_to_samples = defaultdict(list)
        for i, b in enumerate(np.digitize(lengths, bins=edges)):
            self.bins_to_samples[int(b)].append(i)
        self.feat_size = kaldi_io.read_mat_raw(self.mix_feat_ids[0][1]).cols
        self.utt2spk = dict(_read_table(os.path.join(data_dir, 'utt2spk')))
        self.char_list = read_dictionary(dict_file)
        self.num_classes = len(self.char_list)
        unit = getattr(args, 'model_unit', 'char')
        self.targets, self.labeldist = read_targets(os.path.join(data_dir, 'text_char' if unit == 'char' else 'text_word'), self.char_list, unit)
        self.num_utt_cmvn = getattr(args, 'num_utt_cmvn', 20000)
        self.cmvn = self._load_cmvn() if getattr(args, 'normalize_type', 1) == 1 else None

    # ---- features -------------------------------------------------------------------------------------------
    def _spect(self, path):
        m = kaldi_io.read_mat(path)
        return np.square(m) if self.feat_type == 'kaldi_powspec' else m     # data/audioparse.py:407-415

    def compute_cmvn(self):
        """[-mean; 1/sqrt(var)] of 10*log10(max(spect,1e-7)) over ``num_utt_cmvn`` random mixtures (:161-197)."""
        n = min(len(self.mix_feat_ids), self.num_utt_cmvn)
        s = np.zeros((1, self.feat_size), np.float32)
        sq = np.zeros((1, self.feat_size), np.float32)
        frames = 0
        for i in np.random.permutation(len(self.mix_feat_ids))[:n]:
            spect = self._spect(self.mix_feat_ids[i][1])
            spect[spect <= 1e-7] = 1e-7
            f = 10 * np.log10(spect)
            s = s + f.sum(0)
            sq = sq + np.square(f).sum(0)
            frames += f.shape[0]
        mean = s / frames
        var = sq / frames - np.square(mean)
        return np.concatenate([-mean, 1 / np.sqrt(var)], 0).astype(np.float32)

    def _load_cmvn(self):
        exp = getattr(self.args, 'exp_path', None)
        path = os.path.join(exp, 'cmvn.npy') if exp else None
        if path and os.path.exists(path):
            cm = np.load(path)
            if cm.shape[1] == self.feat_size:
                return cm
        cm = self.compute_cmvn()
        if path:
            os.makedirs(exp, exist_ok=True)
            np.save(path, cm)
        return cm

    # ---- samples --------------------------------------------------------------------------------------------
    def __len__(self):
        return len(self.mix_feat_ids)

    def _host_pair(self, path):
        spect = self._spect(path)
        spect[spect <= 1e-7] = 1e-7                                         # in place: the linear stream carries the clamp
        log = 10 * np.log10(spect)
        if self.cmvn is not None:
            log = (log + self.cmvn[0, :]) * self.cmvn[1, :]
        return spect, log.astype(np.float32)

    def __getitem__(self, index):
        mix_id, mix_path = self.mix_feat_ids[index]
        clean_id = mix_id.split('__')[0]
        target = self.targets.get(clean_id)
        if self.raw:
            if self.feat_type == 'kaldi_powspec':
                raise ValueError('raw mode decodes magnitudes on the GPU; squaring (kaldi_powspec) is host-only')
            return (mix_id, self.utt2spk[clean_id], kaldi_io.read_mat_raw(self.clean_feats[clean_id]), kaldi_io.read_mat_raw(mix_path),
                    kaldi_io.read_mat_raw(self.clean_angles[clean_id]), kaldi_io.read_mat_raw(self.mix_angles[mix_id]), list(target))
        mix_spect, mix_log = self._host_pair(mix_path)
        clean_spect, clean_log = self._host_pair(self.clean_feats[clean_id])
        cos = np.cos(kaldi_io.read_mat(self.clean_angles[clean_id]) - kaldi_io.read_mat(self.mix_angles[mix_id]))
        F = torch.FloatTensor
        return (mix_id, self.utt2spk[clean_id], F(clean_spect), F(clean_log), F(mix_spect), F(mix_log), F(cos), torch.LongTensor(target))
