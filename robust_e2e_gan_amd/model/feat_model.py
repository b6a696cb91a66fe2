"""FbankModel (mirror of model/feat_model.py:93-135 and compute_cmvn :62-90)."""
import numpy as np
import torch

from .. import ops
from ..lib import call, Re2eError
from .e2e_common import ModelBase, to_cuda, lens_list, lens_dev


def mel_matrix(nfilt=80, nfft=512, sr=16000.0, low=20.0):
    """(nfft/2+1, nfilt) mel filterbank.  The reference stores it as literal tables
    (feat_model.py:15-16,19-33: Kaldi MelBanks, 80 bins, 20 Hz..Nyquist, 16 kHz, 512-point FFT,
    printed to 5-6 digits); this regenerates them from the formula (|diff| <= 1.1e-5).  Loading a
    reference checkpoint overwrites ``fc`` with the table values through load_state_dict."""
    mel = lambda f: 1127.0 * np.log(1.0 + f / 700.0)
    nb = nfft // 2
    ml, mh = mel(low), mel(sr / 2.0)
    d = (mh - ml) / (nfilt + 1)
    W = np.zeros((nb + 1, nfilt), np.float64)
    fm = mel(np.arange(nb) * sr / nfft)
    for j in range(nfilt):
        left, center, right = ml + j * d, ml + (j + 1) * d, ml + (j + 2) * d
        w = np.where(fm <= center, (fm - left) / (center - left), (right - fm) / (right - center))
        W[:nb, j] = np.where((fm > left) & (fm < right), w, 0.0)
    return W.astype(np.float32)


def _runs(Wn, what, limit):
    """Per column of ``Wn``: first non-zero row, run length (first to last non-zero) and the run's values."""
    n = Wn.shape[1]
    offs, lens = [], []
    for j in range(n):
        nz = np.nonzero(Wn[:, j])[0]
        offs.append(int(nz[0]) if len(nz) else 0)
        lens.append(int(nz[-1] - nz[0] + 1) if len(nz) else 0)
    width = max(max(lens), 1)
    if width > limit:
        raise Re2eError('filterbank is not banded (max %d %s > %d): the banded kernels (re2e_fbank_fwd / _bwd) cover the frozen mel '
                        'matrix only; a dense matrix runs through the trainable path (--fbank-opti-type train: x^2 W on the GEMM engine)'
                        % (width, what, limit))
    taps = np.zeros((n, width), np.float32)
    for j in range(n):
        taps[j, :lens[j]] = Wn[offs[j]:offs[j] + lens[j], j]
    return offs, lens, taps, width


def band_from_matrix(W, device):
    """Banded forms of the (F, NF) filterbank for re2e_fbank_*: by filter (offsets, lengths, taps, maxw), NF, and by bin -- the
    filters covering a bin are a contiguous run too -- (offsets, lengths, weights, maxc) for the backward kernel."""
    Wn = W.detach().cpu().numpy() if isinstance(W, torch.Tensor) else np.asarray(W)
    F, NF = Wn.shape
    offs, lens, taps, maxw = _runs(Wn, 'taps per filter', 32)
    toffs, tlens, tw, maxc = _runs(Wn.T, 'filters per bin', 128)
    dev_i = lambda v: torch.tensor(v, dtype=torch.int32, device=device)
    return (dev_i(offs), dev_i(lens), torch.from_numpy(taps).to(device), maxw, NF, dev_i(toffs), dev_i(tlens), torch.from_numpy(tw).to(device), maxc)


class FbankModel(ModelBase):
    def __init__(self, args):
        super(FbankModel, self).__init__()
        self.opt = args
        idim, odim = args.idim, args.fbank_dim
        if odim != 80:
            raise Re2eError('nfilt must = 80, but get {}'.format(odim))     # feat_model.py:23-33
        self.fc = torch.nn.Parameter(torch.from_numpy(mel_matrix(odim)).clone())
        assert self.fc.shape[0] == idim, 'idim must be 257'
        # 'frozen': the fixed mel matrix is banded (501 non-zeros) -> gather kernel; otherwise ('train', feat_model.py:108-109)
        # it is a trainable dense (257,80) parameter -> GEMM path with dW (ops.fbank_dense)
        self.trainable = getattr(args, 'fbank_opti_type', 'frozen') != 'frozen'
        if not self.trainable:
            self.fc.requires_grad_(False)
        self.sum = np.zeros([1, odim], np.float32)
        self.sum_sq = np.zeros([1, odim], np.float32)
        self.fbank_cmvn = np.zeros([2, odim], np.float32)
        self.cmvn_num = min(args.train_dataset_len, args.num_utt_cmvn)
        self.cmvn_processed_num = 0
        self.frame_count = 0
        self._band = None

    def _apply(self, fn, *a, **k):
        self._band = None
        return super()._apply(fn, *a, **k)

    def load_state_dict(self, *a, **k):
        self._band = None
        return super().load_state_dict(*a, **k)

    def band(self):
        if self._band is None or self._band[0].device != self.fc.device:
            self._band = band_from_matrix(self.fc, self.fc.device)
        return self._band

    def forward(self, xs, fbank_cmvn=None):
        """log(max((x^2) W, 1e-7)) [-> (y + cmvn[0]) * cmvn[1]]   (feat_model.py:118-135)"""
        xs = to_cuda(self, xs)
        if self.trainable:
            return ops.fbank_dense(xs, self.fc, to_cuda(self, fbank_cmvn).float().contiguous() if fbank_cmvn is not None else None)
        if fbank_cmvn is None:
            return ops.fbank(xs, self.band(), None, True, False)[0]
        fbank_cmvn = to_cuda(self, fbank_cmvn).float().contiguous()
        return ops.fbank(xs, self.band(), fbank_cmvn, False, True)[1]

    def forward_both(self, xs, fbank_cmvn):
        """One pass producing (raw, normalised) features -- used by the fused joint step."""
        xs = to_cuda(self, xs)
        if self.trainable:
            return self.forward(xs), self.forward(xs, fbank_cmvn)
        return ops.fbank(xs, self.band(), to_cuda(self, fbank_cmvn).float().contiguous(), True, True)

    def compute_cmvn(self, inputs, input_sizes):
        """feat_model.py:62-90: running sum / sum-of-squares over valid frames; returns None until
        ``cmvn_num`` utterances were seen, then [-mean; 1/sqrt(var)] (2,80) on the next call."""
        with torch.no_grad():
            feats = self.forward(inputs)
        if self.cmvn_processed_num < self.cmvn_num:
            B, T, NF = feats.shape
            s, q = torch.empty(NF, device=feats.device), torch.empty(NF, device=feats.device)
            ld = lens_dev(input_sizes, feats.device)
            call('re2e_cmvn_stats', feats.data_ptr(), ld.data_ptr(), B, T, NF, s.data_ptr(), q.data_ptr())
            self.sum = np.add(self.sum, s.cpu().numpy())
            self.sum_sq = np.add(self.sum_sq, q.cpu().numpy())
            ll = lens_list(input_sizes)
            self.frame_count += int(sum(ll))
            self.cmvn_processed_num += len(ll)
            return None
        mean = self.sum / self.frame_count
        var = self.sum_sq / self.frame_count - np.square(mean)
        self.fbank_cmvn[0, :] = -mean
        self.fbank_cmvn[1, :] = 1 / np.sqrt(var)
        return self.fbank_cmvn
