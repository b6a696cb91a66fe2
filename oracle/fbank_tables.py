"""Kaldi-style 80-bin mel filterbank (TEST INFRASTRUCTURE, see oracle/__init__.py).

The reference hard-codes the matrix as literal tables (model/feat_model.py:15-16,19-33).  Those
literals are the output of Kaldi's MelBanks (num_bins=80, low_freq=20 Hz, high_freq=Nyquist,
16 kHz, 512-point FFT, mel(f)=1127 ln(1+f/700)) printed to 5-6 significant digits; the formula
below regenerates them (max |diff| vs the reference table < 6e-6, asserted against the golden
vector tests/golden/fbank_tiny.npz['W'])."""
import numpy as np


def mel_matrix(nfilt=80, nfft=512, sr=16000.0, low=20.0):
    """Returns W (nfft/2+1, nfilt) float32 such that fbank = power_spectrum @ W
    (FbankModel.fc layout, feat_model.py:106-107)."""
    mel = lambda f: 1127.0 * np.log(1.0 + f / 700.0)
    nb = nfft // 2
    ml, mh = mel(low), mel(sr / 2.0)
    d = (mh - ml) / (nfilt + 1)
    W = np.zeros((nb + 1, nfilt), np.float64)
    fm = mel(np.arange(nb) * sr / nfft)
    for j in range(nfilt):
        left, center, right = ml + j * d, ml + (j + 1) * d, ml + (j + 2) * d
        up = (fm - left) / (center - left)
        dn = (right - fm) / (right - center)
        w = np.where(fm <= center, up, dn)
        W[:nb, j] = np.where((fm > left) & (fm < right), w, 0.0)
    return W.astype(np.float32)
