"""Kaldi matrix record decoding + the loader's log/CMVN transform on CPU (TEST INFRASTRUCTURE, see oracle/__init__.py).

Restates /root/reference data/kaldi_io.py:376-456 (binary 'FM ' / 'DM ' / 'CM ' matrix records; the 8-bit
CompressedMatrix format 1 of kaldi: global min/range, four uint16 percentiles per column, column-major payload with a
three-segment piecewise-linear code) and data/mix_data_loader.py:198-203 + data/audioparse.py:446-459 (clamp at 1e-7,
10*log10, (x + cmvn[0]) * cmvn[1]).  Pinned by tests/golden/kaldi_tiny.npz, whose expected matrices were produced by the
reference's own reader."""
import struct

import numpy as np


def decode_record(buf):
    """bytes of one binary matrix record starting at the type token ('FM ', 'DM ' or 'CM ') -> float32 (rows, cols)."""
    token = buf[:3].decode('latin1')
    if token in ('FM ', 'DM '):
        _, rows, _, cols = struct.unpack('<bibi', buf[3:13])
        dt = '<f4' if token == 'FM ' else '<f8'
        return np.frombuffer(buf[13:13 + rows * cols * int(dt[2])], dt).reshape(rows, cols).astype(np.float32)
    assert token == 'CM ', token
    gmin, grange, rows, cols = struct.unpack('<ffii', buf[3:19])
    gmin, grange = np.float32(gmin), np.float32(grange)        # the reference keeps the header fields as float32 scalars
    hdr = np.frombuffer(buf[19:19 + 8 * cols], '<u2').reshape(cols, 4)
    data = np.frombuffer(buf[19 + 8 * cols:19 + 8 * cols + rows * cols], np.uint8).reshape(cols, rows)
    out = np.empty((cols, rows), np.float32)
    for c in range(cols):                       # kaldi_io.py:421-451, column by column
        p0, p25, p75, p100 = [np.float32(gmin + grange * np.float32(1.52590218966964e-05) * np.float32(v)) for v in hdr[c]]
        v = data[c]
        lo, mid, hi = v <= 64, (v > 64) & (v <= 192), v > 192
        out[c][lo] = p0 + (p25 - p0) / 64. * v[lo]
        out[c][mid] = p25 + (p75 - p25) / 128. * (v[mid] - 64)
        out[c][hi] = p75 + (p100 - p75) / 63. * (v[hi] - 192)
    return out.T.copy()


def loader_streams(mats, cmvn=None):
    """List of (T_i, F) spectra -> (linear, log) zero padded (B,Tmax,F) as the dataset + collate produce them
    (mix_data_loader.py:198-203: the clamp is in place, so the linear stream carries it too; :264-302 padding)."""
    B, T, F_ = len(mats), max(m.shape[0] for m in mats), mats[0].shape[1]
    lin, log = np.zeros((B, T, F_), np.float32), np.zeros((B, T, F_), np.float32)
    for b, m in enumerate(mats):
        s = np.maximum(m.astype(np.float32), np.float32(1e-7))
        l = 10 * np.log10(s)
        if cmvn is not None:
            l = (l + cmvn[0]) * cmvn[1]
        lin[b, :m.shape[0]], log[b, :m.shape[0]] = s, l
    return lin, log
