"""Batch collation for the joint trainer (mirror of data/mix_data_loader.py:264-346).

``_collate_fn`` keeps the reference's host-side semantics (sort by length desc, zero padding,
IntTensor sizes, flat LongTensor targets).  ``collate_device`` is the MI355X-first variant (K1):
the ragged utterances are concatenated once on the host, shipped with ONE H2D copy per stream and
zero-padded on the device by re2e_pack_pad -- 3 live tensors instead of 5 (cos_angles and
clean_log_inputs are never read by joint_train.py)."""
import numpy as np
import torch

from ..lib import call


def _collate_fn(batch):
    """sample = (utt_id, spk_id, clean, clean_log, mix, mix_log, cos_angle, target)"""
    batch = sorted(batch, key=lambda sample: sample[2].size(0), reverse=True)
    longest = batch[0][2]
    F_, B, T = longest.size(1), len(batch), longest.size(0)
    outs = [torch.zeros(B, T, F_) for _ in range(5)]
    input_sizes = torch.IntTensor(B)
    target_sizes = torch.IntTensor(B)
    targets, utt_ids, spk_ids = [], [], []
    for x, s in enumerate(batch):
        utt_ids.append(s[0])
        spk_ids.append(s[1])
        n = s[2].size(0)
        for k in range(5):
            outs[k][x].narrow(0, 0, n).copy_(s[2 + k])
        input_sizes[x] = n
        target_sizes[x] = len(s[7])
        targets.extend(s[7])
    return (utt_ids, spk_ids, outs[0], outs[1], outs[2], outs[3], outs[4], torch.LongTensor(targets), input_sizes, target_sizes)


def pack_pad_device(flat_dev, lens, Tmax):
    """(sum T_i, F) device rows -> zero padded (B, Tmax, F) on the device (re2e_pack_pad)."""
    B, F_ = len(lens), flat_dev.shape[1]
    dev = flat_dev.device
    off = np.concatenate([[0], np.cumsum(lens)[:-1]]).astype(np.int32)
    out = torch.empty(B, Tmax, F_, dtype=torch.float32, device=dev)
    off_d = torch.from_numpy(off).to(dev)                 # keep both index tensors alive across the launch
    len_d = torch.tensor(lens, dtype=torch.int32, device=dev)
    call('re2e_pack_pad', flat_dev.data_ptr(), off_d.data_ptr(), len_d.data_ptr(), B, Tmax, F_, out.data_ptr())
    return out


def collate_device(batch, device, streams=(2, 4, 5)):
    """Device-side collate: returns the same 10-tuple as ``_collate_fn`` with the selected streams
    (default clean, mix, mix_log) padded on ``device``; the unused ones are None."""
    batch = sorted(batch, key=lambda sample: sample[2].size(0), reverse=True)
    lens = [int(s[2].size(0)) for s in batch]
    T = lens[0]
    outs = [None] * 5
    for k in streams:
        flat = torch.cat([s[k] for s in batch], 0).pin_memory() if torch.cuda.is_available() else torch.cat([s[k] for s in batch], 0)
        outs[k - 2] = pack_pad_device(flat.to(device, non_blocking=True), lens, T)
    targets = torch.LongTensor([t for s in batch for t in s[7]])
    return ([s[0] for s in batch], [s[1] for s in batch], outs[0], outs[1], outs[2], outs[3], outs[4], targets, torch.IntTensor(lens),
            torch.IntTensor([len(s[7]) for s in batch]))


def decode_pad_device(raws, device, Tmax=None, cmvn=None, want_log=False):
    """List of ``kaldi_io.RawMat`` (one per utterance, equal column counts) -> zero padded (B,Tmax,F) fp32 on ``device``
    (and, with ``want_log``, the normalised ``(10*log10(max(x,1e-7)) + cmvn[0]) * cmvn[1]`` tensor): the record bytes go
    over PCIe as ONE pinned blob and are decoded by re2e_kaldi_decode_pad.  'DM' records are converted to 'FM' first."""
    B, F_ = len(raws), raws[0].cols
    lens = [r.rows for r in raws]
    Tmax = max(lens) if Tmax is None else Tmax
    chunks, offs, kinds, pos = [], [], [], 0
    for r in raws:
        if r.cols != F_:
            raise ValueError('records of one stream must have the same number of columns')
        if r.kind == 'CM':
            body, kind = r.payload, 2
        else:                                    # data part only, as fp32
            data = r.payload if r.kind == 'FM' else np.frombuffer(r.payload.tobytes(), '<f8').astype('<f4').view(np.uint8)
            body, kind = data, 0
        pad = (-pos) % 16
        if pad:
            chunks.append(np.zeros(pad, np.uint8))
            pos += pad
        offs.append(pos)
        kinds.append(kind)
        chunks.append(np.ascontiguousarray(body))
        pos += body.size
    blob = torch.from_numpy(np.concatenate(chunks))
    on_gpu = torch.device(device).type == 'cuda'
    to = (lambda t: t.pin_memory().to(device, non_blocking=True)) if on_gpu else (lambda t: t.to(device))
    blob_d, off_d = to(blob), to(torch.tensor(offs, dtype=torch.int64))
    kind_d, len_d = to(torch.tensor(kinds, dtype=torch.int32)), to(torch.tensor(lens, dtype=torch.int32))
    out = torch.empty(B, Tmax, F_, dtype=torch.float32, device=device)
    out_log = torch.empty_like(out) if want_log else None
    cm = cmvn.to(device).float().contiguous() if cmvn is not None else None
    call('re2e_kaldi_decode_pad', blob_d.data_ptr(), off_d.data_ptr(), kind_d.data_ptr(), len_d.data_ptr(), B, Tmax, F_, out.data_ptr(),
         out_log.data_ptr() if want_log else None, cm.data_ptr() if cm is not None else None)
    for t in (blob_d, off_d, kind_d, len_d):      # uploaded on this stream, read by the kernel just launched
        t.record_stream(torch.cuda.current_stream()) if on_gpu else None
    return (out, out_log) if want_log else out


def collate_kaldi_device(batch, device, cmvn=None):
    """Device-side collate for Kaldi-table datasets.  ``batch`` = list of
    ``(utt_id, spk_id, clean RawMat, mix RawMat, clean_angle RawMat, mix_angle RawMat, target list)`` (angles may be
    None).  Returns the reference's 10-tuple (mix_data_loader.py:264-302) with all five feature streams on ``device``:
    linear spectra clamped at 1e-7, log spectra CMVN-normalised with the dataset ``cmvn`` (2,F) and
    ``cos_angles = cos(clean_angle - mix_angle)`` (:231), everything sorted by length (descending) and zero padded."""
    batch = sorted(batch, key=lambda s: s[3].rows, reverse=True)
    lens = [s[3].rows for s in batch]
    T = lens[0]
    clean, clean_log = decode_pad_device([s[2] for s in batch], device, T, cmvn, want_log=True)
    mix, mix_log = decode_pad_device([s[3] for s in batch], device, T, cmvn, want_log=True)
    cos = None
    if batch[0][4] is not None:
        ca, ma = decode_pad_device([s[4] for s in batch], device, T), decode_pad_device([s[5] for s in batch], device, T)
        cos = torch.cos(ca - ma)          # data-pipeline glue, as numpy in the reference; padded frames give cos(0) = 1 ...
        cos = ops_mask_rows(cos, lens)    # ... which the reference's zero padding does not have
    targets = torch.LongTensor([t for s in batch for t in s[6]])
    return ([s[0] for s in batch], [s[1] for s in batch], clean, clean_log, mix, mix_log, cos, targets, torch.IntTensor(lens),
            torch.IntTensor([len(s[6]) for s in batch]))


def ops_mask_rows(x, lens):
    from .. import ops
    from ..model.e2e_common import lens_dev
    with torch.no_grad():
        return ops.mask_rows(x, lens_dev(lens, x.device))


class BucketingSampler(object):
    """data/mix_data_loader.py:314-346: batches of similarly sized utterances from length bins."""

    def __init__(self, bins_to_samples, batch_size=1):
        self.bins_to_samples = bins_to_samples
        self.batch_size = batch_size
        self.bins = self.build_bins()

    def build_bins(self):
        ids = []
        for _, sample_idx in self.bins_to_samples.items():
            sample_idx = list(sample_idx)
            np.random.shuffle(sample_idx)
            ids.extend(sample_idx)
        return [ids[i:i + self.batch_size] for i in range(0, len(ids), self.batch_size)]

    def __iter__(self):
        for ids in self.bins:
            np.random.shuffle(ids)
            yield ids

    def __len__(self):
        return len(self.bins)

    def shuffle(self, epoch):
        self.bins = self.build_bins()
        np.random.shuffle(self.bins)
