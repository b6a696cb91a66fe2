"""Single-stream ASR collation (mirror of data/data_loader.py:236-265)."""
import torch


def _collate_fn(batch):
    """sample = (utt_id, spk_id, spect, log_spect, target) -> 7-tuple, sorted by length desc, zero padded."""
    batch = sorted(batch, key=lambda sample: sample[2].size(0), reverse=True)
    longest = batch[0][2]
    F_, B, T = longest.size(1), len(batch), longest.size(0)
    inputs, log_inputs = torch.zeros(B, T, F_), torch.zeros(B, T, F_)
    input_sizes, target_sizes = torch.IntTensor(B), torch.IntTensor(B)
    targets, utt_ids, spk_ids = [], [], []
    for x, s in enumerate(batch):
        utt_ids.append(s[0])
        spk_ids.append(s[1])
        n = s[2].size(0)
        inputs[x].narrow(0, 0, n).copy_(s[2])
        log_inputs[x].narrow(0, 0, n).copy_(s[3])
        input_sizes[x] = n
        target_sizes[x] = len(s[4])
        targets.extend(s[4])
    return utt_ids, spk_ids, inputs, log_inputs, torch.LongTensor(targets), input_sizes, target_sizes
