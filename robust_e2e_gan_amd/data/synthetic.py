"""Synthetic AISHELL-shaped batches: identical on the CPU-oracle and GPU paths."""
import torch

CONFIGS = {   # name: (B, Tmax, L, joint?)
    'config1': (4, 200, 10), 'config2': (16, 500, 25), 'config3': (32, 800, 40), 'config4': (32, 800, 40), 'config5': (8, 3000, 150),
}


def lengths(B, Tmax):
    if B == 1:
        return [Tmax]
    return [int(round(Tmax * (1 - 0.3 * i / (B - 1)))) for i in range(B)]


def make_batch(B=32, Tmax=800, L=40, V=4233, F_=257, seed=1234):
    """Returns (clean, mix, mix_log, targets, input_sizes, target_sizes) as CPU tensors.
    clean = |N+iN|*300 (int16-scale STFT magnitudes), mix = |clean_c + noise| at 10 dB SNR, zero beyond T_i;
    mix_log = per-bin normalised 10*log10(max(mix,1e-7)); targets uniform in [1, V-2]."""
    g = torch.Generator().manual_seed(seed)
    lens = lengths(B, Tmax)
    clean = torch.zeros(B, Tmax, F_)
    mix = torch.zeros(B, Tmax, F_)
    snr = 10.0 ** (-10.0 / 20.0)
    for b, l in enumerate(lens):
        c = torch.complex(torch.randn(l, F_, generator=g), torch.randn(l, F_, generator=g)) * 300.0
        n = torch.complex(torch.randn(l, F_, generator=g), torch.randn(l, F_, generator=g)) * (300.0 * snr)
        clean[b, :l] = c.abs()
        mix[b, :l] = (c + n).abs()
    mix_log = torch.zeros(B, Tmax, F_)
    valid = torch.cat([mix[b, :l] for b, l in enumerate(lens)], 0)
    lg = 10.0 * torch.log10(torch.clamp(valid, min=1e-7))
    mu, sd = lg.mean(0, keepdim=True), lg.std(0, keepdim=True)
    for b, l in enumerate(lens):
        mix_log[b, :l] = (10.0 * torch.log10(torch.clamp(mix[b, :l], min=1e-7)) - mu) / sd
    targets = torch.randint(1, V - 1, (B * L,), generator=g)
    return clean, mix, mix_log, targets, torch.IntTensor(lens), torch.IntTensor([L] * B)
