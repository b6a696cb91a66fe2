#!/usr/bin/env python3
"""What would F(4x4,3x3) cost in accuracy?  (round-5 review, item 1b: price it before building it.)

The VGG 3x3 layers run as Winograd F(2x2,3x3) in fp32 (csrc/winograd.hip: transforms that only add and halve).  F(4x4,3x3) executes 36 instead of
64 multiply-adds per 4x4 outputs (1.78x fewer matrix instructions) but its transforms multiply by 4, 5, 8, 1/6, 1/24: this script emulates BOTH forms
with every intermediate rounded to fp32 (transforms, the channel contraction in fp32 as the matrix cores accumulate it, output transform) on CPU
tensors shaped like the three layers (post-ReLU inputs, LeCun-normal weights, 64 / 128 channels), forward and data gradient (a convolution with the
flipped kernel: the same arithmetic), and reports the largest error against the float64 direct convolution relative to the largest output --
the scale the parity tests use (tests/: 1e-3 on outputs, 1e-3 .. 1.5e-3 on gradients)."""
import torch
import torch.nn.functional as F

torch.manual_seed(0)


def mats(kind):
    if kind == 2:
        BT = [[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]]
        G = [[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]]
        AT = [[1, 1, 1, 0], [0, 1, -1, -1]]
    else:
        BT = [[4, 0, -5, 0, 1, 0], [0, -4, -4, 1, 1, 0], [0, 4, -4, -1, 1, 0], [0, -2, -1, 2, 1, 0], [0, 2, -1, -2, 1, 0], [0, 4, 0, -5, 0, 1]]
        G = [[1 / 4, 0, 0], [-1 / 6, -1 / 6, -1 / 6], [-1 / 6, 1 / 6, -1 / 6], [1 / 24, 1 / 12, 1 / 6], [1 / 24, -1 / 12, 1 / 6], [0, 0, 1]]
        AT = [[1, 1, 1, 1, 1, 0], [0, 1, -1, 2, -2, 0], [0, 1, 1, 4, 4, 0], [0, 1, -1, 8, -8, 1]]
    t = lambda m: torch.tensor(m, dtype=torch.float32)
    return t(BT), t(G), t(AT)


def wino(x, w, kind):
    """x (N, C, H, W), w (K, C, 3, 3), pad 1, fp32 throughout; H, W multiples of the output tile"""
    m = 2 if kind == 2 else 4
    a = m + 2
    BT, G, AT = mats(kind)
    N, C, H, W = x.shape
    K = w.shape[0]
    xp = F.pad(x, (1, 1, 1, 1))
    d = xp.unfold(2, a, m).unfold(3, a, m)                      # (N, C, th, tw, a, a)
    V = torch.einsum('ia,nctuab,jb->nctuij', BT, d, BT)          # input transform
    U = torch.einsum('ia,kcab,jb->kcij', G, w, G)                # weight transform
    M = torch.einsum('nctuij,kcij->nktuij', V, U)                # the products, contracted over channels in fp32
    Y = torch.einsum('pi,nktuij,qj->nktupq', AT, M, AT)          # output transform
    th, tw = Y.shape[2], Y.shape[3]
    return Y.permute(0, 1, 2, 4, 3, 5).reshape(N, K, th * m, tw * m)


def report(name, C, K, H, W, N=2):
    x = F.relu(torch.randn(N, C, H, W)) * 1.3                    # post-ReLU activations
    w = torch.randn(K, C, 3, 3) / (9 * C) ** 0.5
    ref = F.conv2d(x.double(), w.double(), padding=1)
    sc = ref.abs().max().item()
    e_dir = (F.conv2d(x, w, padding=1).double() - ref).abs().max().item() / sc
    e2 = (wino(x, w, 2).double() - ref).abs().max().item() / sc
    e4 = (wino(x, w, 4).double() - ref).abs().max().item() / sc
    # data gradient: dy (N, K, H, W) random-signed, kernel flipped and transposed
    dy = torch.randn(N, K, H, W) * 0.1
    wf = w.flip(2, 3).transpose(0, 1).contiguous()
    rg = F.conv2d(dy.double(), wf.double(), padding=1)
    sg = rg.abs().max().item()
    g2 = (wino(dy, wf, 2).double() - rg).abs().max().item() / sg
    g4 = (wino(dy, wf, 4).double() - rg).abs().max().item() / sg
    print('%-22s forward: direct fp32 %.1e  F(2x2) %.1e  F(4x4) %.1e   | data gradient: F(2x2) %.1e  F(4x4) %.1e   (max error / max |output|)'
          % (name, e_dir, e2, e4, g2, g4))


if __name__ == '__main__':
    report('conv1_2  64 ->  64', 64, 64, 96, 80)
    report('conv2_1  64 -> 128', 64, 128, 48, 40)
    report('conv2_2 128 -> 128', 128, 128, 48, 40)
