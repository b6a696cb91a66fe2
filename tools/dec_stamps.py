#!/usr/bin/env python3
"""Per-token latency budget of the persistent decoder loop (csrc/decloop.hip) from clock stamps (RE2E_EXPERIMENTS build):

    make -C robust_e2e_gan_amd/csrc EXPERIMENTS=1 -j8
    RE2E_EXPERIMENTS=1 RE2E_LIB=robust_e2e_gan_amd/libre2e_hip_exp.so python tools/dec_stamps.py

Thread 0 of every workgroup stamps the 100 MHz chip-wide clock at its phase boundaries for 16 consecutive tokens; printed: the mean
time of every phase per role and the hops between the roles (last publish -> first / last consumer past its wait)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

DEV = 'cuda:0'
stamps = torch.zeros(256 * 16 * 16, dtype=torch.int64, device=DEV)
os.environ['RE2E_DEC_STAMPS'] = '%x' % stamps.data_ptr()
from robust_e2e_gan_amd import ops   # noqa: E402

GATE = ['wait z_i (all gate workgroups)', 'load z_i + MFMA W_hh', 'wait cx_i (all attention workgroups)', 'load pieces + MFMA W_ctx + partials -> LDS + barrier',
        'cell + stores issued', 'drain + barrier (then the flag)']
ATT = ['w_{i-1} -> LDS, location conv (MFMA), barriers', 'u = pre + W_att conv', 'wait z_i', 'load z_i[b], dp slice (3 barriers)', 'partial energies (64 tanh per frame)',
       'drain + barrier (then the flag)', 'wait the partial energies of utterance b', 'sum partials + softmax over T', 'context piece -> LDS + barrier',
       'sum 16 groups + store + drain (then the flag)']


def main(B=32, T=200, L1=41, E=512, A=320, D=300, C=10, Fh=100):
    g = torch.Generator().manual_seed(1)
    r = lambda *s, scale=1.0: (torch.randn(*s, generator=g) * scale).to(DEV)
    hmask, pre = r(B, T, E), r(B, T, A)
    Pm = dict(embed=r(50, D, scale=0.5), w_ih=r(4 * D, D + E, scale=0.08), w_hh=r(4 * D, D, scale=0.08), b_ih=r(4 * D, scale=0.1), b_hh=r(4 * D, scale=0.1),
              mlp_dec=r(A, D, scale=0.1), mlp_att=r(A, C, scale=0.5), loc_conv=r(C, 1, 1, 2 * Fh + 1, scale=0.3), gvec_w=r(1, A, scale=0.3), gvec_b=r(1, scale=0.1))
    ids = torch.randint(0, 50, (L1, B), generator=g).to(torch.int32).to(DEV)
    hlens = torch.full((B,), T, dtype=torch.int32, device=DEV)
    for rep in range(3):
        stamps.zero_()
        with torch.no_grad():
            ops.DecoderLoopFn.apply(hmask, pre, ids, hlens, L1, Pm)
        torch.cuda.synchronize()
    st = stamps.view(256, 16, 16).cpu().double() / 100.0          # us
    NG = (D + 7) // 8
    nwg = int((st[:, 0, 0] > 0).sum())
    if nwg == 0:
        print('no stamps (experiments build? RE2E_LIB=%s)' % os.environ.get('RE2E_LIB'))
        return
    gate, att = st[:NG, 2:-2], st[NG:nwg, 2:-2]
    step = float((gate[:, -1, 0] - gate[:, 0, 0]).mean() / (gate.shape[1] - 1))
    print('B=%d T=%d: %d gate + %d attention workgroups, %.2f us per token' % (B, T, NG, nwg - NG, step))
    for name, x, names, last in (('gate', gate, GATE, 6), ('attention', att, ATT, 10)):
        print(' %s workgroup:' % name)
        for ph in range(last):
            d = x[:, :, ph + 1] - x[:, :, ph]
            print('   %-72s mean %5.2f  max %5.2f us' % (names[ph], float(d.mean()), float(d.max())))
        d = x[:, 1:, 0] - x[:, :-1, last]
        print('   %-72s mean %5.2f' % ('(flag store, loop back to the top)', float(d.mean())))
    # hops (chip-wide clock): z published (gate stamp 6) -> attention past its z wait (stamp 3) of the NEXT token, etc.
    zpub_last = gate[:, :-1, 6].max(0).values
    print(' z hop: last gate publish -> attention workgroups past the wait: first %.2f, last %.2f us; gate workgroups: first %.2f, last %.2f'
          % (float((att[:, 1:, 3].min(0).values - zpub_last).mean()), float((att[:, 1:, 3].max(0).values - zpub_last).mean()),
             float((gate[:, 1:, 1].min(0).values - zpub_last).mean()), float((gate[:, 1:, 1].max(0).values - zpub_last).mean())))
    epub_last = att[:, :, 6].max(0).values
    print(' e hop: last attention publish (chip) -> past the wait: first %.2f, last %.2f us' % (float((att[:, :, 7].min(0).values - epub_last).mean()),
                                                                                             float((att[:, :, 7].max(0).values - epub_last).mean())))
    cpub_last = att[:, :, 10].max(0).values
    print(' cx hop: last attention publish -> gate workgroups past the wait: first %.2f, last %.2f us' % (float((gate[:, :, 3].min(0).values - cpub_last).mean()),
                                                                                                        float((gate[:, :, 3].max(0).values - cpub_last).mean())))
    print(' skew of the attention workgroups at the top of a token: %.2f us' % float((att[:, :, 0].max(0).values - att[:, :, 0].min(0).values).mean()))


UNIT = ['wait d dec_proj_{i+1} (all attention workgroups)', 'load it + MFMA mlp_dec + partials -> LDS + barrier', 'cell backward, d(gates) stores, drain, barrier (then the flag)',
        'wait d(gates)_i (all unit workgroups)', 'gather d(gates)_i + MFMA W_hh']
COL = ['(top)', '-', '-', 'wait d(gates)_i (all unit workgroups)', 'gather d(gates)_i + MFMA W_ctx', 'cross-wave sum, d cx stores, drain, barrier (then the flag)']
ATTB = ['recompute u = pre + W_att conv_i (MFMA), tanh, dtg', 'wait the chunk scalars of the last token', 'wait d cx_i (all column workgroups)', 'load d cx_i[b], dc . cx (block sum)',
        'g = dc . enc (LDS rows) + barrier, de, barrier', 'du, d dec_proj partial (row sums, stores), d conv^T (MFMA) -> LDS, barrier',
        'd conv rows summed + stored, drain, barrier (then the flags)', 'wait the partials of utterance b', 'reduce own columns of d dec_proj, store, drain, barrier (then the flag)',
        'wait the d conv rows of utterance b', 'window -> LDS, transposed location conv, d w, scalar, flag']


def backward(B=32, T=200, L1=41, E=512, A=320, D=300, C=10, Fh=100):
    g = torch.Generator().manual_seed(1)
    r = lambda *s, scale=1.0: (torch.randn(*s, generator=g) * scale).to(DEV)
    hmask, pre = r(B, T, E).requires_grad_(True), r(B, T, A).requires_grad_(True)
    Pm = dict(embed=r(50, D, scale=0.5), w_ih=r(4 * D, D + E, scale=0.08), w_hh=r(4 * D, D, scale=0.08), b_ih=r(4 * D, scale=0.1), b_hh=r(4 * D, scale=0.1),
              mlp_dec=r(A, D, scale=0.1), mlp_att=r(A, C, scale=0.5), loc_conv=r(C, 1, 1, 2 * Fh + 1, scale=0.3), gvec_w=r(1, A, scale=0.3), gvec_b=r(1, scale=0.1))
    Pm = {k: torch.nn.Parameter(v) for k, v in Pm.items()}
    ids = torch.randint(0, 50, (L1, B), generator=g).to(torch.int32).to(DEV)
    hlens = torch.full((B,), T, dtype=torch.int32, device=DEV)
    for rep in range(3):
        z, w = ops.DecoderLoopFn.apply(hmask, pre, ids, hlens, L1, Pm)
        torch.cuda.synchronize()
        stamps.zero_()                                   # (the forward stamped too: keep the backward's only)
        z.sum().backward()
        torch.cuda.synchronize()
    st = stamps.view(256, 16, 16).cpu().double() / 100.0
    NU, NC = (D + 15) // 16, E // 16
    nwg = int((st[:, 4, 0] > 0).sum())
    if nwg == 0:
        print('no stamps')
        return
    unit, col, att = st[:NU, 2:-2], st[NU:NU + NC, 2:-2], st[NU + NC:nwg, 2:-2]
    step = float((att[:, -1, 0] - att[:, 0, 0]).mean() / (att.shape[1] - 1))
    print('BACKWARD B=%d T=%d: %d unit + %d column + %d attention workgroups, %.2f us per token' % (B, T, NU, NC, nwg - NU - NC, step))
    for name, x, names, seq in (('unit', unit, UNIT, [0, 1, 2, 3, 4, 5]), ('column', col, COL, [0, 4, 5, 6]), ('attention', att, ATTB, list(range(12)))):
        print(' %s workgroup:' % name)
        for a_, b_ in zip(seq[:-1], seq[1:]):
            d = x[:, :, b_] - x[:, :, a_]
            nm = names[a_] if name != 'column' else {0: COL[3], 4: COL[4], 5: COL[5]}[a_]
            print('   %-92s mean %5.2f  max %5.2f us' % (nm, float(d.mean()), float(d.max())))
        d = x[:, 1:, 0] - x[:, :-1, seq[-1]]
        print('   %-92s mean %5.2f' % ('(flag store, loop back to the top)', float(d.mean())))


if __name__ == '__main__':
    if 'bwd' in sys.argv:
        backward()
        sys.exit(0)
    main()
    if len(sys.argv) > 1:
        main(B=8, T=750, L1=40)
