#!/usr/bin/env python3
"""Per-step latency budget of the persistent recurrences from s_memtime stamps (RE2E_EXPERIMENTS build):

    make -C robust_e2e_gan_amd/csrc EXPERIMENTS=1 -j8
    RE2E_LIB=robust_e2e_gan_amd/libre2e_hip_exp.so python tools/lstm_stamps.py

16 consecutive steps of one sequence are stamped per wavefront (phase boundaries, csrc/lstm.hip LSTM_STAMP); cycles are turned into
microseconds with the clock measured from the same stamps (s_memtime against the 100 MHz s_memrealtime).  Prints mean / max over
workgroups of every phase for wave 0 and for the slowest wave, and the publish -> seen-by-all-peers hop from the chip-wide clock."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

DEV = 'cuda:0'
NW, NS, NP = 16, 16, 12
stamps = torch.zeros(512 * NW * NS * NP, dtype=torch.int64, device=DEV)
os.environ['RE2E_LSTM_STAMPS'] = '%x' % stamps.data_ptr()
from robust_e2e_gan_amd.lib import call, query   # noqa: E402

FWD = ['wait for h(t-1): sweep until every tag matches', 'MFMA h.W_hh (+ next pre-activation loads issued, partial tiles -> LDS)',
       'barrier', 'sum partial tiles + gates + cell + publish h(t)', 'write gates / c / y', 'barrier']
FWD2 = ['wait for h(t-1): poll one piece per producer, then issue the sweep', 'chunks of 16 k consumed in arrival order (MFMA)', 'K-quarter partials -> LDS', 'barrier',
        'sum K quarters + gates + cell + publish h(t)', '(mode 0: issue the next sweep)', 'next pre-activation loads + write gates / c / y']
FWD2_ORDER = [0, 7, 1, 2, 3, 4, 5, 6]
BWD = ['next operands issued + each wave polls the flags of ITS producers', 'load + sum their partial blocks -> LDS', 'barrier', 'dh sum + cell backward + d(gates) -> LDS',
       'barrier', 'MFMA (dG.W_hh)^T with the sc1 stores of the previous tile between the MFMAs', 'drain (s_waitcnt vmcnt(0)), then the wave\'s own flag', '-']


def analyse(name, nwg, waves, phases, nph, order=None):
    st = stamps.view(512, NW, NS, NP)[:nwg, :waves].cpu().double()
    if (st[:, 0, 2:-2, 0] == 0).any():
        print('%s: no stamps (is this the experiments build? RE2E_LIB=%s)' % (name, os.environ.get('RE2E_LIB')))
        return
    st = st[:, :, 2:-2]                                   # steady-state slots
    if order is not None:                                 # stamps in program order
        st = torch.cat([st[..., order], st[..., 8:]], -1) if len(order) == 8 else st
    # clock: memtime ticks per us, from consecutive step tops on wave 0 against the 100 MHz counter
    dt_cyc = (st[:, 0, -1, 0] - st[:, 0, 0, 0])
    dt_us = (st[:, 0, -1, 10] - st[:, 0, 0, 10]) / 100.0
    mhz = float((dt_cyc / dt_us).median())
    step_us = float(dt_us.mean() / (st.shape[2] - 1))
    print('%s: %d workgroups x %d waves, clock %.0f MHz, %.2f us per step (stamped build)' % (name, nwg, waves, mhz, step_us))
    d = (st[..., 1:nph + 1] - st[..., 0:nph]) / mhz      # [wg, wave, slot, phase] us
    # phases that some waves skip (stamp 0) -> mask
    valid = (st[..., 1:nph + 1] > 0) & (st[..., 0:nph] > 0)
    for i, ph in enumerate(phases):
        v = valid[..., i]
        if not v.any():
            continue
        w0 = d[:, 0, :, i][v[:, 0]]
        allw = torch.where(v, d[..., i], torch.zeros_like(d[..., i]))
        print('  %-90s wave0 mean %5.2f  max-wave mean %5.2f  max %5.2f us' % (ph, float(w0.mean()) if w0.numel() else float('nan'),
                                                                             float(allw.max(dim=1).values.mean()), float(allw.max())))
    # hop: last publish of step s (chip-wide clock, stamp 11) -> first / last workgroup past its wait of step s+1 (phase-1 stamp on memtime is local,
    # so use: top of step s+1 (stamp 10) + the wait phase in us)
    pub = st[:, :, :-1, 11]
    pub = torch.where(pub > 0, pub, torch.full_like(pub, float('nan')))
    last_pub = torch.from_numpy(__import__('numpy').nanmax(pub.numpy(), axis=(0, 1)))             # per slot, 10 ns ticks
    first_pub = torch.from_numpy(__import__('numpy').nanmin(pub.numpy(), axis=(0, 1)))
    seen = st[:, 0, 1:, 10] + d[:, 0, 1:, 0] * 100.0                                              # end of the wait phase of step s+1, wave 0
    print('  publish skew over workgroups %.2f us; last publish -> wait satisfied: first workgroup %.2f us, last %.2f us'
          % (float((last_pub - first_pub).mean()) / 100, float((seen.min(0).values - last_pub).mean()) / 100, float((seen.max(0).values - last_pub).mean()) / 100))


def run(T, B, H):
    g = torch.Generator().manual_seed(T * 7 + B + H)
    xg = [(torch.randn(T * B, 4 * H, generator=g) * 0.5).to(DEV) for _ in range(2)]
    whh = [(torch.randn(4 * H, H, generator=g) / H ** 0.5).to(DEV) for _ in range(2)]
    lens = torch.full((B,), T, dtype=torch.int32, device=DEV)
    dy = (torch.randn(T * B, 2 * H, generator=g) * 0.3).to(DEV)
    wsb = query('re2e_lstm_workspace_bytes', B, H)
    ws = torch.empty(wsb // 4 + 16, device=DEV)
    ybuf, cbuf = torch.zeros(T + 2, B, 2 * H, device=DEV), torch.zeros(T + 2, B, 2 * H, device=DEV)
    MT = (B + 31) // 32
    for rep in range(2):
        stamps.zero_()
        call('re2e_lstm_seq_fwd', xg[0].data_ptr(), xg[1].data_ptr(), whh[0].data_ptr(), whh[1].data_ptr(), ybuf.data_ptr(), cbuf.data_ptr(),
             lens.data_ptr(), T, B, H, ws.data_ptr(), wsb)
        torch.cuda.synchronize()
    if os.environ.get('RE2E_LSTM_FWD2', '1') != '0':
        tiles = int(os.environ.get('RE2E_STAMP_TILES', 2 if H == 256 else 4))
        analyse('forward (fwd2, %d units x 16 utterances per workgroup) T=%d B=%d H=%d' % (4 * tiles, T, B, H), (H // (4 * tiles)) * ((B + 15) // 16) * 2, 4, FWD2, 7,
                FWD2_ORDER)
    else:
        analyse('forward  T=%d B=%d H=%d' % (T, B, H), (H // 8) * MT * 2, int(os.environ.get('RE2E_STAMP_FWD_WAVES', 8)), FWD, 6)
    dc = torch.zeros(B, 2 * H, device=DEV)
    un3 = 0 if os.environ.get('RE2E_LSTM_BWD3') == '0' else int(os.environ.get('RE2E_LSTM_BWD3_UN', 16 if H >= 512 else 8))
    uw = int(os.environ.get('RE2E_LSTM_BWD_UW', 2 if H >= 512 else 1))
    for rep in range(2):
        stamps.zero_()
        call('re2e_lstm_seq_bwd', xg[0].data_ptr(), xg[1].data_ptr(), whh[0].data_ptr(), whh[1].data_ptr(), dy.data_ptr(), ybuf.data_ptr(),
             cbuf.data_ptr(), dc.data_ptr(), lens.data_ptr(), T, B, H, None, ws.data_ptr(), wsb)
        torch.cuda.synchronize()
    if un3:
        analyse('backward (bwd3, %d units x 16 utterances per workgroup) T=%d B=%d H=%d' % (un3, T, B, H), (H // un3) * ((B + 15) // 16) * 2, 4, BWD, 8)
    else:
        analyse('backward T=%d B=%d H=%d' % (T, B, H), (H // (8 * uw)) * MT * 2, 4, BWD, 8)
    print('aborts', query('re2e_lstm_abort_count'), flush=True)


if __name__ == '__main__':
    run(400, 32, 256)
    run(200, 64, 512)
