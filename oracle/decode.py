"""Beam-search decoding on CPU (TEST INFRASTRUCTURE, see oracle/__init__.py).

Restates /root/reference model/e2e_model.py:204-236 (E2E.recognize), model/e2e_decoder.py:171-369
(Decoder.recognize_beam, no LM) , model/e2e_ctc.py:78-155 (CTCPrefixScore) and model/e2e_common.py:226-252 (end_detect)
on the functional nets of oracle/nets.py.  Pinned by tests/golden/recog_tiny.npz (n-best lists of the reference)."""
import numpy as np
import torch
import torch.nn.functional as F

from . import nets

CTC_SCORING_RATIO = 1.5          # e2e_decoder.py:20
LOGZERO = -10000000000.0


class CTCPrefixScore(object):
    """e2e_ctc.py:78-155: log prefix probabilities of ``y + [c]`` for a set of next labels ``cs`` (Watanabe et al., Alg. 2)."""

    def __init__(self, x, blank, eos):
        self.x, self.blank, self.eos, self.T = x, blank, eos, len(x)

    def initial_state(self):
        r = np.full((self.T, 2), LOGZERO, dtype=np.float32)
        r[0, 1] = self.x[0, self.blank]
        for i in range(1, self.T):
            r[i, 1] = r[i - 1, 1] + self.x[i, self.blank]
        return r

    def __call__(self, y, cs, r_prev):
        out_len = len(y) - 1
        r = np.ndarray((self.T, 2, len(cs)), dtype=np.float32)
        xs = self.x[:, cs]
        if out_len == 0:
            r[0, 0] = xs[0]
            r[0, 1] = LOGZERO
        else:
            r[out_len - 1] = LOGZERO
        r_sum = np.logaddexp(r_prev[:, 0], r_prev[:, 1])
        last = y[-1]
        if out_len > 0 and last in cs:
            log_phi = np.ndarray((self.T, len(cs)), dtype=np.float32)
            for i in range(len(cs)):
                log_phi[:, i] = r_sum if cs[i] != last else r_prev[:, 1]
        else:
            log_phi = r_sum
        start = max(out_len, 1)
        log_psi = r[start - 1, 0]
        for t in range(start, self.T):
            phi = log_phi[t - 1]
            r[t, 0] = np.logaddexp(r[t - 1, 0], phi) + xs[t]
            r[t, 1] = np.logaddexp(r[t - 1, 0], r[t - 1, 1]) + self.x[t, self.blank]
            log_psi = np.logaddexp(log_psi, phi + xs[t])
        eos_pos = np.where(cs == self.eos)[0]
        if len(eos_pos) > 0:
            log_psi[eos_pos] = r_sum[-1]
        return log_psi, np.rollaxis(r, 2)


def end_detect(ended, i, M=3, D_end=np.log(1 * np.exp(-10))):
    """e2e_common.py:226-252"""
    if not ended:
        return False
    count = 0
    best = sorted(ended, key=lambda h: h['score'], reverse=True)[0]
    for m in range(M):
        same = [h for h in ended if len(h['yseq']) == i - m]
        if same:
            b = sorted(same, key=lambda h: h['score'], reverse=True)[0]
            if b['score'] - best['score'] < D_end:
                count += 1
    return count == M


def _att_step(p, h, pre, z, a_prev):
    """One AttLoc step for a single hypothesis (B = 1, the whole utterance is valid)."""
    T = h.size(0)
    if a_prev is None:
        a_prev = h.new_full((1, T), 1.0 / T)
    K = p['att.loc_conv.weight'].size(3)
    conv = F.conv2d(a_prev.view(1, 1, 1, T), p['att.loc_conv.weight'], padding=(0, (K - 1) // 2)).squeeze(2).transpose(1, 2)
    conv = F.linear(conv, p['att.mlp_att.weight'])
    dec = F.linear(z, p['att.mlp_dec.weight']).view(1, 1, -1)
    e = F.linear(torch.tanh(conv + pre + dec), p['att.gvec.weight'], p['att.gvec.bias']).squeeze(2)
    w = F.softmax(2.0 * e, dim=1)
    return (h.unsqueeze(0) * w.unsqueeze(2)).sum(1), w


def recognize(p, x, elayers, beam_size, penalty, ctc_weight, maxlenratio, minlenratio, nbest):
    """x: (1, T, F) features of ONE utterance.  Returns the n-best list [{'yseq': [...], 'score': float}]."""
    with torch.no_grad():
        hpad, _ = nets.encoder_forward(p, x, [x.shape[1]], elayers)
        h = hpad[0]
        lpz = F.log_softmax(F.linear(h, p['ctc.ctc_lo.weight'], p['ctc.ctc_lo.bias']), dim=1).numpy() if ctc_weight > 0.0 else None
        V = p['dec.output.weight'].size(0)
        eos = sos = V - 1
        D = p['dec.decoder.0.weight_hh'].size(1)
        pre = F.linear(h.unsqueeze(0), p['att.mlp_enc.weight'], p['att.mlp_enc.bias'])
        maxlen = h.shape[0] if maxlenratio == 0 else max(1, int(maxlenratio * h.size(0)))
        minlen = int(minlenratio * h.size(0))
        hyp = {'score': 0.0, 'yseq': [sos], 'c': h.new_zeros(1, D), 'z': h.new_zeros(1, D), 'a': None}
        if lpz is not None:
            ctc = CTCPrefixScore(lpz, 0, eos)
            hyp['ctc_state'], hyp['ctc_score'] = ctc.initial_state(), 0.0
            ctc_beam = min(lpz.shape[-1], int(beam_size * CTC_SCORING_RATIO)) if ctc_weight != 1.0 else lpz.shape[-1]
        hyps, ended = [hyp], []
        for i in range(maxlen):
            kept = []
            for hyp in hyps:
                ey = F.embedding(torch.tensor([hyp['yseq'][i]]), p['dec.embed.weight'])
                att_c, att_w = _att_step(p, h, pre, hyp['z'], hyp['a'])
                gates = F.linear(torch.cat([ey, att_c], 1), p['dec.decoder.0.weight_ih'], p['dec.decoder.0.bias_ih']) + \
                    F.linear(hyp['z'], p['dec.decoder.0.weight_hh'], p['dec.decoder.0.bias_hh'])
                gi, gf, gg, go = gates.chunk(4, 1)
                c = torch.sigmoid(gf) * hyp['c'] + torch.sigmoid(gi) * torch.tanh(gg)
                z = torch.sigmoid(go) * torch.tanh(c)
                local_att = F.log_softmax(F.linear(z, p['dec.output.weight'], p['dec.output.bias']), dim=1)
                if lpz is not None:
                    _, ids = torch.topk(local_att, ctc_beam, dim=1)
                    ctc_scores, ctc_states = ctc(hyp['yseq'], ids[0].numpy(), hyp['ctc_state'])
                    local = (1.0 - ctc_weight) * local_att[:, ids[0]] + ctc_weight * torch.from_numpy(ctc_scores - hyp['ctc_score'])
                    best_scores, joint = torch.topk(local, beam_size, dim=1)
                    best_ids = ids[:, joint[0]]
                else:
                    best_scores, best_ids = torch.topk(local_att, beam_size, dim=1)
                for j in range(beam_size):
                    new = {'z': z, 'c': c, 'a': att_w, 'score': hyp['score'] + best_scores[0, j], 'yseq': hyp['yseq'] + [int(best_ids[0, j])]}
                    if lpz is not None:
                        new['ctc_state'], new['ctc_score'] = ctc_states[joint[0, j]], ctc_scores[joint[0, j]]
                    kept.append(new)
                kept = sorted(kept, key=lambda h_: h_['score'], reverse=True)[:beam_size]
            hyps = kept
            if i == maxlen - 1:
                for hyp in hyps:
                    hyp['yseq'].append(eos)
            remained = []
            for hyp in hyps:
                if hyp['yseq'][-1] == eos:
                    if len(hyp['yseq']) > minlen:
                        hyp['score'] += (i + 1) * penalty
                        ended.append(hyp)
                else:
                    remained.append(hyp)
            if end_detect(ended, i) and maxlenratio == 0.0:
                break
            hyps = remained
            if not hyps:
                break
        best = sorted(ended, key=lambda h_: h_['score'], reverse=True)[:min(len(ended), nbest)]
        return [{'yseq': b['yseq'], 'score': float(b['score'])} for b in best]
