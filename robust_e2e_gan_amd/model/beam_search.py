"""Beam-search decoding: Decoder.recognize_beam (model/e2e_decoder.py:171-369, no LM) with
CTCPrefixScore (model/e2e_ctc.py:78-155) and end_detect (model/e2e_common.py:226-252).

The reference advances one hypothesis at a time (B = 1 attention / LSTMCell / output-layer calls, ``beam`` of them
per output position).  Here all live hypotheses of a position form ONE batch on the GPU: a single AttLoc step,
LSTMCell, output layer and row-wise log-softmax for the whole beam, one device->host copy of the (beam, V) local
scores per position; the search bookkeeping (pruning, length penalties, end detection) stays on the host and follows the
reference line by line, so the n-best lists agree.

Joint CTC/attention decoding: the CTC prefix scores of ALL live hypotheses x their ``ctc_beam`` candidate labels are one
launch per position (re2e_ctc_prefix_score: top-k pre-selection, Algorithm 2's recursion with the forward variables in
LDS / registers, the combined local score); the hypotheses' CTC states never leave the GPU and only
3 x nh x ctc_beam numbers (candidate labels, local scores, prefix scores) are copied to the host per position -- instead of
the (nh, V) attention scores plus a numpy recursion per hypothesis.  ``ctc_beam`` > 64 (ctc_weight == 1.0 scores all V
labels upstream, e2e_decoder.py:233-234) takes its candidate list from a device-side stable sort of the attention scores
(re2e_ctc_prefix_score_cands, one thread per candidate).  The host scorer below -- upstream's own numpy algorithm -- is kept as the
tests' arbiter (HOST_CTC_SCORER)."""
import numpy as np
import torch

from .. import ops
from ..lib import call
from .e2e_common import host_to_dev, lens_dev

CTC_SCORING_RATIO = 1.5          # e2e_decoder.py:20
LOGZERO = -10000000000.0


class CTCPrefixScore(object):
    """Log prefix probabilities of ``y + [c]`` for the candidate labels ``cs`` given the frame posteriors ``x`` (T, V)
    (Watanabe et al., "Hybrid CTC/attention architecture ...", Algorithm 2, evaluated for all candidates at once)."""

    def __init__(self, x, blank, eos):
        self.x, self.blank, self.eos, self.T = x, blank, eos, len(x)

    def initial_state(self):
        r = np.full((self.T, 2), LOGZERO, dtype=np.float32)
        r[:, 1] = np.cumsum(self.x[:, self.blank], dtype=np.float32)
        return r

    def __call__(self, y, cs, r_prev):
        n = len(y) - 1                                   # output length without <sos>
        xs = self.x[:, cs]
        r = np.empty((self.T, 2, len(cs)), dtype=np.float32)
        if n == 0:
            r[0, 0], r[0, 1] = xs[0], LOGZERO
        else:
            r[n - 1] = LOGZERO
        r_sum = np.logaddexp(r_prev[:, 0], r_prev[:, 1])
        log_phi = np.repeat(r_sum[:, None], len(cs), 1)
        if n > 0:
            log_phi[:, cs == y[-1]] = r_prev[:, 1:2]     # a repeated label needs a blank in between
        start = max(n, 1)
        log_psi = r[start - 1, 0].copy()
        xb = self.x[:, self.blank]
        for t in range(start, self.T):
            r[t, 0] = np.logaddexp(r[t - 1, 0], log_phi[t - 1]) + xs[t]
            r[t, 1] = np.logaddexp(r[t - 1, 0], r[t - 1, 1]) + xb[t]
            log_psi = np.logaddexp(log_psi, log_phi[t - 1] + xs[t])
        log_psi[cs == self.eos] = r_sum[-1]
        return log_psi, np.moveaxis(r, 2, 0)


def end_detect(ended, i, M=3, D_end=np.log(1 * np.exp(-10))):
    if not ended:
        return False
    best = max(h['score'] for h in ended)
    count = 0
    for m in range(M):
        same = [h['score'] for h in ended if len(h['yseq']) == i - m]
        if same and max(same) - best < D_end:
            count += 1
    return count == M


def _topk(row, k):
    idx = np.argsort(-row, kind='stable')[:k]
    return row[idx], idx


DEVICE_CTC_MAX_BEAM = 64        # re2e_ctc_prefix_score: one thread per candidate label, <= 64 candidates per hypothesis (in-kernel top-k);
#                                 more candidates (ctc_weight == 1.0: all V labels): re2e_ctc_prefix_score_cands on a device-sorted list
HOST_CTC_SCORER = False         # tests: upstream's numpy CTCPrefixScore on the host instead (the arbiter of the device scorers)


def recognize_beam(p, h, lpz, recog_args, eos, prefix='', lpz_dev=None):
    """``p``: reference-named decoder / attention Parameters; ``h``: (T, eprojs) encoder states of ONE utterance on the
    GPU; ``lpz``: (T, V) CTC log posteriors (numpy) or None.  Returns the n-best list of {'yseq', 'score'}."""
    dev = h.device
    T, E = h.shape
    beam, penalty, ctc_weight = recog_args.beam_size, recog_args.penalty, recog_args.ctc_weight
    embed, w_ih, w_hh = p[prefix + 'dec.embed.weight'], p[prefix + 'dec.decoder.0.weight_ih'], p[prefix + 'dec.decoder.0.weight_hh']
    b_ih, b_hh = p[prefix + 'dec.decoder.0.bias_ih'], p[prefix + 'dec.decoder.0.bias_hh']
    out_w, out_b = p[prefix + 'dec.output.weight'], p[prefix + 'dec.output.bias']
    mlp_dec, mlp_att = p[prefix + 'att.mlp_dec.weight'], p[prefix + 'att.mlp_att.weight']
    loc_conv, gvec_w, gvec_b = p[prefix + 'att.loc_conv.weight'], p[prefix + 'att.gvec.weight'], p[prefix + 'att.gvec.bias']
    V, Dd, D, A = out_w.shape[0], embed.shape[1], w_hh.shape[1], mlp_dec.shape[0]
    C, Fh = loc_conv.shape[0], (loc_conv.shape[3] - 1) // 2
    ldw = Dd + E
    with torch.no_grad():
        pre1 = ops.linear(h.unsqueeze(0), p[prefix + 'att.mlp_enc.weight'], p[prefix + 'att.mlp_enc.bias'])   # (1,T,A)
        h_rep = h.unsqueeze(0).expand(beam, T, E).contiguous()          # every hypothesis attends over the same utterance
        pre_rep = pre1.expand(beam, T, A).contiguous()
        hl = lens_dev([T] * beam, dev)
        w_decT = torch.empty(D, A, device=dev)
        call('re2e_transpose01', mlp_dec.data_ptr(), w_decT.data_ptr(), A, D, 1)
        w_ctx = w_ih.data_ptr() + 4 * Dd
        maxlen = T if recog_args.maxlenratio == 0 else max(1, int(recog_args.maxlenratio * T))
        minlen = int(recog_args.minlenratio * T)
        hyps = [{'score': np.float32(0.0), 'yseq': [eos], 'parent': 0}]
        dev_ctc = False
        if lpz is not None:
            ctc = CTCPrefixScore(lpz, 0, eos)
            hyps[0]['ctc_state'], hyps[0]['ctc_score'] = ctc.initial_state(), np.float32(0.0)
            ctc_beam = min(V, int(beam * CTC_SCORING_RATIO)) if ctc_weight != 1.0 else V
            dev_ctc = not HOST_CTC_SCORER
            if dev_ctc:
                lpz_d = lpz_dev.float().contiguous() if lpz_dev is not None else torch.from_numpy(np.ascontiguousarray(lpz, np.float32)).to(dev)
                r_prev = torch.from_numpy(hyps[0].pop('ctc_state')).to(dev).view(1, T, 2)      # states stay on the device from here on
        z, c, a_prev = torch.zeros(1, D, device=dev), torch.zeros(1, D, device=dev), None
        ended = []
        for i in range(maxlen):
            nh = len(hyps)
            ids = host_to_dev(np.asarray([hp['yseq'][i] for hp in hyps], np.int32), dev)
            emb = torch.empty(nh, Dd, device=dev)
            call('re2e_embedding_fwd', embed.data_ptr(), ids.data_ptr(), nh, Dd, emb.data_ptr(), Dd)
            w_new, cx = torch.empty(nh, T, device=dev), torch.empty(nh, E, device=dev)
            conv, dpj, e_scr = torch.empty(nh, T, C, device=dev), torch.empty(nh, A, device=dev), torch.empty(nh, T, device=dev)
            call('re2e_attloc_fwd', pre_rep.data_ptr(), h_rep.data_ptr(), z.data_ptr(), a_prev.data_ptr() if a_prev is not None else None,
                 hl.data_ptr(), w_decT.data_ptr(), mlp_att.data_ptr(), loc_conv.data_ptr(), gvec_w.data_ptr(), gvec_b.data_ptr(), nh, T, E, D, A,
                 C, Fh, w_new.data_ptr(), cx.data_ptr(), E, conv.data_ptr(), dpj.data_ptr(), e_scr.data_ptr())
            gates = torch.empty(nh, 4 * D, device=dev)
            ops.gemm(emb, w_ih, gates, nh, 4 * D, Dd, transb=True, ldb=ldw, bias=b_ih, bias2=b_hh)
            ops.gemm(cx, w_ctx, gates, nh, 4 * D, E, transb=True, ldb=ldw, beta=1.0, dev=dev)
            ops.gemm(z, w_hh, gates, nh, 4 * D, D, transb=True, beta=1.0)
            z_new, c_new = torch.empty(nh, D, device=dev), torch.empty(nh, D, device=dev)
            call('re2e_lstm_cell_fwd', gates.data_ptr(), c.data_ptr(), c_new.data_ptr(), z_new.data_ptr(), nh, D)
            logits, lsm = torch.empty(nh, V, device=dev), torch.empty(nh, V, device=dev)
            ops.gemm(z_new, out_w, logits, nh, V, D, transb=True, bias=out_b)
            call('re2e_log_softmax_rows', logits.data_ptr(), nh, V, V, lsm.data_ptr())
            if dev_ctc:
                last = host_to_dev(np.asarray([hp['yseq'][-1] for hp in hyps], np.int32), dev)
                if max(len(hp['yseq']) - 1 for hp in hyps) > T:
                    # CTCPrefixScore indexes r[output_length - 1] over the T frames (e2e_ctc.py:128-133): a hypothesis longer than
                    # the utterance has frames (maxlenratio > 1) is an IndexError upstream
                    raise IndexError('CTC prefix score: hypothesis of %d labels on %d encoder frames (lower maxlenratio)'
                                     % (max(len(hp['yseq']) - 1 for hp in hyps), T))
                olen = host_to_dev(np.asarray([len(hp['yseq']) - 1 for hp in hyps], np.int32), dev)
                prev = host_to_dev(np.asarray([hp['ctc_score'] for hp in hyps], np.float32), dev, torch.float32)
                out_d = torch.empty(2, nh, ctc_beam, device=dev)                  # [0] local scores, [1] prefix scores
                r_new = torch.empty(nh * ctc_beam, 2 * T, device=dev)
                if ctc_beam <= DEVICE_CTC_MAX_BEAM:
                    cand_d = torch.empty(nh, ctc_beam, dtype=torch.int32, device=dev)
                    call('re2e_ctc_prefix_score', lpz_d.data_ptr(), T, V, lsm.data_ptr(), nh, r_prev.data_ptr(), last.data_ptr(), olen.data_ptr(),
                         prev.data_ptr(), ctc_beam, float(np.float32(1.0 - ctc_weight)), float(np.float32(ctc_weight)), 0, eos, cand_d.data_ptr(),
                         out_d[0].data_ptr(), out_d[1].data_ptr(), r_new.data_ptr())
                else:
                    # the candidates in the order torch.topk / _topk give them: attention score descending, ties -> lower label (stable sort;
                    # NaN scores last, as the kernel's own selection ranks them)
                    order = torch.sort(torch.nan_to_num(lsm, nan=float('-inf')), dim=1, descending=True, stable=True)[1]
                    cand_d = order[:, :ctc_beam].to(torch.int32).contiguous()
                    call('re2e_ctc_prefix_score_cands', lpz_d.data_ptr(), T, V, lsm.data_ptr(), nh, r_prev.data_ptr(), last.data_ptr(), olen.data_ptr(),
                         prev.data_ptr(), cand_d.data_ptr(), ctc_beam, float(np.float32(1.0 - ctc_weight)), float(np.float32(ctc_weight)), 0, eos,
                         out_d[0].data_ptr(), out_d[1].data_ptr(), r_new.data_ptr())
                cand_all, out_all = cand_d.cpu().numpy(), out_d.cpu().numpy()      # 3 x nh x ctc_beam numbers: this position's host round trip
            else:
                local_all = lsm.cpu().numpy()                             # host scorer: the (nh, V) local scores cross once per position
            kept = []
            for k, hyp in enumerate(hyps):
                if dev_ctc:
                    best_scores, joint = _topk(out_all[0, k], beam)
                    for j in range(len(joint)):
                        kept.append({'score': np.float32(hyp['score'] + best_scores[j]), 'yseq': hyp['yseq'] + [int(cand_all[k, joint[j]])],
                                     'parent': k, 'ctc_row': k * ctc_beam + int(joint[j]), 'ctc_score': out_all[1, k, joint[j]]})
                    kept = sorted(kept, key=lambda x: x['score'], reverse=True)[:beam]
                    continue
                local_att = local_all[k]
                if lpz is not None:
                    _, cand = _topk(local_att, ctc_beam)
                    ctc_scores, ctc_states = ctc(hyp['yseq'], cand, hyp['ctc_state'])
                    local = (np.float32(1.0 - ctc_weight) * local_att[cand] + np.float32(ctc_weight) * (ctc_scores - hyp['ctc_score'])).astype(np.float32)
                    best_scores, joint = _topk(local, beam)
                    best_ids = cand[joint]
                else:
                    best_scores, best_ids = _topk(local_att, beam)
                    joint = None
                for j in range(len(best_ids)):
                    new = {'score': np.float32(hyp['score'] + best_scores[j]), 'yseq': hyp['yseq'] + [int(best_ids[j])], 'parent': k}
                    if lpz is not None:
                        new['ctc_state'], new['ctc_score'] = ctc_states[joint[j]], ctc_scores[joint[j]]
                    kept.append(new)
                kept = sorted(kept, key=lambda x: x['score'], reverse=True)[:beam]
            hyps = kept
            if i == maxlen - 1:
                for hyp in hyps:
                    hyp['yseq'].append(eos)
            remained = []
            for hyp in hyps:
                if hyp['yseq'][-1] == eos:
                    if len(hyp['yseq']) > minlen:
                        hyp['score'] = np.float32(hyp['score'] + (i + 1) * penalty)
                        ended.append(hyp)
                else:
                    remained.append(hyp)
            if end_detect(ended, i) and recog_args.maxlenratio == 0.0:
                break
            hyps = remained
            if not hyps:
                break
            parents = host_to_dev(np.asarray([hp['parent'] for hp in hyps], np.int64), dev, torch.int64)
            z, c, a_prev = z_new.index_select(0, parents), c_new.index_select(0, parents), w_new.index_select(0, parents)
            if dev_ctc:                                                   # the survivors' CTC states, gathered on the device
                rows = host_to_dev(np.asarray([hp['ctc_row'] for hp in hyps], np.int64), dev, torch.int64)
                r_prev = r_new.index_select(0, rows)
        best = sorted(ended, key=lambda x: x['score'], reverse=True)[:min(len(ended), recog_args.nbest)]
        return [{'yseq': b['yseq'], 'score': float(b['score'])} for b in best]
