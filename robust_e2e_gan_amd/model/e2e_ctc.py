"""CTC head (mirror of CTC.__init__/forward, model/e2e_ctc.py:17-66)."""
import numpy as np
import torch

from .. import ops
from ..lib import Re2eError, call as lib_call
from .e2e_common import LinearParams, host_to_dev, lens_dev


class CTC(torch.nn.Module):
    def __init__(self, odim, eprojs, dropout_rate):
        super(CTC, self).__init__()
        self.dropout_rate = dropout_rate
        self.loss = None
        self.ctc_lo = LinearParams(eprojs, odim)
        self.ignore_id = -1

    def forward_tm(self, hs_tm, hlens, ys):
        """hs_tm: time-major (T',B,eprojs) encoder states (unmasked, Appendix A.8)."""
        dev = hs_tm.device
        ylist = [[int(v) for v in y.tolist() if int(v) != self.ignore_id] for y in ys]
        ll = [len(y) for y in ylist]
        flat = host_to_dev(np.asarray(sum(ylist, []), np.int32), dev)
        off = host_to_dev(np.concatenate([[0], np.cumsum(ll)[:-1]]).astype(np.int32), dev)
        # e2e_ctc.py:51: F.dropout(hs_pad, p) with its DEFAULT training=True -- applied in eval mode too (Appendix A.8)
        hs_tm = ops.dropout(hs_tm, self.dropout_rate)
        logits = ops.linear(hs_tm, self.ctc_lo.weight, self.ctc_lo.bias)           # (T',B,V): warp-ctc's layout
        self.loss = ops.ctc_loss(logits, lens_dev(hlens, dev), flat, off, host_to_dev(np.asarray(ll, np.int32), dev), max(ll))
        return self.loss

    def forward(self, hs_pad, hlens, ys_pad):
        return self.forward_tm(ops.transpose01(hs_pad), hlens, ys_pad)

    def log_softmax(self, hs_pad):
        """e2e_ctc.py:68-75: log_softmax(ctc_lo(hs_pad), dim=2) -- frame posteriors for joint CTC/attention decoding."""
        with torch.no_grad():
            logits = ops.linear(hs_pad, self.ctc_lo.weight, self.ctc_lo.bias)
            V = logits.shape[-1]
            out = torch.empty_like(logits)
            lib_call('re2e_log_softmax_rows', logits.data_ptr(), logits.numel() // V, V, V, out.data_ptr())
            return out
