"""Location-aware attention (mirror of AttLoc, model/e2e_attention.py:199-299).

The reference evaluates AttLoc.forward once per decoder step as ~10 small ATen ops; here the whole
step is one fused HIP kernel (csrc/attloc.hip) driven from the decoder loop (ops.DecoderLoopFn).
The other eleven attention classes of the reference are never selected by the recipe
(--atype location, options/base_options.py:48) and are out of scope."""
import torch

from .e2e_common import ConvParams, LinearParams


class AttLoc(torch.nn.Module):
    def __init__(self, eprojs, dunits, att_dim, aconv_chans, aconv_filts, aact_fuc='softmax'):
        super(AttLoc, self).__init__()
        if aact_fuc != 'softmax':
            raise NotImplementedError('only the softmax attention activation is on the hot path')
        self.mlp_enc = LinearParams(eprojs, att_dim)
        self.mlp_dec = LinearParams(dunits, att_dim, bias=False)
        self.mlp_att = LinearParams(aconv_chans, att_dim, bias=False)
        self.loc_conv = ConvParams(1, aconv_chans, 1, 2 * aconv_filts + 1, bias=False, padding=(0, aconv_filts))
        self.gvec = LinearParams(att_dim, 1)
        self.dunits, self.eprojs, self.att_dim = dunits, eprojs, att_dim
        self.aconv_chans, self.aconv_filts = aconv_chans, aconv_filts

    def reset(self):
        """The reference caches pre_compute_enc_h between steps; the fused loop owns that state."""
        return None
