"""K11: fused multi-tensor optimizers on ONE flat buffer per network.

Replaces torch.optim.Adadelta / Adam + clip_grad_norm_ + the NaN guard of joint_train.py:131-140,
188-193.  All parameters of a network are re-pointed into one flat fp32 buffer (each tensor 256-B
aligned so the GEMM loaders keep their 16-B vector path), all gradients into a second one; the
update is a single HIP launch, the global grad-norm a two-stage deterministic reduction, and the
clip coefficient / NaN flag stay on the device (no host sync).  The same flat gradient buffer is
what dist.py hands to RCCL."""
import torch

from .lib import call, query, workspace

_ALIGN = 64   # floats


class FlatOptimizer:
    def __init__(self, params, kind='adadelta', rho=0.95, eps=1e-8, lr=None, betas=(0.9, 0.999)):
        self.params = [p for p in params if p.requires_grad]
        assert self.params, 'no trainable parameters'
        dev = self.params[0].device
        self.kind = kind
        self.offsets, n = [], 0
        for p in self.params:
            self.offsets.append(n)
            n += (p.numel() + _ALIGN - 1) // _ALIGN * _ALIGN
        self.numel = n
        self.flat = torch.zeros(n, dtype=torch.float32, device=dev)
        self.grad = torch.zeros(n, dtype=torch.float32, device=dev)
        for p, off in zip(self.params, self.offsets):
            view = self.flat[off:off + p.numel()].view(p.shape)
            view.copy_(p.data)
            p.data = view
        self._attach_grads()
        self.s1 = torch.zeros(n, dtype=torch.float32, device=dev)    # Adadelta square_avg / Adam exp_avg
        self.s2 = torch.zeros(n, dtype=torch.float32, device=dev)    # Adadelta acc_delta  / Adam exp_avg_sq
        self.stats = torch.tensor([0.0, 1.0, 1.0, 0.0, 1.0, 1.0], dtype=torch.float32, device=dev)
        self.sumsq = torch.zeros(1, dtype=torch.float32, device=dev)
        self.step_count = 0
        if kind == 'adadelta':
            self.param_groups = [dict(params=self.params, rho=rho, eps=eps, lr=1.0 if lr is None else lr)]
        elif kind == 'adam':
            self.param_groups = [dict(params=self.params, betas=betas, eps=eps, lr=1e-3 if lr is None else lr)]
        else:
            raise ValueError(kind)

    def _attach_grads(self):
        for p, off in zip(self.params, self.offsets):
            g = self.grad[off:off + p.numel()].view(p.shape)
            if p.grad is None or p.grad.data_ptr() != g.data_ptr():
                p.grad = g

    def zero_grad(self):
        self.grad.zero_()
        self._attach_grads()

    def clip_grad_norm(self, max_norm):
        """Global L2 norm of the flat gradient -> device stats [norm, coef, finite | norm, 1, finite]; returns
        the norm as a 0-dim device tensor (reading it is the only host sync, and is optional)."""
        wsb = query('re2e_reduce_workspace_bytes', self.numel)
        ws = workspace(wsb, self.grad.device, 'reduce')
        call('re2e_sumsq', self.grad.data_ptr(), self.numel, self.sumsq.data_ptr(), ws.data_ptr(), wsb)
        call('re2e_clip_coef', self.sumsq.data_ptr(), float(max_norm), self.stats.data_ptr())
        return self.stats[0]

    def grad_sumsq(self):
        """Sum of squares of the flat gradient as a 1-element device tensor (no clipping, no host sync): the trainer's step gate reads it to
        refuse an update whose OTHER optimizer's gradients are not finite (csrc/lstm.hip re2e_step_gate)."""
        wsb = query('re2e_reduce_workspace_bytes', self.numel)
        ws = workspace(wsb, self.grad.device, 'reduce')
        call('re2e_sumsq', self.grad.data_ptr(), self.numel, self.sumsq.data_ptr(), ws.data_ptr(), wsb)
        return self.sumsq

    def gate_stats(self):
        """View usable by ANOTHER optimizer: same NaN guard, clip coefficient 1 (joint_train.py:188-193)."""
        return self.stats[3:6]

    def step(self, stats=None):
        st = self.stats if stats is None else stats
        g = self.param_groups[0]
        self.step_count += 1
        if self.kind == 'adadelta':
            call('re2e_adadelta_step', self.flat.data_ptr(), self.grad.data_ptr(), self.s1.data_ptr(), self.s2.data_ptr(), self.numel,
                 float(g['rho']), float(g['eps']), float(g['lr']), st.data_ptr())
        else:
            call('re2e_adam_step', self.flat.data_ptr(), self.grad.data_ptr(), self.s1.data_ptr(), self.s2.data_ptr(), self.numel,
                 float(g['lr']), float(g['betas'][0]), float(g['betas'][1]), float(g['eps']), self.step_count, st.data_ptr())
