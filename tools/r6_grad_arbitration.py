#!/usr/bin/env python3
"""Who is right where the bench's first GPU step and the fp32 oracle disagree on a gradient tensor?  The bench's own set-up (config 4, the seed-1234
batch, CMVN estimated from the enhanced features), one GPU step, the oracle's step in float32 and in FLOAT64 (oracle.joint.JointState(dtype=float64):
the arbiter of tests/test_modules_gpu.py::test_joint_step_gradients_vs_fp64_reference, here at full size: minutes of host time).  Prints, for the
tensors with the largest GPU-vs-fp32-oracle difference, each side's distance from the float64 result (relative to the tensor's largest entry)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from oracle import joint as oj
from robust_e2e_gan_amd.data.synthetic import make_batch
from robust_e2e_gan_amd.joint_train import JointTrainer, config4_opt
from robust_e2e_gan_amd.model.feat_model import mel_matrix

dev = torch.device('cuda:0')
opt = config4_opt()
enh, fb, asr, gan = bench.build(opt, dev)
B, T, L = bench.CONFIG_SHAPES[4]
batch = make_batch(B, T, L, opt.odim, seed=1234)
cmvn = bench.synthetic_cmvn(enh, fb, [make_batch(B, T, L, opt.odim, seed=77 + i) for i in range(2)], dev)
sd0 = [{k: v.detach().cpu().clone() for k, v in m.state_dict().items()} for m in (enh, asr, gan)]
step, tr, _ = bench.make_stepper(4, opt, (enh, fb, asr, gan), batch, cmvn.to(dev), dev)
step()
torch.cuda.synchronize()
gpu = {pre: {k: p.grad.detach().cpu().clone() for k, p in m.named_parameters() if p.grad is not None} for pre, m in (('enh', enh), ('asr', asr), ('gan', gan))}
torch.set_num_threads(bench.host_cores())
clean, mix, mix_log, targets, il, tl = batch
W = torch.from_numpy(mel_matrix())
r32 = oj.joint_step(bench._oracle_state(opt, sd0, W), (clean, mix, mix_log, targets, il.tolist(), tl.tolist()), cmvn, update=False)
print('fp32 oracle done', flush=True)
st64 = bench._oracle_state(opt, sd0, W)
st64 = oj.JointState(sd0[0], sd0[1], sd0[2], W, st64.cfg, dtype=torch.float64)
r64 = oj.joint_step(st64, (clean.double(), mix.double(), mix_log.double(), targets, il.tolist(), tl.tolist()), cmvn.double(), update=False)
print('fp64 oracle done', flush=True)
rows = []
for pre, key in (('enh', 'g_enh'), ('asr', 'g_asr'), ('gan', 'g_gan')):
    for k, g in gpu[pre].items():
        if k not in r64[key]:
            continue
        t64 = r64[key][k]
        sc = float(t64.abs().max()) + 1e-30
        rows.append((float((g.double() - r32[key][k].double()).abs().max()) / sc, float((g.double() - t64).abs().max()) / sc,
                     float((r32[key][k].double() - t64).abs().max()) / sc, pre + '.' + k))
rows.sort(reverse=True)
print('%-44s %12s %12s %12s' % ('tensor', 'gpu vs fp32', 'gpu vs fp64', 'fp32 vs fp64'))
for a, b, c, n in rows[:12]:
    print('%-44s %12.2e %12.2e %12.2e' % (n, a, b, c))
print('largest over all %d tensors: gpu vs fp64 %.2e, fp32 oracle vs fp64 %.2e' % (len(rows), max(r[1] for r in rows), max(r[2] for r in rows)))
