#!/usr/bin/env python3
"""GB/s of the HBM-bound kernels of one config-4 step: rocprofv3 kernel-stats durations (single-stream run,
RE2E_NO_OVERLAP=1, so durations are not inflated by co-running kernels) against ALGORITHMIC bytes -- every tensor a kernel
must read or write once, summed over its launches in one step.

    python3 tools/hbm_table.py profiles/r03_bench_nooverlap_kernel_stats.csv 7 > profiles/r03_hbm_kernels.md

Second argument = number of training steps in the profiled run (every step the process ran: first / warm-up / timed / input-side leg);
omitted: counted from the Adadelta launches (3 per step).  Shapes: B=32, T=800, F=257, NF=80, V=4233,
T'=200; the single-stream run batches both VGG branches (2B = 64 images)."""
import csv
import sys

MB = 1e6
B, T, F, NF, V, T2 = 32, 800, 257, 80, 4233, 200
rows_bt = B * T
px1, px2, px3 = 2 * B * 800 * 80, 2 * B * 400 * 40, 2 * B * 200 * 20          # VGG pixels per stage (2B images)
f4 = 4.0
d2, d3, d4 = B * 200 * 20 * 128 * f4, B * 100 * 10 * 256 * f4, B * 99 * 9 * 512 * f4     # discriminator BatchNorm inputs (bytes)
dbn = d2 + d3 + d4
NPAR = 29147557 + 2831361 + 2764609                                              # ASR + enhancer + D parameters (Appendix B)

# kernel-name substring -> (what, algorithmic bytes per STEP, formula)
KERNELS = [
    ('fbank_fwd_', 'K2 fbank forward (enhanced + clean; since round 3 a matrix-core kernel, 25 us launches: latency-bound, listed for continuity)', 2 * (rows_bt * F + rows_bt * NF) * f4,
     '2 calls x (read (B*T,257) + write (B*T,80))'),
    ('fbank_bwd_', 'K2 fbank backward (matrix-core kernel, 35-40 us launch; reads the saved band power as well)', (2 * rows_bt * F + rows_bt * NF) * f4, 'read x, dy; write dx'),
    ('conv_cin1_fwd_kernel<3, 3>', 'K5 VGG conv1_1 forward (Cin = 1 direct kernel)', (px1 + px1 * 64) * f4, 'read (2B,800,80,1), write (2B,800,80,64)'),
    ('conv_cout1_rows3x3_kernel', 'K5 VGG conv1_1 data gradient (Cout = 1 row-tile kernel, projection on the matrix cores; only the enhanced branch needs it; 0.5 of peak alone, profiles/r03_kernels_alone.txt)', (px1 * 64 + px1) / 2 * f4, 'read dz (B,800,80,64), write dx'),
    ('wgrad_cin1_kernel', 'K5/K9 Cin = 1 weight gradients (VGG conv1_1; D conv1 real + fake)',
     (px1 * 64 + px1) * f4 + 2 * (B * 400 * 40 * 64 + B * 800 * 80) * f4, 'read dout + input, three launches'),
    ('maxpool2_fwd_vec_kernel', 'K5 2x2 max pooling forward (both pools; absent when the pools run in the convolutions\' epilogue)',
     (px1 * 64 + px2 * 64) * f4 + px2 * 64 + (px2 * 128 + px3 * 128) * f4 + px3 * 128, 'read in, write out + 1-byte argmax'),
    ('maxpool2_bwd_vec_kernel', 'K5 2x2 max pooling backward (both pools)', (px1 * 64 + px2 * 64) * f4 + px2 * 64 + (px2 * 128 + px3 * 128) * f4 + px3 * 128,
     'read dy + argmax, write dx'),
    ('colsum_vec_kernel<true>', 'K10 activation backward fused with the bias gradient (D conv1 LeakyReLU; the VGG ReLUs need no pass any more)',
     3 * 2 * B * 400 * 40 * 64 * f4, 'read dy, y; write dz (+ partial column sums); D conv1 x2'),
    ('bn_partial_vec_kernel', 'K9 BatchNorm statistics passes (forward: mean, variance; backward: sum dz, sum dz*xhat)',
     2 * (2 * dbn) + 2 * (2 * dbn) + 0.5 * 2 * dbn * 2, 'approx.: 2 D forwards x 2 passes x read x; 2 D backwards x read x, dy (+ G-step input-gradient pass)'),
    ('bn_apply_vec_kernel', 'K9 BatchNorm + LeakyReLU apply', 2 * 2 * dbn, '2 D forwards x (read x, write y)'),
    ('bn_bwd_apply_vec_kernel', 'K9 BatchNorm + LeakyReLU backward apply', 3 * 3 * dbn, '3 D backwards x (read dy, x; write dx)'),
    ('ctc_lse_gather', 'K6 CTC log-sum-exp over V + label gather', T2 * B * V * f4, 'read logits (T\',B,V)'),
    ('ctc_grad', 'K6 CTC gradient (softmax - occupancy)', 2 * T2 * B * V * f4, 'read logits, write gradient'),
    ('adadelta_kernel', 'K11 fused Adadelta (3 networks)', NPAR * 7 * f4, 'read p, g, E[g^2], E[dx^2]; write p, E[g^2], E[dx^2]'),
    ('sumsq_partial_kernel', 'K10 global gradient norm (3 networks)', NPAR * f4, 'read g'),
    ('loss_partial_kernel', 'K10 mean losses (enhancement MSE over (B,T,80) x2 inputs; LSGAN MSE to a constant)',
     2 * rows_bt * NF * f4 + 4 * B * 98 * 8 * f4, 'read both operands'),
    ('vgg_pack_tile_kernel', 'K5 VGG output cut + re-pad + transpose to time-major (fwd + bwd)', 2 * 2 * px3 * 128 * f4, 'read + write (2B,200,20,128)'),
    ('act_bwd_kernel', 'K10 activation backward without a bias (G-step through the frozen D conv1)', 3 * B * 400 * 40 * 64 * f4, 'read dy, y; write dz'),
]


def main(path, steps):
    stats = {}
    for r in csv.DictReader(open(path)):
        stats[r['Name']] = (int(r['Calls']), float(r['TotalDurationNs']))
    if not steps:                                        # three Adadelta launches (one per network) per training step
        steps = sum(v[0] for n, v in stats.items() if 'adadelta_kernel' in n) // 3
    import os, re
    m = re.match(r'(r\d+)_', os.path.basename(path))
    print('# HBM-bound kernels of one config-4 training step: achieved GB/s against 8 TB/s (%s)\n' % (m.group(1) if m else '?'))
    print('Source: `%s` (rocprofv3 --kernel-trace --stats of `RE2E_NO_OVERLAP=1 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline`: ONE stream, so a kernel\'s '
          'duration is its own; %d steps in the run).  Bytes are ALGORITHMIC (each tensor the kernel has to touch, once), summed over the '
          'kernel\'s launches in one step; formulas in `tools/hbm_table.py`.  Peak 8.0 TB/s (MI355X_MICROARCH.md; 6.3 TB/s is what a float4 copy '
          'reaches).  Rows marked approx. mix several shapes whose exact launch mix was not separated.\n' % (path, steps))
    print('| kernel | what | launches / step | ms / step | algorithmic MB / step | GB/s | of 8 TB/s |')
    print('|---|---|---|---|---|---|---|')
    for sub, what, nbytes, formula in KERNELS:
        hit = [(n, v) for n, v in stats.items() if sub in n]
        if not hit:
            continue
        calls = sum(v[0] for _, v in hit) / steps
        ms = sum(v[1] for _, v in hit) / steps / 1e6
        gbs = nbytes / (ms * 1e-3) / 1e9
        print('| `%s` | %s (%s) | %.1f | %.3f | %.0f | %.0f | %.2f |' % (sub, what, formula, calls, ms, nbytes / MB, gbs, gbs / 8000.0))
    print('\nRows below 0.40: the two fbank launches and `loss_partial_kernel` are 7-40 us launches (latency, not bandwidth); the conv1_1 data gradient streams, in the '
          'single-stream step, 0.5 GB that the previous kernel has just written (alone it reaches 4.1 TB/s).')
    print('\nNot in the table: the attention-step kernels (K7: 41 launches of 5-23 us each, latency-bound, 21 MB of encoder states re-read from '
          'L2 / Infinity Cache per step -- `attloc_*` rows of the CSV) and the persistent recurrences (K4: dependent-step latency, `profiles/'
          'r01_recurrence_chain_rates.txt`).')


if __name__ == '__main__':
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 0)
