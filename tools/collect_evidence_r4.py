#!/usr/bin/env python3
"""Copies the summaries of `bash tools/final_evidence_r4.sh` (gpurun_out/r4_final) into profiles/r04_* and adds the chain kernels' memory-side
traffic per step (FETCH_SIZE x2 + WRITE_SIZE of every lstm_* kernel, from the whole-step counter passes).  Run in the repo after the GPU call."""
import json
import os
import re
import shutil
import subprocess
import sys

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
F, O = os.path.join(R, 'gpurun_out', 'r4_final'), os.path.join(R, 'profiles')
for src, dst in (('bench_default.json', 'r04_bench_default.json'), ('bench_kernel_stats.csv', 'r04_bench_kernel_stats.csv'),
                 ('bench_repeats.txt', 'r04_bench_repeats.txt'), ('bench_nooverlap_kernel_stats.csv', 'r04_bench_nooverlap_kernel_stats.csv'),
                 ('igemm_calls_nooverlap.txt', 'r04_igemm_calls_nooverlap.txt'), ('step_timeline.txt', 'r04_step_timeline.txt'),
                 ('step_bins_2ms.txt', 'r04_step_bins_2ms.txt'), ('bench_kernels.txt', 'r04_kernels_alone.txt'),
                 ('chain_rates_alone.txt', 'r04_chain_rates_alone.txt'), ('roofline_conv_kernel_stats.csv', 'r04_roofline_conv_kernel_stats.csv'),
                 ('step_pmc.json', 'r04_step_pmc.json'), ('pytest_gpu.log', 'r04_pytest_gpu.txt'), ('bench_config2.json', 'r04_bench_config2.json'),
                 ('bench_config3.json', 'r04_bench_config3.json'), ('bench_config5.json', 'r04_bench_config5.json'),
                 ('decoder_loop_alone.txt', 'r04_decoder_loop_alone.txt'), ('decoder_loop_budget.txt', 'r04_decoder_loop_budget.txt')):
    if os.path.exists(os.path.join(F, src)):
        shutil.copyfile(os.path.join(F, src), os.path.join(O, dst))
    else:
        print('missing', src)
if os.path.exists(os.path.join(O, 'r04_bench_nooverlap_kernel_stats.csv')):
    subprocess.run([sys.executable, os.path.join(R, 'tools', 'hbm_table.py'), 'profiles/r04_bench_nooverlap_kernel_stats.csv'], cwd=R,
                   stdout=open(os.path.join(O, 'r04_hbm_kernels.md'), 'w'), check=False)
    p = os.path.join(O, 'r04_hbm_kernels.md')
    t = re.sub(r'`[^`]*bench_nooverlap_kernel_stats.csv`', '`profiles/r04_bench_nooverlap_kernel_stats.csv`', open(p).read())
    open(p, 'w').write(t)
sp = os.path.join(O, 'r04_step_pmc.json')
if os.path.exists(sp):
    d = json.load(open(sp))
    d['command'] = d['command'].replace('final_evidence_r3.sh', 'final_evidence_r4.sh')
    steps = max(1, d['steps_in_run'])
    chain = [r for r in d['kernels_by_sq_busy'] if r['kernel'].startswith('lstm_')]
    tot = sum(r['FETCH_bytes_x2'] + r['WRITE_bytes'] for r in chain)
    d['recurrent_chain_kernels'] = {'kernels': [{k: r[k] for k in ('kernel', 'launches', 'FETCH_bytes_x2', 'WRITE_bytes', 'mfma_over_busy')} for r in chain],
                                    'bytes_per_step': tot / steps, 'GB_per_step': round(tot / steps / 1e9, 2),
                                    'note': 'FETCH_SIZE x2 + WRITE_SIZE of the persistent recurrence kernels, per training step (round 3: ~34 GB)'}
    json.dump(d, open(sp, 'w'), indent=1)
    print('chain kernels: %.2f GB per step; whole step %.1f GB' % (tot / steps / 1e9, d['hbm_bytes_per_step'] / 1e9))
