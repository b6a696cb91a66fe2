#!/usr/bin/env python3
"""Copies the summaries of `bash tools/final_evidence_r6.sh` (gpurun_out/r6_final) into profiles/r06_* and adds the chain kernels' memory-side
traffic per step (FETCH_SIZE x2 + WRITE_SIZE of every lstm_* kernel, from the whole-step counter passes).  Run in the repo after the GPU call."""
import json
import os
import re
import shutil
import subprocess
import sys

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
F, O = os.path.join(R, 'gpurun_out', 'r6_final'), os.path.join(R, 'profiles')
for src, dst in (('bench_default.json', 'r06_bench_default.json'), ('bench_kernel_stats.csv', 'r06_bench_kernel_stats.csv'),
                 ('bench_repeats.txt', 'r06_bench_repeats.txt'), ('bench_nooverlap_kernel_stats.csv', 'r06_bench_nooverlap_kernel_stats.csv'),
                 ('igemm_calls_nooverlap.txt', 'r06_igemm_calls_nooverlap.txt'), ('step_timeline.txt', 'r06_step_timeline.txt'),
                 ('step_bins_2ms.txt', 'r06_step_bins_2ms.txt'), ('bench_kernels.txt', 'r06_kernels_alone.txt'),
                 ('chain_rates_alone.txt', 'r06_chain_rates_alone.txt'), ('roofline_conv_kernel_stats.csv', 'r06_roofline_conv_kernel_stats.csv'),
                 ('step_pmc.json', 'r06_step_pmc.json'), ('pytest_gpu.log', 'r06_pytest_gpu.txt'), ('bench_config2.json', 'r06_bench_config2.json'),
                 ('bench_config3.json', 'r06_bench_config3.json'), ('bench_config5.json', 'r06_bench_config5.json'),
                 ('decoder_loop_alone.txt', 'r06_decoder_loop_alone.txt'), ('decoder_loop_budget.txt', 'r06_decoder_loop_budget.txt'),
                 ('gemm_nt_variants.txt', 'r06_gemm_nt_variants_final.txt'), ('kernel_clock.txt', 'r06_kernel_clock.txt'),
                 ('wino_final.txt', 'r06_wino_final.txt')):
    if os.path.exists(os.path.join(F, src)):
        shutil.copyfile(os.path.join(F, src), os.path.join(O, dst))
    else:
        print('missing', src)
if os.path.exists(os.path.join(O, 'r06_bench_nooverlap_kernel_stats.csv')):
    subprocess.run([sys.executable, os.path.join(R, 'tools', 'hbm_table.py'), 'profiles/r06_bench_nooverlap_kernel_stats.csv'], cwd=R,
                   stdout=open(os.path.join(O, 'r06_hbm_kernels.md'), 'w'), check=False)
    p = os.path.join(O, 'r06_hbm_kernels.md')
    t = re.sub(r'`[^`]*bench_nooverlap_kernel_stats.csv`', '`profiles/r06_bench_nooverlap_kernel_stats.csv`', open(p).read())
    open(p, 'w').write(t)
sp = os.path.join(O, 'r06_step_pmc.json')
if os.path.exists(sp):
    d = json.load(open(sp))
    d['command'] = d['command'].replace('final_evidence_r3.sh', 'final_evidence_r6.sh')
    steps = max(1, d['steps_in_run'])
    chain = [r for r in d['kernels_by_sq_busy'] if r['kernel'].startswith('lstm_')]
    tot = sum(r['FETCH_bytes_x2'] + r['WRITE_bytes'] for r in chain)
    d['recurrent_chain_kernels'] = {'kernels': [{k: r[k] for k in ('kernel', 'launches', 'FETCH_bytes_x2', 'WRITE_bytes', 'mfma_over_busy')} for r in chain],
                                    'bytes_per_step': tot / steps, 'GB_per_step': round(tot / steps / 1e9, 2),
                                    'note': 'FETCH_SIZE x2 + WRITE_SIZE of the persistent recurrence kernels, per training step (round 5: 27.9 GB)'}
    json.dump(d, open(sp, 'w'), indent=1)
    print('chain kernels: %.2f GB per step; whole step %.1f GB' % (tot / steps / 1e9, d['hbm_bytes_per_step'] / 1e9))

# ---- the roofline kernel's counters of THIS round (final_evidence_r6.sh section 4): same derivations as round 3
cp = os.path.join(F, 'conv_pmc.json')
if os.path.exists(cp):
    s = json.load(open(cp))
    N, H, W, C, K = 64, 800, 80, 64, 64
    inp = N * H * W * C * 4
    fl = 2.0 * 9 * C * K * N * H * W

    def one(d):
        (name, cs), = d.items() if len(d) == 1 else [max(d.items(), key=lambda kv: len(kv[1]))]
        return name, cs

    def stats_of(sub):
        for r in s.get('stats', []):
            if sub in r['Name']:
                return {'calls': int(r['Calls']), 'avg_ms': float(r['AverageNs']) / 1e6, 'min_ms': float(r['MinNs']) / 1e6, 'max_ms': float(r['MaxNs']) / 1e6}
        return None
    wn, wc = one(s['wino_conv3x3_kernel'])
    fe, wr = wc['FETCH_SIZE'], wc['WRITE_SIZE']
    fetch_b, write_b = fe['avg'] * 1024 * 2, wr['avg'] * 1024
    pooled = N * (H // 2) * (W // 2) * K
    alg = inp + 16 * K * C * 4 + pooled * 5
    KERNEL = ('wino_conv3x3_kernel<8> at the VGG conv1_2 shape (64x800x80, 64->64, 3x3) with the bias + ReLU + 2x2 max pool epilogue, '
              'tools/roofline_conv.py (= bench.py conv_roofline: 3 + 20 launches)')
    json.dump({'kernel': KERNEL,
               'command': 'bash tools/final_evidence_r6.sh, section 4 (rocprofv3 --kernel-trace --pmc FETCH_SIZE and --pmc WRITE_SIZE in two separate passes)',
               'FETCH_SIZE': {'launches': fe['launches'], 'avg_KiB_as_reported': fe['avg'], 'min_KiB': fe['min'], 'max_KiB': fe['max'],
                              'gfx950_correction': 'x2 (profiles/r02_fetch_size_calibration.json)', 'avg_bytes_corrected': fetch_b},
               'WRITE_SIZE': {'launches': wr['launches'], 'avg_KiB': wr['avg'], 'avg_bytes': write_b},
               'algorithmic_bytes': {'input': inp, 'transformed_weights': 16 * K * C * 4, 'pooled_output_plus_index_bytes': pooled * 5, 'total': alg},
               'traffic_bytes_per_launch': fetch_b + write_b, 'reads_over_input': fetch_b / inp, 'traffic_over_algorithmic': (fetch_b + write_b) / alg},
              open(os.path.join(O, 'r06_conv1_2_wino_pmc_traffic.json'), 'w'), indent=1)
    cc = {k: v['avg'] for k, v in wc.items()}
    st = stats_of('wino_conv3x3_kernel')
    json.dump({'kernel': KERNEL, 'command': 'bash tools/final_evidence_r6.sh, section 4 (two SQ passes of 6 counters; averages over the launches, summed over the chip)',
               'counters': cc, 'kernel_stats_same_script': dict(st, direct_equivalent_TFLOPs=fl / st['avg_ms'] / 1e9, executed_TFLOPs=fl / 2.25 / st['avg_ms'] / 1e9,
                                                               executed_frac_of_157_3=fl / 2.25 / st['avg_ms'] / 1e9 / 157.3) if st else None,
               'derived': {'mfma_pipe_utilisation': cc['SQ_VALU_MFMA_BUSY_CYCLES'] / (cc['SQ_BUSY_CYCLES'] * 32),
                           # SQ_BUSY_CYCLES is summed over 32 shader engines: / 32 = shader cycles the kernel took; over its duration = the clock it ran at
                           # (the 157.3 TFLOP/s peak assumes 2.4 GHz)
                           'shader_clock_ghz': (cc['SQ_BUSY_CYCLES'] / 32) / (st['avg_ms'] * 1e6) if st else None,
                           'vector_L1_accesses_per_cycle_and_CU': (cc['TCP_TOTAL_CACHE_ACCESSES_sum'] / 256) / (cc['SQ_BUSY_CYCLES'] / 32) if 'TCP_TOTAL_CACHE_ACCESSES_sum' in cc else None,
                           'vector_L1_accesses_per_load_instruction': cc['TCP_TOTAL_CACHE_ACCESSES_sum'] / cc['SQ_INSTS_VMEM_RD'] if 'TCP_TOTAL_CACHE_ACCESSES_sum' in cc and 'SQ_INSTS_VMEM_RD' in cc else None,
                           'other_vector_instructions_per_mfma': (cc['SQ_INSTS_VALU'] - cc['SQ_VALU_MFMA_BUSY_CYCLES'] / 64) / (cc['SQ_VALU_MFMA_BUSY_CYCLES'] / 64),
                           'wave_cycles_split': {k: cc['SQ_' + k] / cc['SQ_WAVE_CYCLES'] for k in ('WAIT_INST_ANY', 'WAIT_ANY', 'ACTIVE_INST_ANY')},
                           'lds_bank_conflict_share_of_lds_active_cycles': cc['SQ_LDS_BANK_CONFLICT'] / cc['SQ_LDS_IDX_ACTIVE']}},
              open(os.path.join(O, 'r06_conv1_2_wino_pmc_sq.json'), 'w'), indent=1)
    print('conv1_2 Winograd: %.3f GB per launch, %.3f x algorithmic' % ((fetch_b + write_b) / 1e9, (fetch_b + write_b) / alg))
# ---- memory-side traffic of the dense engine forms per step (FETCH_SIZE x2 + WRITE_SIZE), next to round 5
if os.path.exists(sp):
    d = json.load(open(sp))
    steps = max(1, d['steps_in_run'])
    fam = {}
    for r in d['kernels_by_sq_busy']:
        k = r['kernel']
        name = ('gemm_nt2 (x W^T, convolutions, K-sliced products: csrc/gemm_nt.hip)' if k.startswith('gemm_nt2_kernel') else
                'igemm DenseM x DenseM (dy^T x weight gradients)' if k.startswith('igemm_kernel<DenseM, DenseM') else
                'igemm DenseK x DenseK' if k.startswith('igemm_kernel<DenseK, DenseK') else
                'igemm other forms' if k.startswith('igemm_kernel') else None)
        if name:
            fam[name] = fam.get(name, 0.0) + (r['FETCH_bytes_x2'] + r['WRITE_bytes']) / steps / 1e9
    d['dense_engine_traffic_GB_per_step'] = {k: round(v, 2) for k, v in fam.items()}
    d['dense_engine_traffic_note'] = 'round 5: gemm_nt2 21.0, igemm other forms 6.6, DenseM x DenseM 4.5, DenseK x DenseK 0.4 GB per step'
    json.dump(d, open(sp, 'w'), indent=1)
    print('dense engine traffic per step:', d['dense_engine_traffic_GB_per_step'])
