#!/usr/bin/env python3
"""Copies the summaries of `bash tools/final_evidence_r3.sh` (gpurun_out/r3_final) into profiles/r03_* and derives the PMC evidence
files of the roofline kernel from the counter summary.  Run in the repo after the GPU call."""
import json
import os
import shutil

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
F, O = os.path.join(R, 'gpurun_out', 'r3_final'), os.path.join(R, 'profiles')
for src, dst in (('bench_default.json', 'r03_bench_default.json'), ('bench_kernel_stats.csv', 'r03_bench_kernel_stats.csv'),
                 ('bench_repeats.txt', 'r03_bench_repeats.txt'), ('bench_nooverlap_kernel_stats.csv', 'r03_bench_nooverlap_kernel_stats.csv'),
                 ('igemm_calls_nooverlap.txt', 'r03_igemm_calls_nooverlap.txt'), ('hbm_kernels.md', 'r03_hbm_kernels.md'),
                 ('step_timeline.txt', 'r03_step_timeline.txt'), ('step_bins_2ms.txt', 'r03_step_bins_2ms.txt'),
                 ('bench_kernels.txt', 'r03_kernels_alone.txt'), ('roofline_conv_kernel_stats.csv', 'r03_roofline_conv_kernel_stats.csv'),
                 ('step_pmc.json', 'r03_step_pmc.json'), ('pytest_gpu.log', 'r03_pytest_gpu.txt')):
    if os.path.exists(os.path.join(F, src)):
        shutil.copyfile(os.path.join(F, src), os.path.join(O, dst))
    else:
        print('missing', src)
# the HBM table is regenerated here from the copied CSV (same numbers, repository-relative source path)
import subprocess, sys
subprocess.run([sys.executable, os.path.join(R, 'tools', 'hbm_table.py'), 'profiles/r03_bench_nooverlap_kernel_stats.csv'], cwd=R,
               stdout=open(os.path.join(O, 'r03_hbm_kernels.md'), 'w'), check=True)
p = os.path.join(O, 'r03_hbm_kernels.md')
if os.path.exists(p):
    t = open(p).read()
    import re
    t = re.sub(r'`[^`]*bench_nooverlap_kernel_stats.csv`', '`profiles/r03_bench_nooverlap_kernel_stats.csv`', t)
    open(p, 'w').write(t)
s = json.load(open(os.path.join(F, 'conv_pmc.json')))
N, H, W, C, K = 64, 800, 80, 64, 64
inp = N * H * W * C * 4


def one(d):
    (name, cs), = d.items() if len(d) == 1 else [max(d.items(), key=lambda kv: len(kv[1]))]
    return name, cs


def stats_of(sub):
    for r in s.get('stats', []):
        if sub in r['Name']:
            return {'calls': int(r['Calls']), 'avg_ms': float(r['AverageNs']) / 1e6, 'min_ms': float(r['MinNs']) / 1e6, 'max_ms': float(r['MaxNs']) / 1e6}
    return None


fl = 2.0 * 9 * C * K * N * H * W
# --- the Winograd kernel, as the step launches it (bias + ReLU + 2x2 max pool epilogue)
wn, wc = one(s['wino_conv3x3_kernel'])
fe, wr = wc['FETCH_SIZE'], wc['WRITE_SIZE']
fetch_b, write_b = fe['avg'] * 1024 * 2, wr['avg'] * 1024
pooled = N * (H // 2) * (W // 2) * K
alg = inp + 16 * K * C * 4 + pooled * 5
KERNEL = ('wino_conv3x3_kernel<8> at the VGG conv1_2 shape (64x800x80, 64->64, 3x3) with the bias + ReLU + 2x2 max pool epilogue, '
          'tools/roofline_conv.py (= bench.py conv_roofline: 3 + 20 launches)')
json.dump({'kernel': KERNEL,
           'command': 'bash tools/final_evidence_r3.sh, section 4 (rocprofv3 --kernel-trace --pmc FETCH_SIZE and --pmc WRITE_SIZE in two separate passes)',
           'FETCH_SIZE': {'launches': fe['launches'], 'avg_KiB_as_reported': fe['avg'], 'min_KiB': fe['min'], 'max_KiB': fe['max'],
                          'gfx950_correction': 'x2 (profiles/r02_fetch_size_calibration.json)', 'avg_bytes_corrected': fetch_b},
           'WRITE_SIZE': {'launches': wr['launches'], 'avg_KiB': wr['avg'], 'avg_bytes': write_b},
           'algorithmic_bytes': {'input': inp, 'transformed_weights': 16 * K * C * 4, 'pooled_output_plus_index_bytes': pooled * 5, 'total': alg},
           'traffic_bytes_per_launch': fetch_b + write_b, 'reads_over_input': fetch_b / inp, 'traffic_over_algorithmic': (fetch_b + write_b) / alg,
           'notes': 'WRITE_SIZE equals the pooled tensor + its index bytes exactly; reads are within a few % of the input tensor: the 2-pixel '
                    'halo of a 16x8 patch is served by the XCD\'s L2 (workgroups dealt XCD-aware).'},
          open(os.path.join(O, 'r03_conv1_2_wino_pmc_traffic.json'), 'w'), indent=1)
c = {k: v['avg'] for k, v in wc.items()}
st = stats_of('wino_conv3x3_kernel')
json.dump({'kernel': KERNEL, 'command': 'bash tools/final_evidence_r3.sh, section 4 (two SQ passes of 6 counters; averages over the launches, summed over the chip)',
           'counters': c, 'kernel_stats_same_script': dict(st, direct_equivalent_TFLOPs=fl / st['avg_ms'] / 1e9, executed_TFLOPs=fl / 2.25 / st['avg_ms'] / 1e9,
                                                           executed_frac_of_157_3=fl / 2.25 / st['avg_ms'] / 1e9 / 157.3) if st else None,
           'derived': {'mfma_instructions_per_launch': c['SQ_VALU_MFMA_BUSY_CYCLES'] / 64,
                       'algorithmic_mfma_per_launch_direct_form': fl / 4096, 'executed_over_direct': c['SQ_VALU_MFMA_BUSY_CYCLES'] / 64 / (fl / 4096),
                       'mfma_pipe_utilisation': c['SQ_VALU_MFMA_BUSY_CYCLES'] / (c['SQ_BUSY_CYCLES'] * 32),
                       'mfma_pipe_utilisation_note': 'SQ_BUSY_CYCLES is summed over the 32 shader engines (8 XCD x 4), SQ_VALU_MFMA_BUSY_CYCLES over the 1024 SIMDs: '
                                                     'busy / (sq_busy / 32 * 1024) = busy / (32 * sq_busy)',
                       'other_vector_instructions_per_mfma': (c['SQ_INSTS_VALU'] - c['SQ_VALU_MFMA_BUSY_CYCLES'] / 64) / (c['SQ_VALU_MFMA_BUSY_CYCLES'] / 64),
                       'wave_cycles_split': {k: c['SQ_' + k] / c['SQ_WAVE_CYCLES'] for k in ('WAIT_INST_ANY', 'WAIT_ANY', 'ACTIVE_INST_ANY')},
                       'lds_bank_conflict_share_of_lds_active_cycles': c['SQ_LDS_BANK_CONFLICT'] / c['SQ_LDS_IDX_ACTIVE']}},
          open(os.path.join(O, 'r03_conv1_2_wino_pmc_sq.json'), 'w'), indent=1)
# --- the direct halo-patch kernel of the same product (round 2's roofline kernel; plain + fused-pool launches of the same script)
hn, hc = one(s['conv3x3_halo_kernel'])
fe, wr = hc['FETCH_SIZE'], hc['WRITE_SIZE']
json.dump({'kernel': 'conv3x3_halo_kernel<16,16,1,true>, same shape, same script: 23 plain (full-resolution output) + 23 fused-pool launches, averaged together',
           'FETCH_SIZE': {'launches': fe['launches'], 'avg_KiB_as_reported': fe['avg'], 'avg_bytes_corrected': fe['avg'] * 2048},
           'WRITE_SIZE': {'launches': wr['launches'], 'avg_KiB': wr['avg'], 'avg_bytes': wr['avg'] * 1024},
           'traffic_bytes_per_launch': fe['avg'] * 2048 + wr['avg'] * 1024, 'reads_over_input': fe['avg'] * 2048 / inp,
           'mfma_pipe_utilisation': hc['SQ_VALU_MFMA_BUSY_CYCLES']['avg'] / (hc['SQ_BUSY_CYCLES']['avg'] * 32), 'kernel_stats_same_script': stats_of('conv3x3_halo_kernel')},
          open(os.path.join(O, 'r03_conv1_2_pmc_traffic.json'), 'w'), indent=1)
# --- whole-step counters: add the utilisation with the same normalisation
p = os.path.join(O, 'r03_step_pmc.json')
if os.path.exists(p):
    d = json.load(open(p))
    d['mfma_pipe_utilisation_while_busy'] = d['SQ_VALU_MFMA_BUSY_CYCLES'] / (d['SQ_BUSY_CYCLES'] * 32)
    d['mfma_pipe_utilisation_note'] = ('SQ_VALU_MFMA_BUSY_CYCLES / (32 * SQ_BUSY_CYCLES): SQ_BUSY_CYCLES is summed over the 32 shader engines, the MFMA counter '
                                       'over the 1024 SIMDs (calibrated on the roofline kernel: profiles/r03_conv1_2_wino_pmc_sq.json agrees with its measured FLOP rate); '
                                       'single-stream run, every kernel of the process')
    for r in d['kernels_by_sq_busy']:
        r['mfma_pipe_utilisation'] = r['SQ_VALU_MFMA_BUSY_CYCLES'] / (r['SQ_BUSY_CYCLES'] * 32) if r['SQ_BUSY_CYCLES'] else None
    json.dump(d, open(p, 'w'), indent=1)
print('profiles/ refreshed')
