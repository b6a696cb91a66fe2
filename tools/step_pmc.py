#!/usr/bin/env python3
"""Whole-step counters of `RE2E_NO_OVERLAP=1 bench.py --steps 3 --no-cpu-baseline` (three separate rocprofv3 --pmc passes:
SQ_VALU_MFMA_BUSY_CYCLES + SQ_BUSY_CYCLES, FETCH_SIZE, WRITE_SIZE) summed per kernel and over the run.
usage: python3 tools/step_pmc.py gpurun_out/r3_final > step_pmc.json"""
import csv
import glob
import json
import os
import re
import sys

O = sys.argv[1]


def short(n):
    n = re.sub(r'\(anonymous namespace\)::', '', n)
    n = re.sub(r'^void ', '', n)
    return n.split('(')[0][:110]


def collect(tag):
    per = {}
    for f in glob.glob(os.path.join(O, 'step_' + tag, '*', '*counter_collection.csv')):
        for r in csv.DictReader(open(f)):
            k = per.setdefault(short(r['Kernel_Name']), {})
            c = k.setdefault(r['Counter_Name'], [0, 0.0])
            c[0] += 1
            c[1] += float(r['Counter_Value'])
    return per


sq, fe, wr = collect('SQ_VALU_MFMA_BUSY_CYCLES_SQ_BUSY_CYCLES'), collect('FETCH_SIZE'), collect('WRITE_SIZE')
steps = max(1, sum(v.get('SQ_BUSY_CYCLES', [0])[0] for k, v in sq.items() if 'adadelta_kernel' in k) // 3)   # 3 Adadelta launches (3 nets) per step
tot = lambda per, c: sum(v[c][1] for v in per.values() if c in v)
mf, busy = tot(sq, 'SQ_VALU_MFMA_BUSY_CYCLES'), tot(sq, 'SQ_BUSY_CYCLES')
fetch_b, write_b = tot(fe, 'FETCH_SIZE') * 1024 * 2, tot(wr, 'WRITE_SIZE') * 1024
rows = []
for k in sq:
    m, b = sq[k].get('SQ_VALU_MFMA_BUSY_CYCLES', [0, 0.0]), sq[k].get('SQ_BUSY_CYCLES', [0, 0.0])
    rows.append({'kernel': k, 'launches': b[0], 'SQ_VALU_MFMA_BUSY_CYCLES': m[1], 'SQ_BUSY_CYCLES': b[1], 'mfma_over_busy': (m[1] / b[1]) if b[1] else None,
                 'FETCH_bytes_x2': fe.get(k, {}).get('FETCH_SIZE', [0, 0.0])[1] * 2048, 'WRITE_bytes': wr.get(k, {}).get('WRITE_SIZE', [0, 0.0])[1] * 1024})
rows.sort(key=lambda r: -r['SQ_BUSY_CYCLES'])
json.dump({'command': 'RE2E_NO_OVERLAP=1 rocprofv3 --kernel-trace --pmc <set> -- python3 bench.py --steps 3 --no-cpu-baseline ; three passes: '
                      '{SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES}, {FETCH_SIZE}, {WRITE_SIZE} (tools/final_evidence_r3.sh)',
           'scope': 'every kernel the process launched: CMVN passes, warm-up and the 3 timed steps (%d steps counted from the Adadelta launches); '
                    'ratios are over the whole run, per-step figures divide by that step count and so include the CMVN passes\' share' % steps,
           'steps_in_run': steps,
           'SQ_VALU_MFMA_BUSY_CYCLES': mf, 'SQ_BUSY_CYCLES': busy, 'mfma_busy_over_sq_busy': mf / busy if busy else None,
           'mfma_instruction_equivalents_32x32x2': mf / 64,
           'executed_mfma_flop_per_step': mf / 64 * 2 * 32 * 32 * 2 * 64 / 64 / steps if steps else None,
           'FETCH_bytes_corrected_x2': fetch_b, 'WRITE_bytes': write_b, 'hbm_bytes_per_step': (fetch_b + write_b) / steps,
           'corrections': 'FETCH_SIZE is reported in KiB and at 1/2 of the bytes read on gfx950 (profiles/r02_fetch_size_calibration.json): x1024 x2; WRITE_SIZE KiB x1024',
           'kernels_by_sq_busy': rows[:40]}, sys.stdout, indent=1)
