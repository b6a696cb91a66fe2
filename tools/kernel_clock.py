#!/usr/bin/env python3
"""The shader clock each kernel family of a step ran at: SQ_BUSY_CYCLES (summed over the 32 shader engines by rocprofv3) / 32 / the launch's
duration, from ONE rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES ... output directory (counter_collection.csv + kernel_trace.csv).
MFMA-dense kernels are power-limited below the 2.4 GHz the 157.3 TFLOP/s fp32-MFMA peak assumes; latency-bound kernels run at the full clock.

    python3 tools/kernel_clock.py gpurun_out/r6_final/step_SQ_VALU_MFMA_BUSY_CYCLES_SQ_BUSY_CYCLES"""
import csv
import glob
import os
import re
import sys
from collections import defaultdict

d = sys.argv[1]
cc = glob.glob(os.path.join(d, '*', '*counter_collection.csv'))[0]
kt = glob.glob(os.path.join(d, '*', '*kernel_trace.csv'))[0]


def short(n):
    n = re.sub(r'\(anonymous namespace\)::', '', n)
    n = re.sub(r'^void ', '', n)
    return n.split('(')[0][:70]


dur = {}
for r in csv.DictReader(open(kt)):
    dur[r['Dispatch_Id']] = (int(r['End_Timestamp']) - int(r['Start_Timestamp']), short(r['Kernel_Name']))
busy, mfma = defaultdict(float), defaultdict(float)
for r in csv.DictReader(open(cc)):
    if r['Counter_Name'] == 'SQ_BUSY_CYCLES':
        busy[r['Dispatch_Id']] += float(r['Counter_Value'])
    elif r['Counter_Name'] == 'SQ_VALU_MFMA_BUSY_CYCLES':
        mfma[r['Dispatch_Id']] += float(r['Counter_Value'])
fam = defaultdict(lambda: [0, 0.0, 0.0, 0.0])
for k, b in busy.items():
    if k not in dur or dur[k][0] <= 0:
        continue
    f = fam[dur[k][1]]
    f[0] += 1
    f[1] += b / 32
    f[2] += dur[k][0]
    f[3] += mfma.get(k, 0.0)
print('%-72s %8s %10s %8s %12s' % ('kernel', 'launches', 'total ms', 'GHz', 'MFMA pipe util'))
for name, (n, cyc, ns, mf) in sorted(fam.items(), key=lambda kv: -kv[1][2])[:40]:
    print('%-72s %8d %10.2f %8.3f %12.3f' % (name, n, ns / 1e6, cyc / ns, mf / (cyc * 1024) if cyc else 0.0))
