#!/usr/bin/env python3
"""Vector-L1 accesses per cycle and CU of every kernel family of a step: TCP_TOTAL_CACHE_ACCESSES_sum / 256 CUs over SQ_BUSY_CYCLES / 32, from ONE
rocprofv3 --kernel-trace --pmc TCP_TOTAL_CACHE_ACCESSES_sum SQ_BUSY_CYCLES SQ_INSTS_VMEM_RD output directory.  The L1 serves about one access per
cycle: a kernel near that is bound by its access PATTERN (profiles/r06_wino_attribution.md), whatever its bytes.

    python3 tools/kernel_l1_rate.py gpurun_out/r6_l1/pass"""
import csv
import glob
import os
import re
import sys
from collections import defaultdict

d = sys.argv[1]
cc = glob.glob(os.path.join(d, '*', '*counter_collection.csv'))[0]


def short(n):
    n = re.sub(r'\(anonymous namespace\)::', '', n)
    n = re.sub(r'^void ', '', n)
    return n.split('(')[0][:70]


per = defaultdict(lambda: defaultdict(float))
name = {}
for r in csv.DictReader(open(cc)):
    per[r['Dispatch_Id']][r['Counter_Name']] += float(r['Counter_Value'])
    name[r['Dispatch_Id']] = short(r['Kernel_Name'])
fam = defaultdict(lambda: [0, 0.0, 0.0, 0.0])
for k, c in per.items():
    f = fam[name[k]]
    f[0] += 1
    f[1] += c.get('TCP_TOTAL_CACHE_ACCESSES_sum', 0.0)
    f[2] += c.get('SQ_BUSY_CYCLES', 0.0) / 32
    f[3] += c.get('SQ_INSTS_VMEM_RD', 0.0)
print('%-72s %8s %12s %16s %18s' % ('kernel', 'launches', 'Mcycles', 'L1 acc/cycle/CU', 'acc per load instr'))
for n, (cnt, acc, cyc, rd) in sorted(fam.items(), key=lambda kv: -kv[1][2])[:45]:
    print('%-72s %8d %12.2f %16.3f %18.1f' % (n, cnt, cyc / 1e6, acc / 256 / cyc if cyc else 0.0, acc / rd if rd else 0.0))
