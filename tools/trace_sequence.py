#!/usr/bin/env python3
"""The kernels of ONE single-stream training step in launch order with durations and the gap in front of each (rocprofv3 kernel_trace.csv of
RE2E_NO_OVERLAP=1 bench.py): where the critical path's small launches and host-side stalls are.

    python3 tools/trace_sequence.py OUT/*/*_kernel_trace.csv [min_us_to_print]"""
import csv
import re
import sys

rows = sorted(((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in csv.DictReader(open(sys.argv[1]))))
thr = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
ends = [i for i, r in enumerate(rows) if 'adadelta' in r[2]]
a, b = ends[-4] + 1, ends[-1] + 1          # the last whole step: behind the previous step's third Adadelta launch


def short(n):
    n = re.sub(r'\(anonymous namespace\)::', '', n)
    n = re.sub(r'^void ', '', n)
    return n.split('(')[0][:64]


t0 = rows[a][0]
tot_gap = tot_busy = 0.0
small = 0
for i in range(a, b):
    s, e, n = rows[i]
    gap = (s - rows[i - 1][1]) / 1e3
    dur = (e - s) / 1e3
    tot_gap += max(gap, 0.0)
    tot_busy += dur
    small += dur < 8.0
    if dur >= thr or gap >= thr:
        print('%8.3f ms  gap %7.1f us  %8.1f us  %s' % ((s - t0) / 1e6, gap, dur, short(n)))
print('step: %d launches, %.2f ms busy, %.2f ms of gaps, %d launches shorter than 8 us' % (b - a, tot_busy / 1e3, tot_gap / 1e3, small))
