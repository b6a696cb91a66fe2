#!/usr/bin/env python3
"""Print a per-kernel summary of a rocprofv3 *_kernel_stats.csv (per-step figures for bench.py runs)."""
import csv
import re
import sys

path, steps = sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
rows = list(csv.DictReader(open(path)))
tot = sum(float(r['TotalDurationNs']) for r in rows)
print('total kernel time %.1f ms (%.1f ms per step over %g steps)' % (tot / 1e6, tot / 1e6 / steps, steps))
for r in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 30]:
    n = re.sub(r'\(anonymous namespace\)::', '', r['Name'])
    n = re.sub(r'\(.*', '', n)[:90]
    print('%8.2f ms/step %5.1f%% calls/step=%7.1f avg=%9.1f us  %s' % (float(r['TotalDurationNs']) / 1e6 / steps, float(r['Percentage']),
                                                                       float(r['Calls']) / steps, float(r['AverageNs']) / 1e3, n))
