#!/usr/bin/env python3
"""Average every counter of a rocprofv3 counter_collection.csv over the launches of the kernels whose name contains argv[2] (summed over the
chip's instances as rocprofv3 reports them): one line per counter."""
import csv
import sys
from collections import defaultdict

path, pat = sys.argv[1], sys.argv[2]
vals = defaultdict(lambda: defaultdict(float))
with open(path) as f:
    for row in csv.DictReader(f):
        name = row.get('Kernel_Name') or row.get('Kernel-Name') or ''
        if pat not in name:
            continue
        vals[row['Counter_Name']][row['Dispatch_Id']] += float(row['Counter_Value'])
for c in sorted(vals):
    v = list(vals[c].values())
    print('%-40s launches %3d  avg %.6e  min %.6e  max %.6e' % (c, len(v), sum(v) / len(v), min(v), max(v)))
