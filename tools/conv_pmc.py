#!/usr/bin/env python3
"""Per-launch counters of the roofline kernels (tools/roofline_conv.py = bench.py conv_roofline) from the passes of
tools/final_evidence_r3.sh.  usage: python3 tools/conv_pmc.py gpurun_out/r3_final > conv_pmc.json"""
import csv
import glob
import json
import os
import sys

O = sys.argv[1]
res = {}
for kern in ('wino_conv3x3_kernel', 'wino_weights_kernel', 'conv3x3_halo_kernel'):
    out = {}
    for d in glob.glob(os.path.join(O, 'conv_*')):
        for f in glob.glob(os.path.join(d, '*', '*counter_collection.csv')):
            for r in csv.DictReader(open(f)):
                if kern in r['Kernel_Name']:
                    out.setdefault(r['Kernel_Name'].split('(')[0][-60:], {}).setdefault(r['Counter_Name'], []).append(float(r['Counter_Value']))
    res[kern] = {n: {c: {'launches': len(v), 'avg': sum(v) / len(v), 'min': min(v), 'max': max(v)} for c, v in cs.items()} for n, cs in out.items()}
for f in glob.glob(os.path.join(O, 'roofline_conv_kernel_stats.csv')):
    res['stats'] = [r for r in csv.DictReader(open(f)) if 'conv3x3' in r['Name'] or 'wino' in r['Name']]
json.dump(res, sys.stdout, indent=1)
