#!/usr/bin/env python3
"""Per-millisecond view of one training step from a rocprofv3 kernel trace (rocpd .db or kernel_trace .csv):
for every bin and stream, the busy fraction and the dominant kernel family, plus the CU-filling (igemm) time.

    python3 tools/trace_bins.py OUT/x_results.db [bin_ms]
"""
import csv
import re
import sqlite3
import sys
from collections import defaultdict


def family(name):
    name = re.sub(r'\(anonymous namespace\)::', '', name)
    name = re.sub(r'^void ', '', name)
    m = re.match(r'igemm_kernel<(\w+), (\w+)', name)
    if m:
        return {'ConvKDenseK': 'cfwd', 'ConvMDenseM': 'cwg', 'DenseKDenseK': 'gNT', 'DenseMDenseM': 'gwg',
                'DenseKDenseM': 'gNN'}.get(m.group(1) + m.group(2), 'ig')
    name = re.sub(r'[<(].*', '', name)
    if name.startswith('lstm_fwd'):
        return 'Lf'
    if name.startswith('lstm_bwd'):
        return 'Lb'
    if name.startswith('attloc') or name.startswith('lstm_cell'):
        return 'dec'
    if name.startswith('splitk'):
        return 'red'
    if name.startswith('at::native'):
        return 'elt'
    return name.replace('_kernel', '')[:6]


def load(path):
    if path.endswith('.db'):
        c = sqlite3.connect(path)
        return [(s, e, n, str(st)) for s, e, n, st in c.execute('select start, end, name, stream_id from kernels')]
    return [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], r['Stream_Id'])
            for r in csv.DictReader(open(path))]


def main(path, bin_ms=1.0):
    ev = sorted(load(path))
    ends = sorted(e[1] for e in ev if 'adadelta' in e[2])
    t1, t0 = ends[-1], ends[-4]
    win = [e for e in ev if e[1] > t0 and e[0] < t1]
    streams = sorted({e[3] for e in win}, key=lambda s: -sum(1 for e in win if e[3] == s))
    print('step window %.2f ms; streams %s' % ((t1 - t0) / 1e6, streams))
    nb = int((t1 - t0) / 1e6 / bin_ms) + 1
    busy = [defaultdict(lambda: defaultdict(float)) for _ in range(nb)]
    for a, b, n, s in win:
        f = family(n)
        a, b = max(a, t0), min(b, t1)
        i = int((a - t0) / 1e6 / bin_ms)
        while a < b:
            edge = t0 + (i + 1) * bin_ms * 1e6
            d = min(b, edge) - a
            busy[i][s][f] += d
            a += d
            i += 1
    for i in range(nb):
        cols = []
        for s in streams:
            fam = busy[i][s]
            tot = sum(fam.values()) / (bin_ms * 1e6)
            top = sorted(fam.items(), key=lambda kv: -kv[1])[:2]
            cols.append('%3.0f%% %-14s' % (100 * tot, ','.join('%s' % k for k, _ in top)))
        print('%5.1f | %s' % (i * bin_ms, ' | '.join(cols)))


if __name__ == '__main__':
    main(sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 1.0)
