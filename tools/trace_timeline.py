#!/usr/bin/env python3
"""Coarse timeline of one training step from a rocprofv3 kernel trace: per stream, consecutive kernels are
merged into phases (split at gaps > 30 us or when the kernel family changes); prints start, duration, busy
time and the kernel family of each phase so that the critical path and cross-stream waits can be read off.

    python3 tools/trace_timeline.py OUT/*/*_kernel_trace.csv [min_phase_us]
"""
import csv
import re
import sys
from collections import defaultdict


def family(name):
    name = re.sub(r'\(anonymous namespace\)::', '', name)
    name = re.sub(r'^void ', '', name)
    m = re.match(r'igemm_kernel<(\w+), (\w+)', name)
    if m:
        return 'igemm'
    name = re.sub(r'[<(].*', '', name)
    if name.startswith('lstm_fwd') or name.startswith('lstm_bwd'):
        return name[:8]
    if name.startswith('attloc') or name.startswith('lstm_cell') or name in ('splitk_reduce_kernel',):
        return 'dec/att' if not name.startswith('splitk') else 'reduce'
    if name.startswith('at::native'):
        return 'torch-eltwise'
    return name.replace('_kernel', '')


def main(path, min_us=200.0):
    ev = []
    for r in csv.DictReader(open(path)):
        ev.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], r['Stream_Id']))
    ev.sort()
    ends = sorted(e[1] for e in ev if 'adadelta' in e[2])
    t1, t0 = ends[-1], ends[-4]          # 3 optimizer launches per step (enhancer, ASR, D)
    win = [e for e in ev if e[0] >= t0 and e[1] <= t1 + 1000]
    print('step window %.2f ms' % ((t1 - t0) / 1e6))
    by = defaultdict(list)
    for e in win:
        by[e[3]].append(e)
    for sid, evs in sorted(by.items(), key=lambda kv: -len(kv[1])):
        print('\nstream %s: %d kernels, busy %.1f ms' % (sid, len(evs), sum(e[1] - e[0] for e in evs) / 1e6))
        phases = []
        for a, b, n, _ in evs:
            f = family(n)
            if phases and (a - phases[-1][1] < 30000) and (phases[-1][3] == f or f in ('reduce', 'torch-eltwise')):
                p = phases[-1]
                p[1] = max(p[1], b); p[2] += b - a; p[4] += 1
            else:
                phases.append([a, b, b - a, f, 1])
        # merge small phases into "misc"
        out = []
        for p in phases:
            if (p[1] - p[0]) / 1e3 < min_us and out and out[-1][3].startswith('misc') and p[0] - out[-1][1] < 200000:
                q = out[-1]; q[1] = p[1]; q[2] += p[2]; q[4] += p[4]
            elif (p[1] - p[0]) / 1e3 < min_us:
                out.append([p[0], p[1], p[2], 'misc', p[4]])
            else:
                out.append(p)
        prev_end = t0
        for a, b, busy, f, n in out:
            gap = (a - prev_end) / 1e3
            print('  t=%7.2f ms  dur %7.2f ms  busy %6.2f  n=%5d  gap-before %7.1f us  %s' % ((a - t0) / 1e6, (b - a) / 1e6, busy / 1e6, n, gap, f))
            prev_end = b


if __name__ == '__main__':
    main(sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 200.0)
