#!/usr/bin/env python3
"""Kernel table of a time window of the last training step in a rocprofv3 kernel trace (rocpd .db).

    python3 tools/trace_window.py OUT/x_results.db START_MS END_MS [stream-rank]
stream-rank 0 = the stream with the most kernels in the step (the main stream)."""
import re
import sqlite3
import sys
from collections import defaultdict


def main(path, a, b, rank=0):
    c = sqlite3.connect(path)
    ev = sorted(c.execute('select start,end,name,stream_id,grid_x,grid_y,grid_z,workgroup_x from kernels').fetchall())
    ends = sorted(e[1] for e in ev if 'adadelta' in e[2])
    t1, t0 = ends[-1], ends[-4]
    win = [e for e in ev if e[1] > t0 and e[0] < t1]
    cnt = defaultdict(int)
    for e in win:
        cnt[e[3]] += 1
    stream = sorted(cnt, key=lambda s: -cnt[s])[rank]
    ks = [e for e in win if e[3] == stream and a <= (e[0] - t0) / 1e6 < b]
    d = defaultdict(lambda: [0, 0.0, None])
    gaps = 0.0
    for i, e in enumerate(ks):
        n = re.sub(r'\(anonymous namespace\)::', '', e[2])
        n = re.sub(r'^void ', '', n)[:72]
        d[n][0] += 1
        d[n][1] += (e[1] - e[0]) / 1e3
        d[n][2] = (e[4] // e[7], e[5], e[6], e[7])
        if i:
            gaps += max(0, e[0] - ks[i - 1][1]) / 1e3
    tot = sum(v[1] for v in d.values())
    print('step %.2f ms; stream %s window %.1f-%.1f ms: %d kernels, busy %.0f us, gaps %.0f us' % ((t1 - t0) / 1e6, stream, a, b, len(ks), tot, gaps))
    for n, v in sorted(d.items(), key=lambda kv: -kv[1][1])[:24]:
        print('  %8.1f us  n=%4d avg %6.1f  grid %s  %s' % (v[1], v[0], v[1] / v[0], v[2], n))


if __name__ == '__main__':
    main(sys.argv[1], float(sys.argv[2]), float(sys.argv[3]), int(sys.argv[4]) if len(sys.argv) > 4 else 0)
