#!/usr/bin/env python3
"""Every kernel of one stream in a time window of the last training step of a rocprofv3 kernel trace (rocpd .db), in start order:
start (ms from the step's start), duration, gap to the previous kernel of that stream, grid, name.

    python3 tools/trace_list.py OUT/x_results.db START_MS END_MS [stream-rank] [min-us]"""
import re
import sqlite3
import sys
from collections import defaultdict


def main(path, a, b, rank=0, min_us=0.0):
    c = sqlite3.connect(path)
    ev = sorted(c.execute('select start,end,name,stream_id,grid_x,grid_y,grid_z,workgroup_x from kernels').fetchall())
    ends = sorted(e[1] for e in ev if 'adadelta' in e[2])
    t1, t0 = ends[-1], ends[-4]
    win = [e for e in ev if e[1] > t0 and e[0] < t1]
    cnt = defaultdict(int)
    for e in win:
        cnt[e[3]] += 1
    stream = sorted(cnt, key=lambda s: -cnt[s])[rank]
    prev = None
    for e in win:
        if e[3] != stream:
            continue
        s = (e[0] - t0) / 1e6
        if a <= s < b and ((e[1] - e[0]) / 1e3 >= min_us or (prev and (e[0] - prev) / 1e3 >= 20.0)):       # long kernels and whatever follows an idle gap
            n = re.sub(r'\(anonymous namespace\)::', '', e[2])
            n = re.sub(r'^void ', '', n)[:64]
            print('%8.3f ms  %8.1f us  gap %7.1f us  grid %5d x%3d x%3d  %s' % (s, (e[1] - e[0]) / 1e3, (e[0] - prev) / 1e3 if prev else 0.0,
                                                                         e[4] // e[7], e[5], e[6], n))
        prev = e[1]


if __name__ == '__main__':
    main(sys.argv[1], float(sys.argv[2]), float(sys.argv[3]), int(sys.argv[4]) if len(sys.argv) > 4 else 0, float(sys.argv[5]) if len(sys.argv) > 5 else 0.0)
