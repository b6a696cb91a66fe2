#!/usr/bin/env python3
"""Idle gaps of the critical (main) stream in one training step, from a rocprofv3 kernel trace (rocpd .db or kernel_trace .csv): the time between
the end of a kernel and the start of the next one ON THE SAME STREAM, summed by size class, with the kernels around the largest ones.

    python3 tools/trace_gaps.py OUT/x_results.db"""
import sys

from trace_bins import family, load


def main(path):
    ev = sorted(load(path))
    ends = sorted(e[1] for e in ev if 'adadelta' in e[2])
    t1, t0 = ends[-1], ends[-4]
    win = [e for e in ev if e[1] > t0 and e[0] < t1]
    streams = sorted({e[3] for e in win}, key=lambda s: -sum(e[1] - e[0] for e in win if e[3] == s and family(e[2]) in ('Lf', 'Lb')))
    print('step window %.2f ms' % ((t1 - t0) / 1e6))
    for s in streams:
        ks = [e for e in win if e[3] == s]
        busy = sum(e[1] - e[0] for e in ks) / 1e6
        gaps = [(ks[i + 1][0] - ks[i][1], i) for i in range(len(ks) - 1)]
        cls = {'< 3 us': 0.0, '3-10 us': 0.0, '10-100 us': 0.0, '> 100 us': 0.0}
        cnt = dict.fromkeys(cls, 0)
        for g, _ in gaps:
            if g <= 0:
                continue
            k = '< 3 us' if g < 3e3 else '3-10 us' if g < 1e4 else '10-100 us' if g < 1e5 else '> 100 us'
            cls[k] += g / 1e6
            cnt[k] += 1
        print('stream %s: %d kernels, busy %.2f ms, span %.2f ms; gaps: %s' % (
            s, len(ks), busy, (ks[-1][1] - ks[0][0]) / 1e6, ', '.join('%s: %d = %.2f ms' % (k, cnt[k], v) for k, v in cls.items())))
        short = sum(1 for e in ks if e[1] - e[0] < 8e3)
        print('   kernels shorter than 8 us: %d (%.2f ms in total)' % (short, sum(e[1] - e[0] for e in ks if e[1] - e[0] < 8e3) / 1e6))
        for g, i in sorted(gaps, reverse=True)[:12]:
            print('   gap %8.1f us at %6.2f ms  after %-28s before %-28s' % (g / 1e3, (ks[i][1] - t0) / 1e6, family(ks[i][2])[:28], family(ks[i + 1][2])[:28]))


if __name__ == '__main__':
    main(sys.argv[1])
