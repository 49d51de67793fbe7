#!/usr/bin/env python3
"""Compact timeline of ONE training step from a rocprofv3 kernel trace: every launch as (start ms from the step's start, duration us,
queue, kernel), plus per-queue busy time and, per kernel name, launches / total us.  A step ends with `multi_sgd_kernel`.

    python tools/r04/step_timeline.py <kernel_trace.csv> [--step -2] [--all] > profiles/r04_step_timeline.txt
"""
import csv
import re
import sys
from collections import defaultdict


def short(n):
    n = re.sub(r'\(.*', '', n)
    n = n.replace('void ', '').replace('mh::pl::', 'pl::').replace('mh::', '')
    n = re.sub(r'at::native::(\(anonymous namespace\)::)?', 'at::', n)
    return n[:78]


def main():
    path = sys.argv[1]
    which = int(sys.argv[sys.argv.index('--step') + 1]) if '--step' in sys.argv else -2
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], r['Queue_Id']))
    rows.sort()
    ends = [i for i, r in enumerate(rows) if 'multi_sgd_kernel' in r[2]]
    s = ends[which - 1] + 1
    e = ends[which] + 1
    seg = rows[s:e]
    t0 = rows[ends[which - 1]][1]
    print('step of %d launches, wall %.3f ms' % (len(seg), (seg[-1][1] - t0) / 1e6))
    qs = sorted(set(q for *_, q in seg))
    for q in qs:
        mine = [(a, b) for a, b, _, qq in seg if qq == q]
        print('queue %s: %d launches, kernel time %.3f ms, first start %.3f ms, last end %.3f ms' % (
            q, len(mine), sum(b - a for a, b in mine) / 1e6, (mine[0][0] - t0) / 1e6, (mine[-1][1] - t0) / 1e6))
    agg = defaultdict(lambda: [0, 0])
    for a, b, n, q in seg:
        agg[(q, short(n))][0] += 1
        agg[(q, short(n))][1] += b - a
    print('\nper kernel (queue, name): launches, total us')
    for (q, n), (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:60]:
        print('  q%s %-80s %4d %9.1f' % (q, n, c, t / 1e3))
    if '--all' in sys.argv:          # round 6: every launch, nothing folded (the glue between the kernels: which torch ops, in which order)
        print('\nevery launch (start ms, duration us, queue, kernel)')
        for a, b, n, q in seg:
            print('  %8.3f %8.1f  q%s  %s' % ((a - t0) / 1e6, (b - a) / 1e3, q, re.sub(r'\(.*', '', n).replace('void ', '')[:150]))
        return
    print('\ntimeline (start ms, duration us, queue, kernel); launches shorter than 20 us are folded into runs')
    run = None
    for a, b, n, q in seg:
        d = (b - a) / 1e3
        if d < 20:
            if run and run[2] == q:
                run[1] = b
                run[3] += 1
                run[4] += d
            else:
                if run:
                    print('  %8.3f %8.1f  q%s  [%d short launches, %.1f us of kernels]' % ((run[0] - t0) / 1e6, (run[1] - run[0]) / 1e3, run[2], run[3], run[4]))
                run = [a, b, q, 1, d]
            continue
        if run:
            print('  %8.3f %8.1f  q%s  [%d short launches, %.1f us of kernels]' % ((run[0] - t0) / 1e6, (run[1] - run[0]) / 1e3, run[2], run[3], run[4]))
            run = None
        print('  %8.3f %8.1f  q%s  %s' % ((a - t0) / 1e6, d, q, short(n)))
    if run:
        print('  %8.3f %8.1f  q%s  [%d short launches, %.1f us of kernels]' % ((run[0] - t0) / 1e6, (run[1] - run[0]) / 1e3, run[2], run[3], run[4]))


if __name__ == '__main__':
    main()
