"""Idle-time analysis of a rocprofv3 kernel trace of the bench step (tools/README.md).

  python tools/trace_gaps.py <kernel_trace.csv> [--steps N] [--top K]

Takes the LAST N steps of the trace (a step ends with the optimizer kernel `multi_sgd_kernel`), merges the kernel intervals of
all queues, and prints per step: wall time, busy time (union of intervals), idle time, and the K largest idle gaps with the
kernel before / after each -- the places where the host cannot keep the GPU fed.  With more than one queue (HIP streams) it also
prints, per step, how long ONLY each queue had a kernel running, how long several had, and what the queues ran: the branch that
runs alone for long is the step's critical path (round 3: the context branch on the second stream, 4.3 ms per step)."""
import csv
import sys
from collections import defaultdict


def main():
    path = sys.argv[1]
    nsteps = int(sys.argv[sys.argv.index('--steps') + 1]) if '--steps' in sys.argv else 3
    top = int(sys.argv[sys.argv.index('--top') + 1]) if '--top' in sys.argv else 25
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], r['Queue_Id']))
    rows.sort()
    ends = [i for i, r in enumerate(rows) if 'multi_sgd_kernel' in r[2]]
    if len(ends) < nsteps + 1:
        raise SystemExit('not enough steps in the trace')
    agg = defaultdict(lambda: [0, 0.0])
    for s in range(len(ends) - nsteps, len(ends)):
        seg = rows[ends[s - 1] + 1:ends[s] + 1]
        t0, t1 = rows[ends[s - 1]][1], seg[-1][1]
        busy, cur_end, gaps = 0, t0, []
        prev = rows[ends[s - 1]][2]
        for a, b, name, q in seg:
            if a > cur_end:
                gaps.append((a - cur_end, prev, name))
                busy += b - a
                cur_end = b
                prev = name
            else:
                if b > cur_end:
                    busy += b - cur_end
                    cur_end = b
                    prev = name
        print(f'step {s}: wall {(t1 - t0) / 1e6:.3f} ms, busy {busy / 1e6:.3f} ms, idle {(t1 - t0 - busy) / 1e6:.3f} ms, '
              f'{len(seg)} launches, sum of kernel durations {sum(b - a for a, b, _, _ in seg) / 1e6:.3f} ms, '
              f'queues {sorted(set(q for *_, q in seg))}')
        queues = sorted(set(q for *_, q in seg))
        if len(queues) > 1:
            ev = sorted([(a, 1, q) for a, b, _, q in seg] + [(b, -1, q) for a, b, _, q in seg])
            active, last, alone, several = defaultdict(int), t0, defaultdict(int), 0
            for t, d, q in ev:
                on = [k for k, v in active.items() if v > 0]
                if len(on) == 1:
                    alone[on[0]] += t - last
                elif len(on) > 1:
                    several += t - last
                active[q] += d
                last = t
            per_q = defaultdict(float)
            for a, b, _, q in seg:
                per_q[q] += b - a
            print('   ' + ', '.join(f'only queue {q}: {alone[q] / 1e6:.2f} ms (its kernels: {per_q[q] / 1e6:.2f} ms)' for q in queues)
                  + f', several queues at once: {several / 1e6:.2f} ms')
        for g, p, n in gaps:
            k = (p[:48], n[:48])
            agg[k][0] += 1
            agg[k][1] += g / 1e3
        if s == len(ends) - 1:
            print('  largest gaps of the last step (us, after -> before):')
            for g, p, n in sorted(gaps, reverse=True)[:top]:
                print(f'   {g / 1e3:8.1f}  {p[:60]}  ->  {n[:60]}')
    print(f'idle time by (previous kernel -> next kernel), all {nsteps} steps, us per step:')
    for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
        print(f'   {t / nsteps:8.1f}  x{c / nsteps:5.1f}  {k[0]}  ->  {k[1]}')


if __name__ == '__main__':
    main()
