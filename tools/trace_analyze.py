#!/usr/bin/env python3
"""Timeline analysis of a rocprofv3 --kernel-trace csv of bench.py: per queue busy time, overlap, gaps (last step)."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# find step boundaries by the fused SGD kernel
sgd = [i for i, r in enumerate(rows) if 'multi_sgd_kernel' in r['Kernel_Name']]
a, b = sgd[-2] + 1, sgd[-1] + 1
step = rows[a:b]
t0, t1 = int(step[0]['Start_Timestamp']), int(step[-1]['End_Timestamp'])
print('step wall %.3f ms, %d kernels' % ((t1 - t0) / 1e6, len(step)))
byq = collections.defaultdict(list)
for r in step: byq[r['Queue_Id']].append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']))
def union(iv):
    iv = sorted(iv); tot = 0; cs, ce = iv[0][0], iv[0][1]
    for s, e in iv[1:]:
        if s > ce: tot += ce - cs; cs, ce = s, e
        else: ce = max(ce, e)
    return tot + ce - cs
allbusy = union([(s, e) for q in byq.values() for s, e, _ in q])
print('GPU busy (any queue) %.3f ms, idle %.3f ms' % (allbusy / 1e6, (t1 - t0 - allbusy) / 1e6))
for q, iv in byq.items():
    print('queue', q, 'kernels', len(iv), 'busy %.3f ms' % (union([(s, e) for s, e, _ in iv]) / 1e6), 'sum %.3f' % (sum(e - s for s, e, _ in iv) / 1e6))
# time where only small (<30us) kernels are running vs big
ev = []
for q in byq.values():
    for s, e, n in q: ev.append((s, e, n))
# gaps on the whole GPU > 5us
iv = sorted((s, e) for s, e, _ in ev); gaps = []; ce = iv[0][1]
for s, e in iv[1:]:
    if s > ce + 5000: gaps.append((s - ce, ce - t0))
    ce = max(ce, e)
print('idle gaps >5us: %d totalling %.3f ms; largest:' % (len(gaps), sum(g for g, _ in gaps) / 1e6), sorted(gaps, reverse=True)[:8])
# phases: print coarse timeline of main-queue big kernels
mainq = max(byq, key=lambda q: sum(e - s for s, e, _ in byq[q]))
print('main queue', mainq)
for q, ivs in byq.items():
    if q == mainq: continue
    s0 = min(s for s, _, _ in ivs); e0 = max(e for _, e, _ in ivs)
    segs = []; cur = None
    for s, e, n in sorted(ivs):
        if cur is None or s - cur[1] > 200000: 
            if cur: segs.append(cur)
            cur = [s, e, 1]
        else: cur[1] = max(cur[1], e); cur[2] += 1
    segs.append(cur)
    print('side queue', q, 'segments (start ms, end ms, kernels):', [(round((s - t0) / 1e6, 2), round((e - t0) / 1e6, 2), n) for s, e, n in segs])
