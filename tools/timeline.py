#!/usr/bin/env python3
"""dev helper: timeline view of a rocprofv3 kernel trace - per training step: wall, union-busy time, idle gaps, time with
one / two+ kernels in flight, and the kernels adjacent to the largest gaps.
usage: timeline.py kernel_trace.csv"""
import csv, sys, re, collections
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted(((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], r['Queue_Id']) for r in rows))
def short(n):
    n = re.sub(r'smg::', '', n); n = re.sub(r'\(.*', '', n); return n[:70]
# steps are delimited by adam_kernel launches
ends = [i for i, e in enumerate(ev) if 'adam_kernel' in e[2]]
print(len(ev), 'dispatches; adam launches at', ends[:20])
# take a middle step: between adam i and adam i+2 (two adam launches per step?) - print gaps between steps first
def analyze(a, b, label):
    seg = ev[a:b]
    t0, t1 = seg[0][0], max(e[1] for e in seg)
    pts = []
    for s, e, n, q in seg: pts += [(s, 1), (e, -1)]
    pts.sort()
    busy1 = busy2 = idle = 0; cur = 0; last = t0
    for t, d in pts:
        dt = t - last
        if cur == 0: idle += dt
        elif cur == 1: busy1 += dt
        else: busy2 += dt
        cur += d; last = t
    print(f'{label}: {len(seg)} kernels, wall {(t1-t0)/1e6:.2f} ms, idle {idle/1e6:.2f} ms, one kernel {busy1/1e6:.2f} ms, two+ {busy2/1e6:.2f} ms, sum of durations {sum(e[1]-e[0] for e in seg)/1e6:.2f} ms')
    # gap histogram: idle intervals
    gaps = []; cur = 0; last = t0; prev = None
    active_end = t0
    for s, e, n, q in seg:
        if s > active_end: gaps.append((s - active_end, short(prev), short(n)))
        if e > active_end: active_end = e; prev = n
    gaps.sort(reverse=True)
    print('   gaps: n=%d total %.2f ms; >20us: %d; 5-20us: %d; <5us: %d' % (len(gaps), sum(g[0] for g in gaps)/1e6, sum(g[0] > 20000 for g in gaps), sum(5000 < g[0] <= 20000 for g in gaps), sum(g[0] <= 5000 for g in gaps)))
    for g in gaps[:8]: print('      %.1f us after %s before %s' % (g[0]/1e3, g[1], g[2]))
    byq = collections.Counter(q for *_, q in seg)
    print('   queues:', dict(byq))
# steps of the timed loop (device-resident inputs): the 4th and 5th Adam pairs; under rocprofv3 the HOST is slower than in
# a plain run (every launch is intercepted), so gaps at the step boundaries and before Adam are the profiler's, not the step's
if len(ends) >= 12:
    analyze(ends[7] + 1, ends[9] + 1, 'step')
    analyze(ends[9] + 1, ends[11] + 1, 'next')
elif len(ends) >= 6:
    analyze(ends[3] + 1, ends[5] + 1, 'step')

def solo(a, b):
    """time with exactly one kernel in flight, by kernel name"""
    seg = ev[a:b]
    pts = []
    for i, (s, e, n, q) in enumerate(seg): pts += [(s, 1, i), (e, -1, i)]
    pts.sort()
    live = set(); last = pts[0][0]; acc = collections.Counter(); acc2 = collections.Counter()
    for t, d, i in pts:
        if len(live) == 1: acc[short(seg[next(iter(live))][2])] += t - last
        elif len(live) >= 2:
            for j in live: acc2[short(seg[j][2])] += (t - last) / len(live)
        if d > 0: live.add(i)
        else: live.discard(i)
        last = t
    print('   time alone on the GPU (ms):')
    for n, v in acc.most_common(14): print('      %7.3f  %s' % (v / 1e6, n))
    print('   time shared (ms, split evenly):')
    for n, v in acc2.most_common(10): print('      %7.3f  %s' % (v / 1e6, n))
if len(ends) >= 12: solo(ends[7] + 1, ends[9] + 1)
elif len(ends) >= 6: solo(ends[3] + 1, ends[5] + 1)
