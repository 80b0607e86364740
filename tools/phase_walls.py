#!/usr/bin/env python3
"""dev helper: wall time per phase of one training step from a rocprofv3 kernel trace of the CONCURRENT step (two forward chains,
data-gradient chain + weight-gradient side stream).  Phases are delimited by the 3x3 halo kernels, whose grids name the dense block.
usage: phase_walls.py kernel_trace.csv [step index]"""
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted(((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], int(r['Grid_Size_X']) // max(1, int(r['Workgroup_Size_X'])), int(r['Grid_Size_Y']), int(r['Grid_Size_Z'])) for r in rows))
adam = [i for i, e in enumerate(ev) if 'adam_kernel' in e[2]]
k = int(sys.argv[2]) if len(sys.argv) > 2 else 3
a, b = adam[2 * k - 1] + 1, adam[2 * k + 1] + 1          # two adam launches per step
seg = ev[a:b]
t0 = seg[0][0]
def blk(e):
    n, gx = e[2], e[3]
    if 'halo' not in n: return None
    return {100: 1, 25: 2 if '<16' in n else 3, 9: 4}.get(gx)
marks = []
for e in seg:
    n = e[2]
    ph = None
    if 'halo_fwd' in n: ph = 'fwd b%d' % blk(e)
    elif 'halo_dgrad' in n: ph = 'bwd b%d' % blk(e)
    elif 'prep_rotate' in n: ph = 'fwd stem'
    elif 'feat_kernel' in n: ph = 'head'
    elif 'value_bwd' in n: ph = 'bwd head'
    elif 'pool0_bwd' in n: ph = 'bwd stem'
    elif 'adam' in n: ph = 'adam'
    if ph and (not marks or marks[-1][0] != ph):
        if ph not in [m[0] for m in marks]: marks.append((ph, e[0]))
marks.append(('end', max(e[1] for e in seg)))
print('step wall %.2f ms' % ((marks[-1][1] - t0) / 1e6))
for (p, t), (_, t2) in zip(marks, marks[1:]):
    ks = [e for e in seg if t <= e[0] < t2]
    busy = sum(e[1] - e[0] for e in ks)
    print('%-10s starts %7.3f  wall %6.3f ms  kernels %4d  sum of durations %6.3f ms' % (p, (t - t0) / 1e6, (t2 - t) / 1e6, len(ks), busy / 1e6))
