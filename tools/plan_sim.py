"""The 3D IoU's launch plan in numbers (development aid; CPU only): tasks,
(task, chunk) steps, the pairs' span unions / overlaps, and how many lanes of a
task hold a pair whose two tracks meet in a chunk (the general step's lanes).
    python tools/plan_sim.py [videos]        
"""
import os, sys, time, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from tao_amodal_amd import flatten, engine
from tao_amodal_amd.synth import synth
V = int(sys.argv[1]) if len(sys.argv) > 1 else 300
t=time.time()
gt, dt = synth(seed=20240807, V=V, F=300, C=1203, dets_per_frame=50)
dt.track_id, _ = flatten.make_track_ids_unique(dt)
ft = flatten.flatten_tao(gt, dt)
print('built', time.time()-t)
meta, sides, n_slots = engine.track_meta(ft)
tasks, rows, pairs, out = engine.track_iou_plan(ft, meta)
first, last = meta[:,0], meta[:,1]
P=8
tot=0; useful_both=0; useful_any=0; lanes=0
for (r0,nr,p0,npair) in tasks:
    rr = rows[r0:r0+nr]
    lo = first[rr].min() & ~7; hi = last[rr].max()
    nch = (hi-lo)//P+1
    tot += nch
print('tasks', len(tasks), 'task-chunks', tot, 'per task', tot/len(tasks), 'pairs', len(pairs), 'rows', len(rows))
# per-pair union / intersection chunk counts
pd = pairs & 0xFF; pg = (pairs>>8)&0xFF
tid = np.repeat(np.arange(len(tasks)), tasks[:,3])
rd = rows[tasks[tid,0]+pd]; rg = rows[tasks[tid,0]+pg]
fd, ld, fg, lg = first[rd], last[rd], first[rg], last[rg]
un = (np.maximum(ld,lg)//P - np.minimum(fd,fg)//P + 1)
both = np.maximum(0, np.minimum(ld,lg)//P - np.maximum(fd,fg)//P + 1)
print('sum over pairs of span-union chunks /64 =', un.sum()/64, ' both-chunks/64 =', both.sum()/64)
# distribution of the number of 'both' lanes per (task, chunk)
from collections import Counter
hist = Counter()
cd0 = np.maximum(fd,fg)//P; cd1 = np.minimum(ld,lg)//P
p_first = np.zeros(len(tasks),dtype=np.int64)
nchs = np.zeros(len(tasks),dtype=np.int64)
for k,(r0,nr,p0,npair) in enumerate(tasks):
    rr = rows[r0:r0+nr]
    lo = (first[rr].min() & ~7)//P; hi = last[rr].max()//P
    cnt = np.zeros(hi-lo+2, dtype=np.int64)
    a = cd0[p0:p0+npair]-lo; b = cd1[p0:p0+npair]-lo
    ok = b>=a
    np.add.at(cnt, a[ok], 1); np.add.at(cnt, b[ok]+1, -1)
    c = np.cumsum(cnt)[:hi-lo+1]
    for v in c: hist[int(v)] += 1
tot = sum(hist.values())
cum=0
for lo_,hi_ in [(0,0),(1,8),(9,16),(17,24),(25,32),(33,48),(49,64)]:
    n = sum(v for k,v in hist.items() if lo_<=k<=hi_)
    print(f'both-lanes {lo_:2d}-{hi_:2d}: {n:7d} {100*n/tot:5.1f}%')

# rounds of 8 rows the stagers work through per (task, chunk): groups of the task's rows (in
# plan order) that hold a present row, against ceil(present rows / 8) if present rows were compacted
r_now = r_cmp = n_tc = 0
for (r0, nr, p0, npair) in tasks:
    rr = rows[r0:r0 + nr]
    f, l = first[rr] // P, last[rr] // P
    lo = (first[rr].min() & ~7) // P
    hi = last[rr].max() // P
    c = np.arange(lo, hi + 1)[:, None]
    pres = (f[None, :] <= c) & (l[None, :] >= c)            # [chunks, rows]
    pad = np.zeros((pres.shape[0], 32), bool); pad[:, :nr] = pres
    r_now += pad.reshape(-1, 4, 8).any(2).sum()
    r_cmp += np.ceil(pres.sum(1) / 8).sum()
    n_tc += pres.shape[0]
print('rounds per (task, chunk): %.2f as planned, %.2f with the present rows compacted' % (r_now / n_tc, r_cmp / n_tc))
