"""Per-queue timeline of a rocprofv3 --kernel-trace CSV of bench.py: for every engine (queue) the time between successive
k_finish_step ends (= one step of that engine's chain), split into kernel time by family and gaps between kernels.
usage: chain_timeline.py <dir or csv> [first_step last_step]"""
import csv
import glob
import os
import sys
from collections import defaultdict

import numpy as np

path = sys.argv[1]
if os.path.isdir(path):
    path = sorted(glob.glob(os.path.join(path, '**', '*kernel_trace.csv'), recursive=True))[0]
lo, hi = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (100, 160)
rows = defaultdict(list)
with open(path) as f:
    for r in csv.DictReader(f):
        rows[r['Queue_Id']].append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0].split('<')[0]))
for q, ks in sorted(rows.items()):
    ks.sort()
    ends = [i for i, k in enumerate(ks) if k[2] == 'k_finish_step']
    if len(ends) < hi + 1:
        continue
    fam = defaultdict(list)
    gaps, tot, nk = [], [], []
    for s in range(lo, hi):
        a, b = ends[s] + 1, ends[s + 1] + 1
        seg = ks[a:b]
        tot.append((seg[-1][1] - ks[a - 1][1]) / 1e3)
        g = 0.0
        prev = ks[a - 1][1]
        per = defaultdict(float)
        for st, en, nm in seg:
            g += max(0, st - prev) / 1e3
            per[nm] += (en - st) / 1e3
            prev = en
        gaps.append(g)
        nk.append(len(seg))
        for nm, v in per.items():
            fam[nm].append(v)
    print(f'queue {q}: steps {lo}..{hi}: chain {np.mean(tot):.1f} us/step, {np.mean(nk):.1f} kernels, gaps {np.mean(gaps):.1f} us')
    for nm, v in sorted(fam.items(), key=lambda kv: -np.sum(kv[1])):
        print(f'    {nm:36s} {np.sum(v) / (hi - lo):8.1f} us/step')
