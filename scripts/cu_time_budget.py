"""CU-time budget of the benchmark's steady state from a rocprofv3 --kernel-trace CSV of bench.py: for
every kernel family the launches, workgroups, the CUs a launch can hold at most (workgroups / workgroups that fit one CU by LDS
and threads), duration, CU x time; per engine (queue) the chain of a step; and what the other engines run while one engine's
projection kernel is in flight.   usage: cu_time_budget.py <rocprof dir> [timed steps per engine = 20]"""
import csv, glob, sys
from collections import defaultdict

CUS, LDS_CU, THREADS_CU = 256, 160 * 1024, 2048
f = glob.glob(sys.argv[1] + "/*/*kernel_trace.csv")[0]
skip = float(sys.argv[2]) if len(sys.argv) > 2 else 20
rows = list(csv.DictReader(open(f)))


def fam(name):
    for key, label in (('k_rproj', 'projection (k_rproj)'), ('k_polar', 'polar (complex)'), ('k_proj', 'projection products'),
                       ('k_sht_chain', 'chained inverse -> forward SHT'), ('k_sht_fwd', 'forward SHT'), ('k_sht_inv', 'inverse SHT'), ('k_hankel', 'Hankel'), ('k_finish', 'finish_step'),
                       ('k_sw_', 'shrink-wrap'), ('k_deg2', 'B_l')):
        if key in name:
            return label
    return 'other'


# workgroups of a kernel that fit one CU (dynamic LDS is not in the trace: from the launchers -- k_rproj 113 KB, the inverse SHT
# keeps a shell's spectra (one workgroup per CU), the forward SHT two, the Hankel tiles one or two)
PER_CU = {'k_rproj': 1, 'k_polar': 1, 'k_sht_chain': 1, 'k_sht_inv': 1, 'k_sht_fwd': 2, 'k_hankel': 2, 'k_proj': 4}
ev = []
for r in rows:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    wg = int(r['Workgroup_Size_X']) * int(r['Workgroup_Size_Y']) * int(r['Workgroup_Size_Z'])
    nwg = 1
    for ax in 'XYZ':
        nwg *= max(1, -(-int(r['Grid_Size_' + ax]) // max(int(r['Workgroup_Size_' + ax]), 1)))
    per_cu = min(THREADS_CU // max(wg, 1), 8)
    for key, v in PER_CU.items():
        if key in r['Kernel_Name']:
            per_cu = min(per_cu, v)
    per_cu = max(per_cu, 1)
    cus = min(CUS, (nwg + per_cu - 1) // per_cu)
    ev.append(dict(s=s, e=e, name=r['Kernel_Name'], fam=fam(r['Kernel_Name']), q=r.get('Queue_Id', '?'), nwg=nwg, wg=wg, cus=cus))
ev.sort(key=lambda x: x['s'])
# steady state = the timed region = the last `timed` steps of every engine: from the start of the (timed x engines)-th projection
# launch counted from the end
proj_ev = [x for x in ev if 'projection (k_rproj)' in x['fam'] or x['fam'] == 'polar (complex)']
n_q = len({x['q'] for x in proj_ev})
timed = int(skip) if skip >= 1 else max(1, int(len(proj_ev) / max(n_q, 1) * (1 - skip)))
lo = proj_ev[-timed * n_q]['s'] - 200000 if len(proj_ev) >= timed * n_q else ev[0]['s']
win = [x for x in ev if x['s'] >= lo and x['fam'] != 'other']
span = (win[-1]['e'] - win[0]['s']) / 1e3
n_proj = sum(1 for x in win if x['fam'].startswith('projection (k_rproj)') or x['fam'] == 'polar (complex)')
queues = sorted({x['q'] for x in win})
steps = n_proj / max(len(queues), 1)
print('window: %.1f us, %d kernel launches on %d queues, %.1f steps per engine -> %.1f us per step' % (span, len(win), len(queues), steps, span / max(steps, 1)))
print('\n%-32s %8s %9s %9s %10s %12s %9s' % ('family', 'launches', 'wg/launch', 'CUs held', 'avg us', 'CU x us/step', 'of chip'))
acc = defaultdict(lambda: [0, 0, 0, 0.0, 0.0, 0])
for x in win:
    a = acc[x['fam']]
    a[0] += 1; a[1] += x['nwg']; a[3] += (x['e'] - x['s']) / 1e3; a[4] += x['cus'] * (x['e'] - x['s']) / 1e3; a[5] += x['cus']
tot = 0.0
for k, a in sorted(acc.items(), key=lambda kv: -kv[1][4]):
    per_step = a[4] / max(steps, 1)
    tot += per_step
    print('%-24s %8d %9d %9d %10.1f %12.0f %8.1f%%' % (k, a[0], a[1] // a[0], a[5] // a[0], a[3] / a[0], per_step,
                                                          100 * per_step / (CUS * span / max(steps, 1))))
print('%-24s %39s %12.0f %8.1f%%   (an upper bound: a launch is charged all the CUs it can hold for its whole duration)' % (
    'sum', '', tot, 100 * tot / (CUS * span / max(steps, 1))))
print('\nper engine (queue): kernel time per step by family, us')
for q in queues:
    mine = [x for x in win if x['q'] == q]
    by = defaultdict(float)
    for x in mine:
        by[x['fam']] += (x['e'] - x['s']) / 1e3
    nst = sum(1 for x in mine if 'projection' in x['fam'] or 'polar' in x['fam'])
    busy = sum(by.values())
    print('  queue %s: %s | busy %.0f of %.0f us per step' % (q, ', '.join('%s %.0f' % (k, v / max(nst, 1)) for k, v in sorted(by.items(), key=lambda kv: -kv[1])),
                                                             busy / max(nst, 1), span / max(nst, 1)))
# what runs on the other queues while a projection kernel is in flight
proj = [x for x in win if 'projection (k_rproj)' in x['fam'] or x['fam'] == 'polar (complex)']
other_cu_time = 0.0
proj_time = 0.0
held = 0.0
for p in proj:
    proj_time += (p['e'] - p['s']) / 1e3
    held += p['cus']
    for x in win:
        if x['q'] == p['q'] or x['fam'] == p['fam']:
            continue
        ov = min(p['e'], x['e']) - max(p['s'], x['s'])
        if ov > 0:
            other_cu_time += ov / 1e3 * x['cus']
if proj:
    print('\nprojection kernel: %.0f us per launch holding <= %d CUs; meanwhile the transforms of the other engines hold on average %.0f of the %d CUs'
          % (proj_time / len(proj), held / len(proj), other_cu_time / proj_time, CUS))
