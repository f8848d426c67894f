"""Where the waves of a kernel spend their cycles, from a rocprofv3 --pmc pass with SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY
SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_BUSY_CYCLES (8 SQ slots, /opt/skills/guides/MI355X_MICROARCH.md:
WAIT_ANY = parked at s_waitcnt / barrier, WAIT_INST_ANY = ready but not issued, ACTIVE_INST_ANY = issuing; the three are disjoint and add up
to WAVE_CYCLES).   usage: pmc_issue.py <dir> [kernel substring]"""
import csv, glob, os, sys
from collections import defaultdict

files = sorted(glob.glob(sys.argv[1] + "/*/*counter_collection.csv") + glob.glob(sys.argv[1] + "/*counter_collection.csv"), key=os.path.getmtime)
pat = sys.argv[2] if len(sys.argv) > 2 else ''
acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
dur = defaultdict(lambda: [0.0, 0])
seen = set()
for r in csv.DictReader(open(files[-1])):
    k = r["Kernel_Name"]
    if pat not in k:
        continue
    a = acc[k][r["Counter_Name"]]
    a[0] += float(r["Counter_Value"]); a[1] += 1
    key = (k, r["Dispatch_Id"])
    if key not in seen:
        seen.add(key)
        dur[k][0] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"]); dur[k][1] += 1
for k, c in acc.items():
    m = {n: v[0] / v[1] for n, v in c.items()}
    wc = m.get('SQ_WAVE_CYCLES', 0.0)
    if not wc:
        continue
    print(k[:110])
    print('   launches %d, avg %.1f us under the counters' % (dur[k][1], dur[k][0] / dur[k][1] / 1e3))
    for n in ('SQ_ACTIVE_INST_ANY', 'SQ_WAIT_INST_ANY', 'SQ_WAIT_ANY'):
        if n in m:
            print('   %-20s %6.1f %% of the wave cycles' % (n, 100 * m[n] / wc))
    if 'SQ_ACTIVE_INST_VALU' in m:
        print('   %-20s %6.1f %% of the wave cycles' % ('SQ_ACTIVE_INST_VALU', 100 * m['SQ_ACTIVE_INST_VALU'] / wc))
    if 'SQ_INSTS_VALU' in m and 'SQ_INSTS_LDS' in m:
        print('   instructions per launch: VALU %.3g, LDS %.3g (per wave-level issue)' % (m['SQ_INSTS_VALU'], m['SQ_INSTS_LDS']))
    if 'SQ_BUSY_CYCLES' in m:
        print('   SQ_BUSY_CYCLES %.3g, SQ_WAVE_CYCLES %.3g (quad-cycles)' % (m['SQ_BUSY_CYCLES'], wc))
