"""Summarise a rocprofv3 --kernel-trace csv: per-queue busy time, overlap, and a text timeline of a window.
usage: kernel_timeline.py <dir> [t0_ms t1_ms]"""
import csv, glob, sys
from collections import defaultdict

f = glob.glob(sys.argv[1] + "/*/*kernel_trace.csv")[0]
rows = list(csv.DictReader(open(f)))
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:36], r.get("Queue_Id", "?"), r.get("Stream_Id", "?")) for r in rows]
ev.sort()
t0 = ev[0][0]
span = ev[-1][1] - t0
busy = defaultdict(int)
for s, e, n, q, st in ev:
    busy[q] += e - s
# union of busy intervals
union = 0
cur_s, cur_e = ev[0][0], ev[0][1]
for s, e, *_ in ev[1:]:
    if s > cur_e:
        union += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
union += cur_e - cur_s
print("kernels", len(ev), "span_ms %.3f" % (span / 1e6), "union_busy_ms %.3f" % (union / 1e6),
      "sum_busy_ms %.3f" % (sum(busy.values()) / 1e6))
for q in sorted(busy):
    print(" queue", q, "busy_ms %.3f" % (busy[q] / 1e6))
if len(sys.argv) > 3:
    a, b = float(sys.argv[2]) * 1e6, float(sys.argv[3]) * 1e6
    for s, e, n, q, st in ev:
        if a <= s - t0 <= b:
            print("%10.1f %8.1f  q%-3s s%-3s %s" % ((s - t0) / 1e3, (e - s) / 1e3, q, st, n))
