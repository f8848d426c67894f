"""Per-kernel MFMA utilisation from a rocprofv3 --pmc pass with SQ_VALU_MFMA_BUSY_CYCLES, GRBM_GUI_ACTIVE,
SQ_INSTS_VALU_MFMA_MOPS_F64 (gfx950 has no derived-metric section in ROCm 7.2: the MI300 formula is applied by hand:
MfmaUtil = sum(SQ_VALU_MFMA_BUSY_CYCLES) / (max(GRBM_GUI_ACTIVE) x 1024 SIMDs); MfmaFlopsF64 = MOPS_F64 x 512).
usage: pmc_mfma.py <dir>"""
import csv, glob, sys
from collections import defaultdict

f = glob.glob(sys.argv[1] + "/*/*counter_collection.csv")[0]
acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
dur = defaultdict(lambda: [0.0, 0])
seen = set()
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"]
    a = acc[k][r["Counter_Name"]]
    a[0] += float(r["Counter_Value"]); a[1] += 1
    key = (k, r["Dispatch_Id"])
    if key not in seen:
        seen.add(key)
        dur[k][0] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"]); dur[k][1] += 1
print("%-50s %6s %10s %10s %12s %12s" % ("kernel", "calls", "avg_us", "MfmaUtil%", "MFMA_GFLOP", "TFLOP/s"))
for k, c in acc.items():
    if "SQ_VALU_MFMA_BUSY_CYCLES" not in c:
        continue
    busy = c["SQ_VALU_MFMA_BUSY_CYCLES"][0] / c["SQ_VALU_MFMA_BUSY_CYCLES"][1]
    if busy == 0:
        continue
    act = c["GRBM_GUI_ACTIVE"][0] / c["GRBM_GUI_ACTIVE"][1]
    mops = c["SQ_INSTS_VALU_MFMA_MOPS_F64"][0] / c["SQ_INSTS_VALU_MFMA_MOPS_F64"][1] if "SQ_INSTS_VALU_MFMA_MOPS_F64" in c else 0.0
    us = dur[k][0] / dur[k][1] / 1e3
    print("%-50s %6d %10.1f %10.2f %12.3f %12.2f" % (k[:50], dur[k][1], us, 100 * busy / (act * 1024), mops * 512 / 1e9,
                                                     mops * 512 / (us * 1e-6) / 1e12 if us else 0))
