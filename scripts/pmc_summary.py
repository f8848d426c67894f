"""Per-kernel HBM traffic from two rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE, collected separately as the
TCC slots require).  Units: both counters are in KiB; on gfx950 FETCH_SIZE tallies 128-byte read requests at 64 bytes,
so wide coalesced reads are doubled (MI355X_MICROARCH.md, HBM section) -- printed raw and corrected.
usage: pmc_summary.py <fetch_dir> <write_dir> [out.json restarts_per_launch config]"""
import csv
import os, glob, sys
from collections import defaultdict


def per_kernel(d, counter):
    """mean counter value per kernel over its launches with the LARGEST grid (with several engines per GPU the launches of
    the engine that holds fewer restarts are smaller; the bench line quotes engine 0, which holds the most)"""
    f = max(glob.glob(d + "/*/*counter_collection.csv"), key=os.path.getmtime)       # (the newest run when several were merged into d)
    rows = [r for r in csv.DictReader(open(f)) if r["Counter_Name"] == counter]
    gmax = defaultdict(int)
    for r in rows:
        gmax[r["Kernel_Name"]] = max(gmax[r["Kernel_Name"]], int(r["Grid_Size"]))
    acc = defaultdict(lambda: [0.0, 0])
    for r in rows:
        if int(r["Grid_Size"]) != gmax[r["Kernel_Name"]]:
            continue
        a = acc[r["Kernel_Name"]]
        a[0] += float(r["Counter_Value"])
        a[1] += 1
    return acc


fetch = per_kernel(sys.argv[1], "FETCH_SIZE")
write = per_kernel(sys.argv[2], "WRITE_SIZE")
rows = []
for k in fetch:
    fk = fetch[k][0] / fetch[k][1] * 1024
    wk = write[k][0] / write[k][1] * 1024 if k in write else 0.0
    rows.append((2 * fk + wk, k, fetch[k][1], fk, wk))
rows.sort(reverse=True)
print("%-52s %6s %12s %12s %12s %12s" % ("kernel", "calls", "fetch_raw_MB", "fetch_x2_MB", "write_MB", "hbm_MB"))
for tot, k, n, fk, wk in rows[:24]:
    print("%-52s %6d %12.2f %12.2f %12.2f %12.2f" % (k[:52], n, fk / 1e6, 2 * fk / 1e6, wk / 1e6, tot / 1e6))

if len(sys.argv) > 3:
    import json
    fam_of = [("k_sht_fwd_pair", "sht_fwd"), ("k_sht_fwd_reg", "sht_fwd"), ("k_sht_inv_wide<0", "sht_inv"), ("k_sht_inv_wide<1", "sht_inv_modulus"),
              ("k_sht_inv_wide<4", "sht_inv_real"), ("k_hankel", "hankel"), ("k_real_update", "real_update"), ("k_rproj", "polar"),
              ("k_sht_chain<0", "sht_chain"), ("k_sht_chain<1", "sht_chain_modulus"), ("k_sht_chain<4", "sht_chain_real")]
    fam = defaultdict(lambda: [0.0, 0])
    for k in fetch:
        for pat, name in fam_of:
            if pat in k:
                wb = write[k][0] * 1024 if k in write else 0.0
                # per-launch mean of (2 x FETCH_SIZE + WRITE_SIZE), weighting the variants of a family by their calls
                fam[name][0] += 2 * fetch[k][0] * 1024 + wb * fetch[k][1] / max(write[k][1], 1) if k in write else 2 * fetch[k][0] * 1024
                fam[name][1] += fetch[k][1]
    # what one phasing step of one restart physically moves: every kernel of the loop (chained / plain SHTs, Hankel, projection, finish,
    # shrink wrap, copies) summed over the run, divided by the restart-steps the run made (k_finish_step: one 256-thread block per restart)
    def all_rows(d, counter):
        f = max(glob.glob(d + "/*/*counter_collection.csv"), key=os.path.getmtime)
        return [r for r in csv.DictReader(open(f)) if r["Counter_Name"] == counter]
    loop = ("k_sht_", "k_hankel", "k_rproj", "k_finish_step", "k_sw_", "k_proj", "k_polar", "k_real_update", "k_coeff", "k_modulus", "k_deg2", "copyBuffer", "fillBuffer", "k_pack_masks")
    fr, wr_ = all_rows(sys.argv[1], "FETCH_SIZE"), all_rows(sys.argv[2], "WRITE_SIZE")
    tot_f = sum(float(r["Counter_Value"]) for r in fr if any(p in r["Kernel_Name"] for p in loop)) * 1024
    tot_w = sum(float(r["Counter_Value"]) for r in wr_ if any(p in r["Kernel_Name"] for p in loop)) * 1024
    rsteps = sum(int(r["Grid_Size"]) // 256 for r in fr if "k_finish_step" in r["Kernel_Name"])
    per_rs = (2 * tot_f + tot_w) / max(rsteps, 1)
    print("loop kernels: fetch x2 %.1f MB + write %.1f MB over %d restart-steps = %.1f MB per restart-step" % (2 * tot_f / 1e6, tot_w / 1e6, rsteps, per_rs / 1e6))
    out = {"restarts_per_launch": int(sys.argv[4]), "config": int(sys.argv[5]), "hbm_bytes_per_step_per_restart": per_rs,
           "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) of bench.py; KiB units; FETCH_SIZE doubled (gfx950 "
                     "tallies 128-byte read requests at 64 bytes); Infinity-Cache hits are included in FETCH_SIZE",
           "hbm_bytes_per_launch": {k: v[0] / v[1] for k, v in fam.items()}}
    json.dump(out, open(sys.argv[3], "w"), indent=1)
    print(json.dumps(out["hbm_bytes_per_launch"], indent=1))
