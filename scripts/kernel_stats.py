import glob,csv,sys
for f in glob.glob(sys.argv[1]+"/*/*kernel_stats.csv"):
    rows=list(csv.DictReader(open(f)))
    for r in rows[:18]:
        print(r["Name"][:50].ljust(50), r["Calls"].rjust(5), ("%.1f"%(float(r["AverageNs"])/1e3)).rjust(9), r["Percentage"].rjust(7))
