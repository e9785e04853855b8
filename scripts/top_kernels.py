import csv, glob, sys
for d in sys.argv[1:]:
    f = glob.glob(d + "/**/*kernel_stats.csv", recursive=True)
    if not f:
        print(d, "no stats"); continue
    rows = list(csv.DictReader(open(f[0])))
    print("==", d)
    for r in rows[:12]:
        n = r["Name"]
        n = n.replace("void ", "").replace("mgs::", "")
        print(f"  {n[:70]:70s} calls={r['Calls']:>5} avg_us={float(r['AverageNs'])/1e3:9.1f} tot_ms={float(r['TotalDurationNs'])/1e6:8.2f}")
