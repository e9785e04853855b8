import csv, glob, sys, collections
for d in sys.argv[1:]:
    fs = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
    if not fs:
        print(d, "no counter csv"); continue
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for row in csv.DictReader(open(fs[0])):
        k = row["Kernel_Name"].replace("void ", "").replace("mgs::", "")[:40]
        agg[k][row["Counter_Name"]] += float(row["Counter_Value"]); cnt[(k, row["Counter_Name"])] += 1
    print("==", d)
    for k, c in agg.items():
        if not any(s in k for s in ("chunk_", "coop_", "render_", "preprocess", "ranges", "duplicate", "bin_", "sort", "gm_")):
            continue
        print("  ", k, {n: round(v / cnt[(k, n)]) for n, v in c.items()})
