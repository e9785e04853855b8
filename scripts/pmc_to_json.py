"""rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE counter CSVs -> per-kernel average HBM traffic per launch (JSON).

  python scripts/pmc_to_json.py <dir with FETCH_SIZE pass> <dir with WRITE_SIZE pass> > profiles/pmc_traffic.json

Units and corrections as /opt/skills/guides/MI355X_MICROARCH.md prescribes: both counters are in KiB; on gfx950
FETCH_SIZE reports half the bytes of wide coalesced reads, so it is doubled; WRITE_SIZE is taken as is."""
import collections, csv, glob, json, sys


def per_kernel(d, counter):
    fs = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
    agg, cnt = collections.defaultdict(float), collections.Counter()
    for row in csv.DictReader(open(fs[0])):
        if row["Counter_Name"] != counter:
            continue
        k = row["Kernel_Name"].replace("void ", "").replace("mgs::", "").split("(")[0]
        agg[k] += float(row["Counter_Value"])
        cnt[k] += 1
    return {k: agg[k] / cnt[k] for k in agg}


fetch, write = per_kernel(sys.argv[1], "FETCH_SIZE"), per_kernel(sys.argv[2], "WRITE_SIZE")
out = {}
for k in sorted(set(fetch) | set(write)):
    if not any(s in k for s in ("preprocess", "bin_", "coop_", "gm_", "render_", "deform")):
        continue
    f_kb, w_kb = fetch.get(k, 0.0), write.get(k, 0.0)
    out[k] = {"FETCH_SIZE_KiB": round(f_kb, 1), "WRITE_SIZE_KiB": round(w_kb, 1),
              "hbm_bytes_per_launch": int((2.0 * f_kb + w_kb) * 1024)}
json.dump({"source": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes) over bench.py --steps 10 --warmup 3",
           "correction": "bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024  (gfx950: FETCH_SIZE counts 64 B per 128-B request)",
           "kernels": out}, sys.stdout, indent=1)
