# Timing-only experiment: what do the forward preprocess's returning reservation atomics cost?  (preprocess alone, nothing
# downstream reads the variants' garbage reservations)   noret: plain atomics, reservations 0;  noatom: no atomics at all
export TMPDIR=/tmp; OUT=gpurun_out/preatom; mkdir -p $OUT
cp manigaussian_amd/libmgsplat.so /tmp/libmgsplat_keep.so
for v in ${VARIANTS:-shipped noret noatom}; do
  [ $v = shipped ] && cp /tmp/libmgsplat_keep.so manigaussian_amd/libmgsplat.so || cp manigaussian_amd/variants/libmgsplat_$v.so manigaussian_amd/libmgsplat.so
  timeout 150 rocprofv3 --kernel-trace --output-format csv -d $OUT/$v -o t -- python scripts/diag/pre_only.py 40 > $OUT/$v.log 2>&1
  echo "== $v rc=$? $(grep 'P=' $OUT/$v.log | tr '\n' ' ')"
  python - <<PY
import csv, glob, collections
f = glob.glob("$OUT/$v/**/*kernel_trace.csv", recursive=True)[0]
d = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if "preprocess_fwd" in r["Kernel_Name"]:
        d[int(r["Grid_Size_X"]) if "Grid_Size_X" in r else int(r["Grid_Size"])].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for g, v in sorted(d.items()):
    v = v[5:]
    print("   grid", g, "launches", len(v), "avg_us", round(sum(v) / len(v) / 1e3, 2), "min_us", round(min(v) / 1e3, 2))
PY
  find $OUT/$v -name "*kernel_trace.csv" -delete
done
cp /tmp/libmgsplat_keep.so manigaussian_amd/libmgsplat.so
