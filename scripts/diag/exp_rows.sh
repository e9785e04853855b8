# kernel stats of the default bench (eager-st, async) at c3, c5shape and 8 views + the graph-mode headline
export TMPDIR=/tmp; OUT=gpurun_out/${1:-rows}; mkdir -p $OUT
for cfg in "--config c3" "--config c5shape" "--views 8" "--config c2"; do
n=$(echo $cfg | tr -d ' -'); n=${n#config}
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$n -o stats -- python bench.py $cfg --mode eager-st --only-mode --steps 100 --warmup 20 --no-cpu-baseline > $OUT/$n.log 2>&1
echo "$n rc=$? $(tail -1 $OUT/$n.log | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(round(j['ms_per_step'],4), (j.get('long_run') or {}).get('ms_per_step'))")"
python scripts/top_kernels.py $OUT/$n | head -11
find $OUT/$n -name "*kernel_trace.csv" -delete
done
timeout 300 python bench.py --no-cpu-baseline > $OUT/bench.log 2>&1; tail -1 $OUT/bench.log | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('graph', round(j['ms_per_step'],4), j['modes_ms_per_step'], (j.get('long_run') or {}).get('ms_per_step'))"
