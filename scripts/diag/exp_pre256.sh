export TMPDIR=/tmp; OUT=gpurun_out/pre256; mkdir -p $OUT
for cfg in c3 c5shape; do
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$cfg -o stats -- python bench.py --config $cfg --bin-mode 0 --forward-mode blocking --mode eager-st --only-mode --steps 60 --warmup 10 --no-cpu-baseline > $OUT/$cfg.log 2>&1
echo "$cfg rc=$?"; tail -2 $OUT/$cfg.log | cut -c1-300
python scripts/top_kernels.py $OUT/$cfg | head -14
find $OUT/$cfg -name "*kernel_trace.csv" -delete
done
