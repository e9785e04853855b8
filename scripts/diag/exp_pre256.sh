# Round-4 experiment (run at commit 7916027, when bin_mode 0 still was the rocPRIM binning): kernel stats of the 256-thread
# forward preprocess WITHOUT a histogram -- the floor of what a preprocess on all CUs can reach (DESIGN.md section 10,
# profiles/r04_exp_preprocess_rows.log).  Today bin_mode 0 selects the binning with its tables in memory, whose preprocess
# adds one atomic per instance to the same kernel.
export TMPDIR=/tmp; OUT=gpurun_out/pre256; mkdir -p $OUT
for cfg in c3 c5shape; do
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$cfg -o stats -- python bench.py --config $cfg --bin-mode 0 --forward-mode blocking --mode eager-st --only-mode --steps 60 --warmup 10 --no-cpu-baseline > $OUT/$cfg.log 2>&1
echo "$cfg rc=$?"; tail -2 $OUT/$cfg.log | cut -c1-300
python scripts/top_kernels.py $OUT/$cfg | head -14
find $OUT/$cfg -name "*kernel_trace.csv" -delete
done
