#!/bin/bash
# One gpurun call: parity tests, bench line, rocprofv3 kernel stats, PMC traffic passes.  Outputs -> gpurun_out/$TAG
TAG=${1:-run}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log
tail -3 $OUT/pytest_gpu.log
timeout 600 python bench.py > $OUT/bench.log 2>&1; echo "bench rc=$?"
tail -1 $OUT/bench.log | cut -c1-1500
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o stats -- python bench.py --steps 50 --warmup 10 --no-cpu-baseline > $OUT/bench_rocprof.log 2>&1
python scripts/top_kernels.py $OUT/stats
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o pmc -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $OUT/pmc_fetch.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o pmc -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $OUT/pmc_write.log 2>&1
python scripts/pmc_summary.py $OUT/pmc_fetch $OUT/pmc_write
find $OUT -name "*kernel_trace.csv" -size +8M -delete
python scripts/pmc_to_json.py $OUT/pmc_fetch $OUT/pmc_write > $OUT/pmc_traffic.json
