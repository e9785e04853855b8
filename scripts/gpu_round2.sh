#!/bin/bash
# One gpurun call of round 2: [tests] [bench] [rocprofv3 kernel stats] [counter passes].  Outputs -> gpurun_out/$TAG
# usage: scripts/gpu_round2.sh TAG [tests] [bench] [stats] [pmc] [configs]
TAG=${1:-run}; shift
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
want() { for a in "$@"; do [ "$a" = "$W" ] && return 0; done; return 1; }
for W in tests bench stats pmc configs; do
  want "$@" || continue
  case $W in
  tests)
    timeout 900 python -m pytest tests -m gpu -q --maxfail 6 --durations=8 --timeout 180 -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log
    tail -25 $OUT/pytest_gpu.log
    for PP in 20000 100000; do timeout 120 python tests/tools/graph_capture_check.py $PP > $OUT/graph_check_$PP.log 2>&1; echo "graph check P=$PP rc=$?"; grep -v "^  File\|^$" $OUT/graph_check_$PP.log | head -12; done ;;
  bench)
    timeout 900 python bench.py > $OUT/bench.log 2>&1; echo "bench rc=$?"
    tail -1 $OUT/bench.log | cut -c1-3000 ;;
  configs)
    for c in c2 c5shape ref16k c4; do
      timeout 900 python bench.py --config $c --no-cpu-baseline > $OUT/bench_$c.log 2>&1; echo "bench $c rc=$?"
      tail -1 $OUT/bench_$c.log | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['config']['name'], round(j['ms_per_step'],4), 'ms', round(j['value']/1e6,1), 'M/s', j['modes_ms_per_step'], j.get('mode_errors'))"
    done ;;
  stats)
    for m in graph eager-st; do
      timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_$m -o stats -- python bench.py --mode $m --only-mode --steps 100 --warmup 20 --no-cpu-baseline > $OUT/bench_rocprof_$m.log 2>&1
      python scripts/top_kernels.py $OUT/stats_$m
      find $OUT/stats_$m -name "*kernel_trace.csv" -size +8M -delete
    done ;;
  pmc)
    H=$(python -c "import bench; print(bench.lib_hash())")
    i=0
    for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU" "SQ_INSTS_SALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_ACTIVE_INST_VALU" \
               "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_IDX_ACTIVE" \
               "FETCH_SIZE" "WRITE_SIZE"; do
      i=$((i+1))
      timeout 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/pmc_$i -o pmc -- python bench.py --mode eager-st --only-mode --calibrate --steps 10 --warmup 5 --no-cpu-baseline > $OUT/pmc_$i.log 2>&1
      echo "pmc pass $i ($grp) rc=$?"
      find $OUT/pmc_$i -name "*kernel_trace.csv" -size +4M -delete
    done
    python scripts/sq_counters.py $OUT/sq_counters.json $H $OUT/pmc_1 $OUT/pmc_2 $OUT/pmc_3 $OUT/pmc_4 $OUT/pmc_5 $OUT/pmc_6 ;;
  esac
done
