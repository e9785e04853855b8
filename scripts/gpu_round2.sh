#!/bin/bash
# One gpurun call of round 2: [tests] [bench] [rocprofv3 kernel stats] [counter passes].  Outputs -> gpurun_out/$TAG
# usage: scripts/gpu_round2.sh TAG [tests] [bench] [stats] [pmc] [configs]
TAG=${1:-run}; shift
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
want() { for a in "$@"; do [ "$a" = "$W" ] && return 0; done; return 1; }
for W in tests graph trace stats5 multi variants skips bisect noscratch scratch dump oob host bench stats pmc configs; do
  want "$@" || continue
  case $W in
  tests)
    timeout 900 python -m pytest tests -m gpu -q --maxfail 6 --durations=8 --timeout 180 -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log
    tail -25 $OUT/pytest_gpu.log
    for PP in 20000 100000; do timeout 120 python tests/tools/graph_capture_check.py $PP > $OUT/graph_check_$PP.log 2>&1; echo "graph check P=$PP rc=$?"; grep -v "^  File\|^$" $OUT/graph_check_$PP.log | head -12; done ;;
  graph)
    for PP in 20000; do timeout 120 python tests/tools/graph_capture_check.py $PP > $OUT/graph_check_$PP.log 2>&1; echo "graph check P=$PP rc=$?"; grep -v "^  File\|^$" $OUT/graph_check_$PP.log | head -12; done ;;
  trace)
    timeout 200 python scripts/trace_fwd.py > $OUT/trace_fwd.log 2>&1; echo "trace rc=$?"; cat $OUT/trace_fwd.log | tail -22
    timeout 200 python scripts/trace_bwd.py > $OUT/trace_bwd.log 2>&1; echo "trace bwd rc=$?"; cat $OUT/trace_bwd.log | tail -22 ;;
  variants)
    # experimental builds manigaussian_amd/libmgsplat_<tag>.so swapped in one at a time: kernel stats of a short bench
    cp manigaussian_amd/libmgsplat.so /tmp/libmgsplat_keep.so
    for so in manigaussian_amd/libmgsplat_*.so; do
      tag=$(basename $so .so); tag=${tag#libmgsplat_}
      cp $so manigaussian_amd/libmgsplat.so
      timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/var_$tag -o stats -- python bench.py --mode eager-st --only-mode --steps 200 --warmup 30 --no-cpu-baseline > $OUT/var_$tag.log 2>&1
      echo "variant $tag: $(tail -1 $OUT/var_$tag.log | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(round(j['ms_per_step'],4))")"
      python scripts/top_kernels.py $OUT/var_$tag | head -10
      find $OUT/var_$tag -name "*kernel_trace.csv" -delete
    done
    cp /tmp/libmgsplat_keep.so manigaussian_amd/libmgsplat.so ;;
  skips)
    for sk in status images grads status,images,grads; do
      MGS_GRAPH_SKIP=$sk timeout 100 python tests/tools/graph_capture_check.py 20000 > $OUT/skip_$sk.log 2>&1; echo "skip $sk rc=$? $(grep -c GRAPH_OK $OUT/skip_$sk.log) $(grep -c fault $OUT/skip_$sk.log) last: $(grep stage $OUT/skip_$sk.log | tail -1)"
    done ;;
  bisect)
    for part in full; do for b in verify leafmove; do
      timeout 100 python tests/tools/graph_bisect.py $part $b > $OUT/bisect_${part}_$b.log 2>&1; echo "bisect $part $b rc=$? $(grep -c BISECT_OK $OUT/bisect_${part}_$b.log)"; grep "replay \|eager " $OUT/bisect_${part}_$b.log
    done; done ;;
  noscratch)
    # library built with -DMGS_FWD_WAVES8 in manigaussian_amd/libmgsplat_w8.so is swapped in for this check only
    cp manigaussian_amd/libmgsplat.so /tmp/libmgsplat_keep.so; cp manigaussian_amd/libmgsplat_w8.so manigaussian_amd/libmgsplat.so
    MGS_GM_WAVES=8 timeout 120 python tests/tools/graph_capture_check.py 20000 > $OUT/graph_noscratch.log 2>&1; echo "graph no-scratch kernels rc=$?"; grep "stage\|fault\|GRAPH_OK\|Error" $OUT/graph_noscratch.log | tail -5
    cp /tmp/libmgsplat_keep.so manigaussian_amd/libmgsplat.so ;;
  scratch)
    HSA_NO_SCRATCH_RECLAIM=1 MGS_GRAPH_MOVE=0.0 timeout 120 python tests/tools/graph_capture_check.py 20000 > $OUT/graph_noreclaim.log 2>&1; echo "graph HSA_NO_SCRATCH_RECLAIM=1 rc=$?"; grep "stage\|fault\|GRAPH_OK\|Error" $OUT/graph_noreclaim.log | tail -5
    HSA_NO_SCRATCH_RECLAIM=1 timeout 120 python tests/tools/graph_capture_check.py 20000 > $OUT/graph_noreclaim2.log 2>&1; echo "graph HSA_NO_SCRATCH_RECLAIM=1 moved rc=$?"; grep "stage\|fault\|GRAPH_OK\|Error" $OUT/graph_noreclaim2.log | tail -5 ;;
  dump)
    MGS_GRAPH_DUMP=1 MGS_GRAPH_MOVE=0.0 timeout 120 python tests/tools/graph_capture_check.py 20000 > $OUT/graph_dump.log 2>&1; echo "graph dump rc=$?"; grep -v "^  File" $OUT/graph_dump.log | cut -c1-600 | tail -60 ;;
  oob)
    # every tensor its own hipMalloc: an out-of-bounds write faults instead of landing in the caching allocator's slack
    PYTORCH_NO_CUDA_MEMORY_CACHING=1 timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x --timeout 300 -p no:cacheprovider -k "parity_with_oracle or translucent or async or view_batch_matches_oracle_b or full_size" > $OUT/pytest_oob.log 2>&1; echo "oob pytest rc=$?"; tail -5 $OUT/pytest_oob.log
    MGS_GRAPH_MOVE=0.0 timeout 120 python tests/tools/graph_capture_check.py 20000 > $OUT/graph_move0.log 2>&1; echo "graph move 0 rc=$?"; grep "stage\|fault\|GRAPH_OK" $OUT/graph_move0.log | tail -4
    MGS_GRAPH_EAGER_FIRST=1 timeout 120 python tests/tools/graph_capture_check.py 20000 > $OUT/graph_eager_first.log 2>&1; echo "graph eager-first rc=$?"; grep "stage\|fault\|GRAPH_OK" $OUT/graph_eager_first.log | tail -4 ;;
  host)
    timeout 300 python scripts/host_profile3.py > $OUT/host_profile.log 2>&1; echo "host rc=$?"; head -60 $OUT/host_profile.log ;;
  bench)
    timeout 900 python bench.py > $OUT/bench.log 2>&1; echo "bench rc=$?"
    tail -1 $OUT/bench.log | cut -c1-3000 ;;
  stats5)
    timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_c5 -o stats -- python bench.py --config c5shape --only-mode --steps 60 --warmup 10 --no-cpu-baseline > $OUT/bench_rocprof_c5.log 2>&1
    python scripts/top_kernels.py $OUT/stats_c5; find $OUT/stats_c5 -name "*kernel_trace.csv" -delete
    timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_v8 -o stats -- python bench.py --views 8 --only-mode --steps 60 --warmup 10 --no-cpu-baseline > $OUT/bench_rocprof_v8.log 2>&1
    python scripts/top_kernels.py $OUT/stats_v8; find $OUT/stats_v8 -name "*kernel_trace.csv" -delete ;;
  multi)
    for c in "--config c5shape" "--views 8" "--views 4" "--config c2 --size 256"; do
      timeout 600 python bench.py $c --no-cpu-baseline --only-mode > $OUT/bench_multi.log 2>&1; echo "bench $c rc=$?"
      tail -1 $OUT/bench_multi.log | python -c "import sys,json; j=json.loads(sys.stdin.read()); c=j['config']; print(c['name'], c['P'], c['W'], 'V', c['views_per_gpu'], round(j['ms_per_step'],4), 'ms/step', round(j['ms_per_step']/c['renders_per_step_per_gpu'],4), 'ms/render', round(j['value']/1e6,1), 'M/s', {k: round(v,4) for k,v in j['stages_ms'].items() if v})"
    done ;;
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
      timeout 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/pmc_$i -o pmc -- python bench.py $PMC_BENCH_ARGS --mode eager-st --only-mode --calibrate --steps 10 --warmup 5 --no-cpu-baseline > $OUT/pmc_$i.log 2>&1
      echo "pmc pass $i ($grp) rc=$?"
      find $OUT/pmc_$i -name "*kernel_trace.csv" -size +4M -delete
    done
    python scripts/sq_counters.py $OUT/sq_counters.json $H $OUT/pmc_1 $OUT/pmc_2 $OUT/pmc_3 $OUT/pmc_4 $OUT/pmc_5 $OUT/pmc_6 ;;
  esac
done
