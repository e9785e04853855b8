#!/bin/bash
# One gpurun call of round 4.  Outputs -> gpurun_out/$TAG (merged back by gpurun).
# usage: scripts/gpu_round4.sh TAG section [section ...]
#   tests      pytest -m gpu (+ the measured parity statistics, MGS_PARITY_REPORT)
#   quick      pytest -m gpu on the tests named in $QUICK_K (-k expression)
#   bench      python bench.py (the driver's default command) + the driver's short form (--steps 20 --warmup 5)
#   configs    bench lines of c2, c5shape, ref16k, views 4/8, c4, c5
#   exp        what fast_exp = 0 costs: kernel stats of the default bench with --fast-exp 0 and 1 (c3 and c5shape)
#   host       scripts/host_profile3.py: Python cost of one fwd+bwd through the public API
#   stats      rocprofv3 --kernel-trace --stats of the default bench (graph and eager-st)
#   stats_dyn  ... of c4, c5, c5shape and 8 views
#   pmc        the six counter passes (one group per pass, never combined with other trace domains) -> sq_counters.json
#   variants   libmgsplat_<tag>.so builds swapped in one at a time: kernel stats of a short bench each (+ optional tests)
#   trace      s_memtime timelines of the render kernels
TAG=${1:-run}; shift
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
want() { for a in "$@"; do [ "$a" = "$W" ] && return 0; done; return 1; }
line() { python -c "
import sys, json
j = json.loads(sys.stdin.read())
c = j['config']
lr = j.get('long_run') or {}
bk = j.get('roofline_by_kernel') or {}
print(c['name'], c['P'], f\"{c['W']}x{c['H']}\", 'renders/gpu', c['renders_per_step_per_gpu'], 'mode', j['mode'], round(j['ms_per_step'], 4), 'ms/step',
      round(j['value'] / 1e6, 1), 'M/s', 'long_run', round(lr.get('ms_per_step', 0), 4), 'modes', {k: round(v, 4) for k, v in j['modes_ms_per_step'].items()},
      'errors', j.get('mode_errors'), 'roof', round(j['roofline']['frac'], 3), 'check', j['roofline_check'],
      'kernels_us', {k: round(v['avg_launch_ms'] * 1e3, 1) for k, v in bk.items()}, 'fracs', {k: round(v['frac'], 3) for k, v in bk.items()},
      'mlp', (j.get('roofline_mlp') or {}).get('frac'), 'cpu', (j.get('cpu_baseline') or {}).get('value'))
"; }
for W in tests quick exp host pmc bench configs stats stats_dyn variants trace; do
  want "$@" || continue
  case $W in
  tests)
    rm -f $OUT/parity_report.jsonl
    MGS_PARITY_REPORT=$PWD/$OUT/parity_report.jsonl timeout 1500 python -m pytest tests -m gpu -q --maxfail 8 --durations=12 --timeout 400 -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log
    tail -40 $OUT/pytest_gpu.log ;;
  quick)
    MGS_PARITY_REPORT=$PWD/$OUT/parity_report_quick.jsonl timeout 900 python -m pytest tests -m gpu -q --maxfail 6 --timeout 400 -p no:cacheprovider -k "$QUICK_K" > $OUT/pytest_quick.log 2>&1; echo "pytest quick rc=$?" >> $OUT/pytest_quick.log
    tail -40 $OUT/pytest_quick.log ;;
  exp)
    for cfg in c3 c5shape; do for fe in 1 0; do
      timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/exp_${cfg}_fe$fe -o stats -- python bench.py --config $cfg --fast-exp $fe --mode eager-st --only-mode --steps 100 --warmup 20 --no-cpu-baseline > $OUT/exp_${cfg}_fe$fe.log 2>&1
      echo "fast_exp=$fe $cfg: $(tail -1 $OUT/exp_${cfg}_fe$fe.log | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(round(j['ms_per_step'],4), (j.get('long_run') or {}).get('ms_per_step'))")"
      python scripts/top_kernels.py $OUT/exp_${cfg}_fe$fe | head -9
      find $OUT/exp_${cfg}_fe$fe -name "*kernel_trace.csv" -delete
    done; done ;;
  host)
    timeout 300 python scripts/host_profile3.py > $OUT/host_profile.log 2>&1; echo "host rc=$?"; head -40 $OUT/host_profile.log ;;
  bench)
    timeout 900 python bench.py --strict-roofline > $OUT/bench.log 2>&1; echo "bench rc=$?"
    tail -1 $OUT/bench.log > $OUT/bench.json; line < $OUT/bench.json
    timeout 600 python bench.py --steps 20 --warmup 5 > $OUT/bench_driver_form.log 2>&1; echo "bench (driver form) rc=$?"
    tail -1 $OUT/bench_driver_form.log > $OUT/bench_driver_form.json; line < $OUT/bench_driver_form.json ;;
  configs)
    for c in "--config c2" "--config c5shape" "--config ref16k" "--views 4" "--views 8" "--config c4 --steps 30 --warmup 10" "--config c5 --steps 20 --warmup 8"; do
      n=$(echo $c | tr -d ' -' ); n=${n#config}
      case "$c" in *"config c4"*|*"config c5 "*) cpu="" ;; *) cpu="--no-cpu-baseline" ;; esac
      timeout 1200 python bench.py $c $cpu --strict-roofline > $OUT/bench_$n.log 2>&1; echo "bench $c rc=$?"
      tail -1 $OUT/bench_$n.log > $OUT/bench_$n.json; line < $OUT/bench_$n.json
    done ;;
  stats)
    for m in graph eager-st; do
      timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_$m -o stats -- python bench.py --mode $m --only-mode --steps 100 --warmup 20 --no-cpu-baseline > $OUT/bench_rocprof_$m.log 2>&1
      python scripts/top_kernels.py $OUT/stats_$m
      find $OUT/stats_$m -name "*kernel_trace.csv" -delete
    done ;;
  stats_dyn)
    for c in c4 c5 c5shape; do
      MGS_NO_GEMM_TUNING=1 timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_$c -o stats -- python bench.py --config $c --mode eager-st --only-mode --steps 12 --warmup 4 --no-cpu-baseline > $OUT/bench_rocprof_$c.log 2>&1
      python scripts/top_kernels.py $OUT/stats_$c | head -24
      find $OUT/stats_$c -name "*kernel_trace.csv" -delete
    done
    timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_v8 -o stats -- python bench.py --views 8 --mode eager-st --only-mode --steps 60 --warmup 10 --no-cpu-baseline > $OUT/bench_rocprof_v8.log 2>&1
    python scripts/top_kernels.py $OUT/stats_v8 | head -12; find $OUT/stats_v8 -name "*kernel_trace.csv" -delete ;;
  pmc)
    H=$(python -c "from manigaussian_amd import _lib; print(_lib.build_id())")
    for cfg in ${PMC_CONFIGS:-c3}; do
      i=0
      for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU" "SQ_INSTS_SALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_ACTIVE_INST_VALU" \
                 "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_IDX_ACTIVE" \
                 "FETCH_SIZE" "WRITE_SIZE"; do
        i=$((i+1))
        timeout 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/pmc_${cfg}_$i -o pmc -- python bench.py --config $cfg --mode eager-st --only-mode --calibrate --steps 10 --warmup 5 --no-cpu-baseline > $OUT/pmc_${cfg}_$i.log 2>&1
        echo "pmc $cfg pass $i ($grp) rc=$?"
        find $OUT/pmc_${cfg}_$i -name "*kernel_trace.csv" -size +4M -delete
      done
      sfx=""; [ "$cfg" != "c3" ] && sfx="_$cfg"
      python scripts/sq_counters.py $OUT/sq_counters$sfx.json $H $OUT/pmc_${cfg}_1 $OUT/pmc_${cfg}_2 $OUT/pmc_${cfg}_3 $OUT/pmc_${cfg}_4 $OUT/pmc_${cfg}_5 $OUT/pmc_${cfg}_6 | tail -3
      # a bench section that follows in the same call prints these counters as `traffic` (same binary: bench.py checks the build id)
      cp $OUT/sq_counters$sfx.json profiles/r04_sq_counters$sfx.json
    done ;;
  variants)
    cp manigaussian_amd/libmgsplat.so /tmp/libmgsplat_keep.so
    for so in manigaussian_amd/variants/libmgsplat_*.so; do
      tag=$(basename $so .so); tag=${tag#libmgsplat_}
      cp $so manigaussian_amd/libmgsplat.so
      for cfg in $VARIANT_CONFIGS; do
        timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/var_${tag}_$cfg -o stats -- python bench.py --config $cfg --mode eager-st --only-mode --steps 100 --warmup 20 --no-cpu-baseline > $OUT/var_${tag}_$cfg.log 2>&1
        echo "variant $tag $cfg: $(tail -1 $OUT/var_${tag}_$cfg.log | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(round(j['ms_per_step'],4), (j.get('long_run') or {}).get('ms_per_step'))")"
        python scripts/top_kernels.py $OUT/var_${tag}_$cfg | head -10
        find $OUT/var_${tag}_$cfg -name "*kernel_trace.csv" -delete
      done
      if [ -n "$VARIANT_TEST_K" ]; then
        timeout 600 python -m pytest tests -m gpu -q -x --timeout 300 -p no:cacheprovider -k "$VARIANT_TEST_K" > $OUT/var_${tag}_pytest.log 2>&1; echo "variant $tag pytest rc=$? $(tail -1 $OUT/var_${tag}_pytest.log)"
      fi
    done
    cp /tmp/libmgsplat_keep.so manigaussian_amd/libmgsplat.so ;;
  trace)
    timeout 200 python scripts/trace_fwd.py > $OUT/trace_fwd.log 2>&1; echo "trace rc=$?"; tail -22 $OUT/trace_fwd.log
    timeout 200 python scripts/trace_bwd.py > $OUT/trace_bwd.log 2>&1; echo "trace bwd rc=$?"; tail -22 $OUT/trace_bwd.log ;;
  esac
done
