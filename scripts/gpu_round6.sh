#!/bin/bash
# One gpurun call of round 6.  Outputs -> gpurun_out/$TAG (merged back by gpurun).
# usage: scripts/gpu_round5.sh TAG section [section ...]     (sections run in the order given below)
#   golden     tests/golden/make_golden_ref.py for the cases named in $GOLDEN_ONLY (default: the round-5 cases) -> $OUT/golden_ref
#   quick      pytest -m gpu on the tests named in $QUICK_K (-k expression)
#   tests      pytest -m gpu (+ the measured parity statistics, MGS_PARITY_REPORT)
#   host       scripts/host_profile5.py: host cost of one fwd+bwd, compiled binding vs ctypes shim
#   bench      python bench.py (the driver's default command) + the driver's short form (--steps 20 --warmup 5)
#   configs    bench lines of c2, c5shape, ref16k, views 4/8, c4, c5
#   stats      rocprofv3 --kernel-trace --stats of the default bench (graph and eager-st)
#   cfgstats   rocprofv3 --kernel-trace --stats of c2, c5shape and ref16k (eager-st)
#   pmc        the six counter passes (one group per pass) -> sq_counters.json
#   trace      s_memtime timelines of the render kernels and of the bucket-rank binning
#   extra      $EXTRA_CMD (a shell command), output -> $OUT/extra.log
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
      'events', {k: (round(v, 4) if isinstance(v, float) else v) for k, v in (j.get('ms_per_step_events') or {}).items() if k != 'what'},
      'errors', j.get('mode_errors'), 'roof', round(j['roofline']['frac'], 3), 'check', j['roofline_check'],
      'kernels_us', {k: round(v['avg_launch_ms'] * 1e3, 1) for k, v in bk.items()}, 'fracs', {k: round(v['frac'], 3) for k, v in bk.items()},
      'mlp', (j.get('roofline_mlp') or {}).get('frac'), 'cpu', (j.get('cpu_baseline') or {}).get('value'),
      'ref_kernels', {k: v for k, v in (j.get('reference_kernels_same_gpu') or {}).items() if k in ('ms_per_step', 'this_library_over_reference_kernels', 'note')},
      'binding', j.get('host_binding'))
"; }
for W in golden quick tests host pmc bench configs stats cfgstats trace extra; do
  want "$@" || continue
  case $W in
  golden)
    ONLY=${GOLDEN_ONLY:-ref_sh_f32_negfocal_64x64,ref_sh2_f32_ragged_80x48,ref_sh_f8_48x48,ref_scalemod0p5_f3_64x64,ref_scalemod2_f32_64x64} \
      timeout 600 python tests/golden/make_golden_ref.py $OUT/golden_ref > $OUT/golden_ref.log 2>&1; echo "golden rc=$?"; cat $OUT/golden_ref.log | tail -12 ;;
  quick)
    MGS_PARITY_REPORT=$PWD/$OUT/parity_report_quick.jsonl timeout 900 python -m pytest tests -m gpu -q --maxfail 10 --timeout 400 -p no:cacheprovider -k "$QUICK_K" > $OUT/pytest_quick.log 2>&1; echo "pytest quick rc=$?" >> $OUT/pytest_quick.log
    tail -60 $OUT/pytest_quick.log ;;
  tests)
    rm -f $OUT/parity_report.jsonl
    MGS_PARITY_REPORT=$PWD/$OUT/parity_report.jsonl timeout 1800 python -m pytest tests -m gpu -q --maxfail 12 --durations=12 --timeout 400 -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log
    tail -60 $OUT/pytest_gpu.log ;;
  host)
    timeout 600 python scripts/host_profile5.py > $OUT/host_profile.log 2>&1; echo "host rc=$?"; grep -v "^$" $OUT/host_profile.log | head -60 ;;
  bench)
    timeout 900 python bench.py --strict-roofline > $OUT/bench.log 2>&1; echo "bench rc=$?"
    tail -1 $OUT/bench.log > $OUT/bench.json; line < $OUT/bench.json
    timeout 600 python bench.py --steps 20 --warmup 5 > $OUT/bench_driver_form.log 2>&1; echo "bench (driver form) rc=$?"
    tail -1 $OUT/bench_driver_form.log > $OUT/bench_driver_form.json; line < $OUT/bench_driver_form.json ;;
  configs)
    IFS=';' read -ra CFGS <<< "${CONFIGS:---config c2;--config c5shape;--config ref16k;--views 4;--views 8;--config c4 --steps 30 --warmup 10;--config c5 --steps 20 --warmup 8}"
    for c in "${CFGS[@]}"; do
      n=$(echo $c | tr -d ' -' ); n=${n#config}
      case "$c" in *"config c4"*|*"config c5 "*) cpu="" ;; *) cpu="--no-cpu-baseline" ;; esac
      timeout 1200 python bench.py $c $cpu --strict-roofline > $OUT/bench_$n.log 2>&1; echo "bench $c rc=$?"
      tail -1 $OUT/bench_$n.log > $OUT/bench_$n.json; line < $OUT/bench_$n.json
    done ;;
  stats)
    for m in ${STATS_MODES:-graph eager-st}; do
      timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_$m -o stats -- python bench.py --mode $m --only-mode --steps 100 --warmup 20 --no-cpu-baseline > $OUT/bench_rocprof_$m.log 2>&1
      python scripts/top_kernels.py $OUT/stats_$m
      find $OUT/stats_$m -name "*kernel_trace.csv" -delete
    done ;;
  cfgstats)
    for c in ${STATS_CONFIGS:-c2 c5shape ref16k}; do
      timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_$c -o stats -- python bench.py --config $c --mode eager-st --only-mode --steps 100 --warmup 20 --no-cpu-baseline --no-reference-kernels > $OUT/bench_rocprof_$c.log 2>&1
      python scripts/top_kernels.py $OUT/stats_$c
      find $OUT/stats_$c -name "*kernel_trace.csv" -delete
    done ;;
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
      cp $OUT/sq_counters$sfx.json profiles/r06_sq_counters$sfx.json
    done ;;
  trace)
    timeout 200 python scripts/trace_fwd.py > $OUT/trace_fwd.log 2>&1; echo "trace rc=$?"; tail -22 $OUT/trace_fwd.log
    timeout 200 python scripts/trace_bwd.py > $OUT/trace_bwd.log 2>&1; echo "trace bwd rc=$?"; tail -22 $OUT/trace_bwd.log
    (timeout 200 python scripts/trace_bin.py; timeout 200 python scripts/trace_bin.py 16384 128) > $OUT/trace_bin.log 2>&1; echo "trace bin rc=$?"; tail -34 $OUT/trace_bin.log ;;
  extra)
    timeout ${EXTRA_TIMEOUT:-600} bash -c "$EXTRA_CMD" > $OUT/extra.log 2>&1; echo "extra rc=$?"; tail -40 $OUT/extra.log ;;
  esac
done
