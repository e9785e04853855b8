#!/bin/bash
# quick per-kernel timing of one or more bench configurations (eager, single-threaded autograd): usage quick_bench.sh "--config c3" "--views 8" ...
[ $# -eq 0 ] && set -- "--config c3"
for c in "$@"; do
  echo "== $c"
  timeout 600 python bench.py $c --mode eager-st --only-mode --no-cpu-baseline --no-reference-kernels --steps 300 --warmup 50 2>&1 | tail -1 | python -c "
import sys, json
j = json.loads(sys.stdin.read()); bk = j.get('roofline_by_kernel') or {}
print(round(j['ms_per_step'], 4), 'ms/step', {k: round(v['avg_launch_ms'] * 1e3, 1) for k, v in bk.items()})"
done
