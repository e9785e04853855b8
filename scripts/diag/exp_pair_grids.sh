#!/bin/bash
# Experiment (round 5): the two-pixel render backward on grids with more blocks than CUs (the forward ran 8-wave workgroups).
# gm_waves 12 = shipped (two 8-wave one-pixel workgroups per CU), 6 = two 6-wave two-pixel workgroups per CU, 7 = one 12-wave
# two-pixel workgroup per CU.
for gw in 12 6 7; do
  for c in "--views 8" "--views 4" "--config c5shape" "--config ref16k"; do
    echo "== gm_waves $gw $c"
    timeout 600 python bench.py $c --gm-waves $gw --mode eager-st --only-mode --no-cpu-baseline --no-reference-kernels --steps 200 --warmup 30 2>&1 | tail -1 | python -c "
import sys, json
j = json.loads(sys.stdin.read()); bk = j.get('roofline_by_kernel') or {}
print(round(j['ms_per_step'], 4), 'ms/step', {k: round(v['avg_launch_ms'] * 1e3, 1) for k, v in bk.items() if 'render' in k})"
  done
done
