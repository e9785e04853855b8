#!/bin/bash
# like ab_lib.sh, whole-step graph timing: scripts/diag/ab_lib_graph.sh variant...   (files .alt/libmgsplat_VARIANT.so; "shipped" = the tree's)
cp manigaussian_amd/libmgsplat.so /tmp/shipped.so
for v in "$@"; do
  if [ "$v" = shipped ]; then cp /tmp/shipped.so manigaussian_amd/libmgsplat.so; else cp .alt/libmgsplat_$v.so manigaussian_amd/libmgsplat.so; fi
  for c in ${AB_CONFIGS:-c3 ref16k}; do
    timeout 200 python bench.py $( if [[ $c == views* ]]; then echo "--config c3 --views ${c#views}"; else echo "--config $c"; fi ) --steps 300 --warmup 50 --mode graph --only-mode --no-cpu-baseline --no-reference-kernels 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v $c graph ms/step %.4f'%d['ms_per_step'], {k: round(v*1e3,1) for k,v in d['stages_ms'].items() if v})"
  done
done
cp /tmp/shipped.so manigaussian_amd/libmgsplat.so
