#!/bin/bash
# A/B of alternative builds of libmgsplat.so (files under .alt/, built by hand with other -D flags) ON THE GPU BOX's copy of the tree:
# scripts/diag/ab_lib.sh OUT variant...   ("shipped" = the tree's own library)
out=gpurun_out/$1; shift; mkdir -p $out
cp manigaussian_amd/libmgsplat.so /tmp/shipped.so
for v in "$@"; do
  if [ "$v" = shipped ]; then cp /tmp/shipped.so manigaussian_amd/libmgsplat.so; else cp .alt/libmgsplat_$v.so manigaussian_amd/libmgsplat.so; fi
  [ -n "$AB_NOTEST" ] || timeout 300 python -m pytest tests/test_binning.py tests/test_gpu_parity.py -x -q -m gpu -k "binning or golden or c3" -p no:cacheprovider > $out/pytest_$v.log 2>&1; [ -n "$AB_NOTEST" ] || { echo "$v pytest rc=$?"; tail -1 $out/pytest_$v.log; }
  for c in ${AB_CONFIGS:-ref16k c3}; do
    timeout 200 python bench.py $( if [[ $c == views* ]]; then echo "--config c3 --views ${c#views}"; elif [[ $c == P* ]]; then echo "--config c3 --P ${c#P}"; else echo "--config $c"; fi ) --steps 300 --warmup 50 --mode eager-st --only-mode --no-cpu-baseline --no-reference-kernels > $out/bench_${v}_$c.json 2> $out/bench_${v}_$c.err
    python - $out/bench_${v}_$c.json $v $c <<'P'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], sys.argv[3], "ms/step %.4f" % d["ms_per_step"], {k: round(v * 1e3, 1) for k, v in d["stages_ms"].items() if v})
P
  done
done
cp /tmp/shipped.so manigaussian_amd/libmgsplat.so
