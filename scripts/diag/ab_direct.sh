for c in c3 ref16k c2; do for d in 0 32768 0 32768; do
  timeout 200 python bench.py --config $c --steps 300 --warmup 50 --mode graph --only-mode --no-cpu-baseline --no-reference-kernels --dbg $d 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$c dbg=$d graph ms/step %.4f'%d['ms_per_step'], {k: round(v*1e3,1) for k,v in d['stages_ms'].items() if v})"
done; done
