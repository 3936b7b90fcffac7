#!/bin/bash
# Same-box A/B of one environment switch: tools/ab_env.sh VAR valueA valueB [rounds]   (bench.py, 10 steps, interleaved)
var=$1; a=$2; b=$3; rounds=${4:-2}
for r in $(seq $rounds); do
  for v in $a $b; do
    env $var=$v timeout 400 python bench.py --steps 10 --warmup 3 --no-exact-f32 --no-cpu-baseline --no-kernel-timing 2>/dev/null \
      | python -c "import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$var=$v', round(d['ms_per_step'],2), round(d['median_ms_per_step'],2), d['final_loss'])"
  done
done
