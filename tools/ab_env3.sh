#!/bin/bash
# Same-box comparison of several values of one environment switch: tools/ab_env3.sh VAR rounds v1 v2 v3 ...
var=$1; rounds=$2; shift 2
for r in $(seq $rounds); do
  for v in "$@"; do
    env $var=$v timeout 400 python bench.py --steps 10 --warmup 3 --no-exact-f32 --no-cpu-baseline --no-kernel-timing 2>/dev/null \
      | python -c "import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$var=$v', round(d['ms_per_step'],2), round(d['median_ms_per_step'],2))"
  done
done
