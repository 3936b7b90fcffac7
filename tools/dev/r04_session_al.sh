#!/bin/bash
set -uo pipefail
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for i in 1 2; do
python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-exact-f32 --no-extensions > gpurun_out/r04al_$i.json 2>/dev/null
python - <<PY
import json
d=json.loads(open('gpurun_out/r04al_$i.json').read().strip().split('\n')[-1])
print('run $i: mean',round(d['ms_per_step'],2),'median',round(d['median_ms_per_step'],2),'steps',d['step_ms'])
PY
done
