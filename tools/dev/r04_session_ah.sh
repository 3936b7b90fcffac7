#!/bin/bash
set -uo pipefail
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
rocm-smi --showclocks --showpower --showtemp 2>/dev/null | grep -v "^=\|^$" | head -12
for i in 1 2; do
python bench.py --gpus 1 --steps 60 --warmup 2 --no-cpu-baseline --no-exact-f32 --no-extensions --no-kernel-timing > gpurun_out/r04ah_$i.json 2>/dev/null
python - <<PY
import json
d=json.loads(open('gpurun_out/r04ah_$i.json').read().strip().split('\n')[-1])
print('run $i: mean',round(d['ms_per_step'],2),'median',round(d['median_ms_per_step'],2),'steps',[round(v,1) for v in d['step_ms']])
PY
rocm-smi --showclocks --showpower --showtemp 2>/dev/null | grep -i "sclk\|mclk\|power\|Temperature (Sensor junction)\|edge" | head -8
done
