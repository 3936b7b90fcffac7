#!/bin/bash
set -uo pipefail
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-exact-f32 --no-extensions --layer-table gpurun_out/r04w_layers.txt > gpurun_out/r04w_bench.json 2> gpurun_out/r04w_bench.err; tail -c 600 gpurun_out/r04w_bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04w_bench.json').read().strip().split('\n')[-1])
print('ms_per_step',d['ms_per_step'],'median',d.get('median_ms_per_step'))
print('voxel_class',d.get('voxel_class'))
kc=d.get('kernel_classes',{})
for k,v in sorted(kc.items(), key=lambda kv:-kv[1]['seconds']): print(f"{k:40s} {v['seconds']*1e3/ d.get('kernel_class_steps',2):8.3f} ms/step  launches {v['launches']}")
PY
