#!/bin/bash
set -uo pipefail
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
python tools/dev/z16_case.py 2>&1 | tail -6 | cut -c1-100
timeout 1200 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x > gpurun_out/r04af_tests.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r04af_tests.log
tail -3 gpurun_out/r04af_tests.log | cut -c1-300
timeout 900 python -m pytest tests/test_model_gpu.py -m gpu -q -x -k "b1s2 or b2s4" > gpurun_out/r04af_model.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r04af_model.log
tail -3 gpurun_out/r04af_model.log | cut -c1-300
for i in 1 2; do
python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-exact-f32 --no-extensions --no-kernel-timing > gpurun_out/r04af_$i.json 2>/dev/null
python - <<PY
import json
d=json.loads(open('gpurun_out/r04af_$i.json').read().strip().split('\n')[-1])
print('run $i: mean',round(d['ms_per_step'],2),'median',round(d['median_ms_per_step'],2))
PY
done
