#!/bin/bash
set -uo pipefail
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r04ap_bench.json 2> gpurun_out/r04ap_bench.err; echo "rc=$?"; grep -c "AccumulateGrad" gpurun_out/r04ap_bench.err; wc -c gpurun_out/r04ap_bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04ap_bench.json').read().strip().split('\n')[-1])
print('mean',round(d['ms_per_step'],2),'median',round(d['median_ms_per_step'],2),'peak_hbm_gb',d['peak_hbm_gb'],'roofline',d['roofline'].get('bracketed_steps'),round(d['roofline']['frac'],3))
PY
timeout 900 python -m pytest tests/test_dp_gpu.py tests/test_model_gpu.py -m gpu -q -x -k "two_rank or lightning or b1s2" 2>&1 | tail -3
