#!/bin/bash
# round-4 GPU session K: suite after the launch-hygiene changes; loss-curve tests with their final bars; two bench lines
set -uo pipefail
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
rm -f gpurun_out/loss_curve.txt
python -m pytest tests -m gpu -q --maxfail=8 > gpurun_out/r04k_tests.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r04k_tests.log
tail -8 gpurun_out/r04k_tests.log | cut -c1-300
grep drift gpurun_out/loss_curve.txt
for k in 1 2; do python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-exact-f32 --no-kernel-timing 2>/dev/null | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ms/step', round(d['ms_per_step'],2), round(d['median_ms_per_step'],2))"; done
