#!/bin/bash
# round-4 GPU session D: the new parity tests (loss curve x3, headline with float64 gradient truth, EVAL.RESOLUTION, deployment forward)
set -uo pipefail
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
rm -f gpurun_out/loss_curve.txt gpurun_out/headline_parity.txt
python -m pytest tests/test_model_gpu.py tests/test_eval_resolution.py tests/test_sim_forward.py tests/test_abi.py -m gpu -q -k "loss_curve or headline or resolution or deployment or abi or resize" > gpurun_out/r04d_tests.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r04d_tests.log
tail -25 gpurun_out/r04d_tests.log
grep drift gpurun_out/loss_curve.txt
cat gpurun_out/headline_parity.txt
