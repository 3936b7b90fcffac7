#!/bin/bash
# round-4 GPU session E: BatchNorm-writes-planes kernels and wiring: kernel tests, model parity, same-box A/B
set -uo pipefail
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "planes or batchnorm or basic_block" > gpurun_out/r04e_tests.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r04e_tests.log
tail -30 gpurun_out/r04e_tests.log | cut -c1-300
python -m pytest tests/test_model_gpu.py -m gpu -q -x -k "b1s2 or b2s4" > gpurun_out/r04e_model.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r04e_model.log
tail -12 gpurun_out/r04e_model.log | cut -c1-300
bash tools/ab_env3.sh MUVO_BN_PLANES 3 0 1 > gpurun_out/r04e_ab.txt 2>&1; cat gpurun_out/r04e_ab.txt
