#!/bin/bash
set -uo pipefail
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py tests/test_train_loop.py tests/test_dp_gpu.py -m gpu -q -x -k "repack or b1s2 or deterministic or train_loop or resume or dp" > gpurun_out/r04l_tests.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r04l_tests.log
tail -6 gpurun_out/r04l_tests.log | cut -c1-300
bash tools/ab_env3.sh MUVO_DGR_PACK_SIDE 3 0 1 > gpurun_out/r04l_ab.txt 2>&1; cat gpurun_out/r04l_ab.txt
