#!/bin/bash
set -uo pipefail
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x > gpurun_out/r04aa_tests.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r04aa_tests.log
tail -6 gpurun_out/r04aa_tests.log | cut -c1-300
timeout 900 python -m pytest tests/test_model_gpu.py -m gpu -q -x -k "b1s2 or b2s4" > gpurun_out/r04aa_model.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r04aa_model.log
tail -6 gpurun_out/r04aa_model.log | cut -c1-300
bash tools/ab_env3.sh MUVO_PACK_TILED 2 0 1 > gpurun_out/r04aa_ab1.txt 2>&1; cat gpurun_out/r04aa_ab1.txt
bash tools/ab_env3.sh MUVO_UNPACK_TILED 2 0 1 > gpurun_out/r04aa_ab2.txt 2>&1; cat gpurun_out/r04aa_ab2.txt
