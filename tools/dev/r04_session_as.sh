#!/bin/bash
set -uo pipefail
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_model_gpu.py -m gpu -q -x -k "b1s2 or b2s4" > gpurun_out/r04as_model.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r04as_model.log
tail -4 gpurun_out/r04as_model.log | cut -c1-300
bash tools/ab_env3.sh MUVO_EMIT_PLANES 3 0 1 > gpurun_out/r04as_ab.txt 2>&1; cat gpurun_out/r04as_ab.txt
