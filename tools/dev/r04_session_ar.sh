#!/bin/bash
set -uo pipefail
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "emitted_planes or conv_stage_with_head or conv_family" > gpurun_out/r04ar_tests.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r04ar_tests.log
tail -25 gpurun_out/r04ar_tests.log | cut -c1-300
