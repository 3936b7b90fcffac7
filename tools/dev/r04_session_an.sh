#!/bin/bash
set -uo pipefail
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x > gpurun_out/r04an_tests.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r04an_tests.log
tail -6 gpurun_out/r04an_tests.log | cut -c1-300
for v in 1 2 4; do echo "== MUVO_BF3_WGRAD_TPT=$v"; MUVO_BF3_WGRAD_TPT=$v python tools/layer_bench.py --mode bf16x3 --layers res64,res64rv,ds64,vox64 --what wgrad --iters 10 2>&1 | tail -4 | cut -c1-200; done
