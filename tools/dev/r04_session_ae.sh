#!/bin/bash
set -uo pipefail
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
python tools/dev/z16_case.py 2>&1 | tail -7 | cut -c1-250
timeout 1200 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x > gpurun_out/r04ae_tests.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r04ae_tests.log
tail -8 gpurun_out/r04ae_tests.log | cut -c1-300
for v in 0 1; do echo "== MUVO_VOX_PS=$v"; MUVO_VOX_PS=$v python tools/layer_bench.py --mode bf16x3 --layers vox64s,vox32s --what fwd,dgrad,wgrad --iters 10 2>&1 | tail -6 | cut -c1-200; done
