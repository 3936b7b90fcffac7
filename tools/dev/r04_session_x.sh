#!/bin/bash
set -uo pipefail
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "conv_family or decoder_block3d or adain or lrelu or vox" > gpurun_out/r04x_tests.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r04x_tests.log
tail -15 gpurun_out/r04x_tests.log | cut -c1-300
for v in 0 1; do echo "== MUVO_VOX_WGRAD_PS=$v"; MUVO_VOX_WGRAD_PS=$v python tools/layer_bench.py --mode bf16x3 --layers vox64s,vox32s --what fwd,dgrad,wgrad --iters 10 2>&1 | tail -6 | cut -c1-200; done
