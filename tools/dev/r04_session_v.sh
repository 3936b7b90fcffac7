#!/bin/bash
set -uo pipefail
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "conv_family or decoder_block3d or adain or lrelu or vox" > gpurun_out/r04v_tests.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r04v_tests.log
tail -3 gpurun_out/r04v_tests.log | cut -c1-300
echo "== TY=8"; python tools/layer_bench.py --mode bf16x3 --layers vox16,vox32,vox16b --what fwd,dgrad --iters 10 2>&1 | tail -6 | cut -c1-200
echo "== TY=4"; MUVO_HIP_LIB=muvo_amd/build_ab/psty4/libmuvo_hip.so python tools/layer_bench.py --mode bf16x3 --layers vox16,vox32,vox16b --what fwd,dgrad --iters 10 2>&1 | tail -6 | cut -c1-200
MUVO_HIP_LIB=muvo_amd/build_ab/psty4/libmuvo_hip.so timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "conv_family or decoder_block3d" 2>&1 | tail -2
