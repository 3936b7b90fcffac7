#!/bin/bash
set -uo pipefail
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "conv_family or decoder_block3d or adain or lrelu" > gpurun_out/r04n_tests.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r04n_tests.log
tail -12 gpurun_out/r04n_tests.log | cut -c1-300
timeout 900 python -m pytest tests/test_model_gpu.py -m gpu -q -x -k "b1s2 or b2s4" > gpurun_out/r04n_model.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r04n_model.log
tail -6 gpurun_out/r04n_model.log | cut -c1-300
bash tools/ab_env3.sh MUVO_VOX_PS 3 0 1 > gpurun_out/r04n_ab.txt 2>&1; cat gpurun_out/r04n_ab.txt
for v in 0 1; do MUVO_VOX_PS=$v python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-exact-f32 --no-extensions --layer-table gpurun_out/r04n_layers_$v.txt > /dev/null 2>&1; echo "MUVO_VOX_PS=$v"; grep "vox_bf16x3" gpurun_out/r04n_layers_$v.txt | cut -c1-150; done
