#!/bin/bash
set -uo pipefail
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
python -m muvo_amd.build > gpurun_out/r04m_build.log 2>&1 || { tail -5 gpurun_out/r04m_build.log; exit 1; }
timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "stem" > gpurun_out/r04m_tests.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r04m_tests.log
tail -12 gpurun_out/r04m_tests.log | cut -c1-300
timeout 900 python -m pytest tests/test_model_gpu.py -m gpu -q -x -k "b1s2 and policy" > gpurun_out/r04m_model.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r04m_model.log
tail -6 gpurun_out/r04m_model.log | cut -c1-300
bash tools/ab_env3.sh MUVO_STEM_KERNEL 3 0 1 > gpurun_out/r04m_ab.txt 2>&1; cat gpurun_out/r04m_ab.txt
python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-exact-f32 --no-extensions --layer-table gpurun_out/r04m_layers.txt > /dev/null 2>&1
grep -n "k(1, 7, 7)" gpurun_out/r04m_layers.txt | cut -c1-200
