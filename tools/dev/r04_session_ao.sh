#!/bin/bash
set -uo pipefail
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
bash tools/ab_env3.sh MUVO_BF3_WGRAD_TPT 3 1 2 > gpurun_out/r04ao_ab.txt 2>&1; cat gpurun_out/r04ao_ab.txt
for v in 1 2; do MUVO_BF3_WGRAD_TPT=$v python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-exact-f32 --no-extensions --layer-table gpurun_out/r04ao_layers_$v.txt > /dev/null 2>&1; echo "TPT=$v"; grep "small_tile:wgrad" gpurun_out/r04ao_layers_$v.txt | head -8 | cut -c1-150; done
