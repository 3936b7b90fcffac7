#!/bin/bash
set -uo pipefail
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for v in 1 2 3 4; do echo "== VOX_PSA=$v"; MUVO_HIP_LIB=muvo_amd/build_ab/psa$v/libmuvo_hip.so python tools/layer_bench.py --mode bf16x3 --layers vox16,vox32,vox16b --what fwd --iters 10 2>&1 | tail -3 | cut -c1-200; done
