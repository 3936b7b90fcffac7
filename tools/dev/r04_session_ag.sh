#!/bin/bash
set -uo pipefail
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
bash tools/ab_env3.sh MUVO_VOX_Z16 3 0 1 > gpurun_out/r04ag_ab.txt 2>&1; cat gpurun_out/r04ag_ab.txt
