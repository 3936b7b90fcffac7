#!/bin/bash
set -uo pipefail
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
python tools/hbm_bench.py split 2>&1 | tail -12 | cut -c1-160
