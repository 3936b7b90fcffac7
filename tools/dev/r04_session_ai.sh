#!/bin/bash
set -uo pipefail
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 3300 python -m pytest tests -m gpu -q -x > gpurun_out/r04ai_full.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r04ai_full.log
tail -15 gpurun_out/r04ai_full.log | cut -c1-300
