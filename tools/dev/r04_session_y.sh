#!/bin/bash
set -uo pipefail
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
python tools/dev/z16_case.py 2>&1 | tail -7 | cut -c1-250
