#!/bin/bash
set -uo pipefail
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for v in 1 0; do echo "== MUVO_VOX_PS=$v MUVO_VOX_WGRAD_PS=$v MUVO_VOX_Z16=$v"; MUVO_VOX_PS=$v MUVO_VOX_WGRAD_PS=$v MUVO_VOX_Z16=$v python tools/dev/dp_checksum_diag.py 2>&1 | tail -6 | cut -c1-400; done
