#!/bin/bash
set -uo pipefail
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
echo "== new kernels"; python tools/dev/evalres_diag.py 2>&1 | tail -9 | cut -c1-200
echo "== ps off"; MUVO_VOX_PS=0 MUVO_VOX_WGRAD_PS=0 MUVO_VOX_Z16=0 python tools/dev/evalres_diag.py 2>&1 | tail -9 | cut -c1-200
echo "== ps off, stem off"; MUVO_STEM_KERNEL=0 MUVO_VOX_PS=0 MUVO_VOX_WGRAD_PS=0 MUVO_VOX_Z16=0 python tools/dev/evalres_diag.py 2>&1 | tail -9 | cut -c1-200
echo "== f32"; python tools/dev/evalres_diag.py f32 2>&1 | tail -9 | cut -c1-200
