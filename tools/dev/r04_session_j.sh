#!/bin/bash
# round-4 GPU session J: the whole GPU suite on the current code, then a same-box A/B of this session's switches
set -uo pipefail
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
rm -f gpurun_out/loss_curve.txt gpurun_out/headline_parity.txt gpurun_out/argmax_flips.txt
python -m pytest tests -m gpu -q --maxfail=8 > gpurun_out/r04j_tests.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r04j_tests.log
tail -15 gpurun_out/r04j_tests.log | cut -c1-300
grep drift gpurun_out/loss_curve.txt
bash tools/ab_env3.sh MUVO_BN_PLANES 2 0 1 > gpurun_out/r04j_ab.txt 2>&1; cat gpurun_out/r04j_ab.txt
