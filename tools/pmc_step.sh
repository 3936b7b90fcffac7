#!/bin/bash
# HBM-side traffic of every kernel of the training step (GPU box): FETCH_SIZE and WRITE_SIZE in separate rocprofv3 passes
# over a short bench.py run, as /opt/skills/guides/MI355X_MICROARCH.md prescribes.  usage: tools/pmc_step.sh <tag>
set -uo pipefail
: ${GRAFT_REPO_ROOT:?}
tag=$1
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
GPU_MAX_HW_QUEUES=8 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/gpurun_out/pmc_${tag}_rd -o p -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-exact-f32 --no-kernel-timing > $R/gpurun_out/pmc_${tag}_rd.log 2>&1
GPU_MAX_HW_QUEUES=8 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/gpurun_out/pmc_${tag}_wr -o p -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-exact-f32 --no-kernel-timing > $R/gpurun_out/pmc_${tag}_wr.log 2>&1
cd $R
rd=$(ls gpurun_out/pmc_${tag}_rd/*/p_results.db gpurun_out/pmc_${tag}_rd/p_results.db 2>/dev/null | head -1)
wr=$(ls gpurun_out/pmc_${tag}_wr/*/p_results.db gpurun_out/pmc_${tag}_wr/p_results.db 2>/dev/null | head -1)
python tools/rocpd_pmc.py $rd > gpurun_out/${tag}_pmc_fetch.txt
python tools/rocpd_pmc.py $wr > gpurun_out/${tag}_pmc_write.txt
python tools/pmc_summary.py $rd $wr > gpurun_out/${tag}_hbm_traffic.json
rm -rf gpurun_out/pmc_${tag}_rd gpurun_out/pmc_${tag}_wr
cat gpurun_out/${tag}_hbm_traffic.json
