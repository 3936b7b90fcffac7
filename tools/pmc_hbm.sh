#!/bin/bash
# HBM traffic of one layer (GPU box): FETCH_SIZE and WRITE_SIZE in separate passes, as /opt/skills/guides/MI355X_MICROARCH.md
# prescribes.  usage: tools/pmc_hbm.sh <tag> <layer_bench args...>
set -uo pipefail
: ${GRAFT_REPO_ROOT:?}
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
GPU_MAX_HW_QUEUES=8 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/gpurun_out/hbm_${tag}_rd -o p -- python3 $R/tools/layer_bench.py "$@" > $R/gpurun_out/hbm_${tag}_rd.log 2>&1
GPU_MAX_HW_QUEUES=8 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/gpurun_out/hbm_${tag}_wr -o p -- python3 $R/tools/layer_bench.py "$@" > $R/gpurun_out/hbm_${tag}_wr.log 2>&1
ls $R/gpurun_out/hbm_${tag}_rd $R/gpurun_out/hbm_${tag}_wr
