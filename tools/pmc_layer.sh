#!/bin/bash
# PMC passes over tools/layer_bench.py for one layer/mode (GPU box).  usage: tools/pmc_layer.sh <tag> <layer_bench args...>
# Writes gpurun_out/pmc_<tag>_pass{1,2,3}/ (rocpd databases); summarise with tools/rocpd_pmc.py.
set -uo pipefail
: ${GRAFT_REPO_ROOT:?}
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
GPU_MAX_HW_QUEUES=8 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES -d $R/gpurun_out/pmc_${tag}_pass1 -o p -- python3 $R/tools/layer_bench.py "$@" > $R/gpurun_out/pmc_${tag}_1.log 2>&1
GPU_MAX_HW_QUEUES=8 rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_INSTS_MFMA -d $R/gpurun_out/pmc_${tag}_pass2 -o p -- python3 $R/tools/layer_bench.py "$@" > $R/gpurun_out/pmc_${tag}_2.log 2>&1
GPU_MAX_HW_QUEUES=8 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VMEM SQ_LDS_IDX_ACTIVE SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_VALU_CVT SQ_INSTS_SMEM GRBM_GUI_ACTIVE -d $R/gpurun_out/pmc_${tag}_pass3 -o p -- python3 $R/tools/layer_bench.py "$@" > $R/gpurun_out/pmc_${tag}_3.log 2>&1
ls $R/gpurun_out/pmc_${tag}_pass*/
