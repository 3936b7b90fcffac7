#!/bin/bash
# GPU box: kernel trace of a short bench run -> idle time between dispatches.  usage: tools/prof_gaps.sh <tag> [steps]
: ${GRAFT_REPO_ROOT:?}
R=$GRAFT_REPO_ROOT; tag=$1; steps=${2:-6}
cd /tmp && export TMPDIR=/tmp
GPU_MAX_HW_QUEUES=8 rocprofv3 --kernel-trace -d $R/gpurun_out/prof_${tag} -o p -- python3 $R/bench.py --steps $steps --warmup 2 --no-cpu-baseline --no-kernel-timing --no-exact-f32 > $R/gpurun_out/${tag}_prof.log 2>&1
cd $R
db=$(ls gpurun_out/prof_${tag}/*/p_results.db gpurun_out/prof_${tag}/p_results.db 2>/dev/null | head -1)
python tools/rocpd_gaps.py $db $steps > gpurun_out/${tag}_gaps.txt
rm -rf gpurun_out/prof_${tag}
cat gpurun_out/${tag}_gaps.txt
