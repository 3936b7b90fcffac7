#!/bin/bash
# round-4 GPU session H: where the step's time is now - kernel statistics with and without side streams, queue activity
set -uo pipefail
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT; tag=r04h
cd /tmp && export TMPDIR=/tmp
GPU_MAX_HW_QUEUES=8 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_${tag} -o p -- python3 $R/bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-kernel-timing --no-exact-f32 > $R/gpurun_out/${tag}_prof.log 2>&1
cd $R
db=$(ls gpurun_out/prof_${tag}/*/p_results.db gpurun_out/prof_${tag}/p_results.db 2>/dev/null | head -1)
python tools/rocpd_stats.py $db --top 70 > gpurun_out/${tag}_kernel_stats.txt
python tools/rocpd_timeline.py $db > gpurun_out/${tag}_timeline.txt 2>&1
python tools/rocpd_queues.py $db 2 > gpurun_out/${tag}_queues.txt 2>&1
rm -rf gpurun_out/prof_${tag}
cd /tmp
GPU_MAX_HW_QUEUES=8 MUVO_STREAMS=0 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_${tag}_off -o p -- python3 $R/bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-kernel-timing --no-exact-f32 > $R/gpurun_out/${tag}_prof_off.log 2>&1
cd $R
db=$(ls gpurun_out/prof_${tag}_off/*/p_results.db gpurun_out/prof_${tag}_off/p_results.db 2>/dev/null | head -1)
python tools/rocpd_stats.py $db --top 90 > gpurun_out/${tag}_kernel_stats_streams_off.txt
rm -rf gpurun_out/prof_${tag}_off
head -40 gpurun_out/${tag}_kernel_stats_streams_off.txt | cut -c1-170
tail -40 gpurun_out/${tag}_queues.txt | cut -c1-200
