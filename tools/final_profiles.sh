#!/bin/bash
# GPU box: the set of measurements a round's profiles/ are made of.  usage: tools/final_profiles.sh <tag>
#   <tag>_bench.json / _conv_layers.txt   default bench line (+ per-layer table)
#   <tag>_kernel_stats.txt                rocprofv3 --kernel-trace --stats of the bench command, side streams on
#   <tag>_kernel_stats_streams_off.txt    the same with MUVO_STREAMS=0 (kernel durations without neighbours on the chip)
#   <tag>_timeline.txt                    busy / small-only / idle shares of the step (tools/rocpd_timeline.py)
#   <tag>_hbm_traffic.json, _pmc_*.txt    FETCH_SIZE / WRITE_SIZE passes (tools/pmc_step.sh)
#   <tag>_force_dist.txt                  bench with a one-rank RCCL group attached (MUVO_BENCH_FORCE_DIST=1) next to the plain one
set -uo pipefail
: ${GRAFT_REPO_ROOT:?}
tag=$1
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd $R
python bench.py --layer-table gpurun_out/${tag}_conv_layers.txt > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err
echo "bench rc=$?"; cut -c1-300 gpurun_out/${tag}_bench.json | tail -1
{
  echo "# python bench.py --steps 20 --warmup 5, one MI355X, same box, alternating"
  for k in 1 2; do
    python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-exact-f32 --no-kernel-timing 2>/dev/null | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('alone                      ms/step', round(d['ms_per_step'],2))"
    MUVO_BENCH_FORCE_DIST=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-exact-f32 --no-kernel-timing 2>/dev/null | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('one-rank RCCL group attached ms/step', round(d['ms_per_step'],2), {k: d[k] for k in d if 'exchange' in k or 'allreduce' in k or 'comm' in k})"
  done
} > gpurun_out/${tag}_force_dist.txt 2>&1
cat gpurun_out/${tag}_force_dist.txt
cd /tmp && export TMPDIR=/tmp
GPU_MAX_HW_QUEUES=8 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_${tag} -o p -- python3 $R/bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-kernel-timing --no-exact-f32 > $R/gpurun_out/${tag}_prof.log 2>&1
cd $R
db=$(ls gpurun_out/prof_${tag}/*/p_results.db gpurun_out/prof_${tag}/p_results.db 2>/dev/null | head -1)
python tools/rocpd_stats.py $db --top 90 > gpurun_out/${tag}_kernel_stats.txt
python tools/rocpd_timeline.py $db > gpurun_out/${tag}_timeline.txt 2>&1
rm -rf gpurun_out/prof_${tag}
cd /tmp
GPU_MAX_HW_QUEUES=8 MUVO_STREAMS=0 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_${tag}_off -o p -- python3 $R/bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-kernel-timing --no-exact-f32 > $R/gpurun_out/${tag}_prof_off.log 2>&1
cd $R
db=$(ls gpurun_out/prof_${tag}_off/*/p_results.db gpurun_out/prof_${tag}_off/p_results.db 2>/dev/null | head -1)
python tools/rocpd_stats.py $db --top 60 > gpurun_out/${tag}_kernel_stats_streams_off.txt
rm -rf gpurun_out/prof_${tag}_off
head -12 gpurun_out/${tag}_kernel_stats.txt | cut -c1-150
head -8 gpurun_out/${tag}_kernel_stats_streams_off.txt | cut -c1-150
bash tools/pmc_step.sh ${tag} | tail -c 600
