#!/bin/bash
# GPU box: the set of measurements a round's profiles/ are made of.  usage: tools/final_profiles.sh <tag>
#   <tag>_bench.json / _conv_layers.txt   default bench line (+ per-layer table)
#   <tag>_kernel_stats.txt                rocprofv3 --kernel-trace --stats of the bench command, side streams on
#   <tag>_kernel_stats_streams_off.txt    the same with MUVO_STREAMS=0 (kernel durations without neighbours on the chip)
#   <tag>_timeline.txt                    busy / small-only / idle shares of the step (tools/rocpd_timeline.py)
#   <tag>_hbm_traffic.json, _pmc_*.txt    FETCH_SIZE / WRITE_SIZE passes (tools/pmc_step.sh)
#   <tag>_dp_evidence.txt                 bench alone / with a one-rank RCCL group / with the stand-in collective of 8 fake peers / on 2 host cores
set -uo pipefail
: ${GRAFT_REPO_ROOT:?}
tag=$1
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd $R
python bench.py --gpus 1 --steps 20 --warmup 5 --layer-table gpurun_out/${tag}_conv_layers.txt > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err
echo "bench rc=$?"; cut -c1-300 gpurun_out/${tag}_bench.json | tail -1
{
  echo "# python bench.py --steps 20 --warmup 5, one MI355X, same box, alternating (ms/step, median; gradient_exchange of the last run)"
  B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-exact-f32 --no-kernel-timing"
  P='import sys,json; d=json.loads(sys.stdin.read()); g=d.get("gradient_exchange") or {}; print(sys.argv[1].ljust(44), round(d["ms_per_step"],2), round(d["median_ms_per_step"],2), "host_cpu_ms", round(d["host_cpu_ms"],1), "exposed_ms", g.get("exposed_ms_per_step"))'
  for k in 1 2; do
    $B 2>/dev/null | grep "^{" | python -c "$P" "alone"
    MUVO_BENCH_FORCE_DIST=1 $B 2>/dev/null | grep "^{" | python -c "$P" "one-rank RCCL group attached"
    MUVO_DP_FAKE_PEERS=8 $B 2>/dev/null | grep "^{" | python -c "$P" "8 fake peers (stand-in collective kernel)"
    MUVO_BENCH_CORES=2 $B 2>/dev/null | grep "^{" | python -c "$P" "2 host cores"
    MUVO_BENCH_CORES=2 MUVO_DP_FAKE_PEERS=8 $B 2>/dev/null | grep "^{" | python -c "$P" "2 host cores + 8 fake peers"
  done
  echo "# per-segment report of the last fake-peer run:"
  MUVO_DP_FAKE_PEERS=8 $B 2>/dev/null | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(json.dumps(d['gradient_exchange'], indent=1))"
} > gpurun_out/${tag}_dp_evidence.txt 2>&1
cat gpurun_out/${tag}_dp_evidence.txt | head -14
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
