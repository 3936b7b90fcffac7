#!/bin/bash
# One GPU-box session: GPU tests, bench line, rocprofv3 kernel trace of the bench.  usage: tools/gpu_session.sh <tag> [pytest -k expr]
set -uo pipefail
: ${GRAFT_REPO_ROOT:?}
tag=$1; kexpr=${2:-}
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd $R
if [ -n "$kexpr" ]; then
  python -m pytest tests -m gpu -q --maxfail=12 -k "$kexpr" > gpurun_out/${tag}_tests.log 2>&1
else
  python -m pytest tests -m gpu -q --maxfail=12 > gpurun_out/${tag}_tests.log 2>&1
fi
echo "pytest rc=$?" >> gpurun_out/${tag}_tests.log
tail -5 gpurun_out/${tag}_tests.log
python bench.py --steps 10 --warmup 3 --layer-table gpurun_out/${tag}_conv_layers.txt > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err
echo "bench rc=$?"; cut -c1-600 gpurun_out/${tag}_bench.json
cd /tmp && export TMPDIR=/tmp
GPU_MAX_HW_QUEUES=8 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_${tag} -o p -- python3 $R/bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-kernel-timing --no-exact-f32 > $R/gpurun_out/${tag}_prof.log 2>&1
cd $R
db=$(ls gpurun_out/prof_${tag}/*/p_results.db gpurun_out/prof_${tag}/p_results.db 2>/dev/null | head -1)
python tools/rocpd_stats.py $db --top 90 > gpurun_out/${tag}_kernel_stats.txt
rm -rf gpurun_out/prof_${tag}
head -30 gpurun_out/${tag}_kernel_stats.txt | cut -c1-160
