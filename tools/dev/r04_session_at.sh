#!/bin/bash
set -uo pipefail
: ${GRAFT_REPO_ROOT:?}
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
run() {
  tag=$1; shift
  cd /tmp && export TMPDIR=/tmp
  ( export "$@"; export GPU_MAX_HW_QUEUES=8 MUVO_STREAMS=0; rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_$tag -o p -- python3 $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-kernel-timing --no-exact-f32 --no-extensions > $R/gpurun_out/${tag}_prof.log 2>&1 )
  cd $R
  db=$(ls gpurun_out/prof_$tag/*/p_results.db gpurun_out/prof_$tag/p_results.db 2>/dev/null | head -1)
  python tools/rocpd_stats.py $db --top 90 > gpurun_out/${tag}_kernel_stats.txt
  rm -rf gpurun_out/prof_$tag
  echo "== $tag"; head -2 gpurun_out/${tag}_kernel_stats.txt | tail -1 | cut -c1-120; grep -n "nchw_split\|conv_bf3_kernel" gpurun_out/${tag}_kernel_stats.txt | cut -c1-130
}
run r04at_off MUVO_EMIT_PLANES=0
run r04at_on MUVO_EMIT_PLANES=1
