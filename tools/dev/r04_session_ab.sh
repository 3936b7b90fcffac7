#!/bin/bash
set -uo pipefail
: ${GRAFT_REPO_ROOT:?}
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
run() {  # tag, env...
  tag=$1; shift
  cd /tmp && export TMPDIR=/tmp
  env "$@" GPU_MAX_HW_QUEUES=8 MUVO_STREAMS=0 true
  ( export "$@"; export GPU_MAX_HW_QUEUES=8 MUVO_STREAMS=0; rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_$tag -o p -- python3 $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-kernel-timing --no-exact-f32 > $R/gpurun_out/${tag}_prof.log 2>&1 )
  cd $R
  db=$(ls gpurun_out/prof_$tag/*/p_results.db gpurun_out/prof_$tag/p_results.db 2>/dev/null | head -1)
  python tools/rocpd_stats.py $db --top 90 > gpurun_out/${tag}_kernel_stats.txt
  rm -rf gpurun_out/prof_$tag
  echo "== $tag"; grep -n "unpack\|pack_table\|bf3_pack" gpurun_out/${tag}_kernel_stats.txt | cut -c1-140
}
run r04ab_old MUVO_PACK_TILED=0 MUVO_UNPACK_TILED=0
run r04ab_new MUVO_PACK_TILED=1 MUVO_UNPACK_TILED=1
run r04ab_new6k MUVO_PACK_TILED=1 MUVO_UNPACK_TILED=1 MUVO_UNPACK_LDS=6144
