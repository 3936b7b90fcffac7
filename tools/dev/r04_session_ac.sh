#!/bin/bash
set -uo pipefail
: ${GRAFT_REPO_ROOT:?}
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x > gpurun_out/r04ac_tests.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r04ac_tests.log
tail -3 gpurun_out/r04ac_tests.log | cut -c1-300
timeout 900 python -m pytest tests/test_model_gpu.py -m gpu -q -x -k "b1s2 or b2s4" > gpurun_out/r04ac_model.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r04ac_model.log
tail -3 gpurun_out/r04ac_model.log | cut -c1-300
run() {  # tag, env...
  tag=$1; shift
  cd /tmp && export TMPDIR=/tmp
  ( export "$@"; export GPU_MAX_HW_QUEUES=8 MUVO_STREAMS=0; rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_$tag -o p -- python3 $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-kernel-timing --no-exact-f32 > $R/gpurun_out/${tag}_prof.log 2>&1 )
  cd $R
  db=$(ls gpurun_out/prof_$tag/*/p_results.db gpurun_out/prof_$tag/p_results.db 2>/dev/null | head -1)
  python tools/rocpd_stats.py $db --top 90 > gpurun_out/${tag}_kernel_stats.txt
  rm -rf gpurun_out/prof_$tag
  echo "== $tag"; grep -n "unpack\|pack_table\|bf3_pack" gpurun_out/${tag}_kernel_stats.txt | cut -c1-140
}
run r04ac_new MUVO_PACK_TILED=1 MUVO_UNPACK_TILED=1
bash tools/ab_env3.sh MUVO_PACK_TILED 2 0 1 > gpurun_out/r04ac_ab1.txt 2>&1; cat gpurun_out/r04ac_ab1.txt
bash tools/ab_env3.sh MUVO_UNPACK_TILED 2 0 1 > gpurun_out/r04ac_ab2.txt 2>&1; cat gpurun_out/r04ac_ab2.txt
