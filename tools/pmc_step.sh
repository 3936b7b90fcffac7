#!/bin/bash
# HBM-side traffic of every kernel of the training step (GPU box): FETCH_SIZE and WRITE_SIZE in separate rocprofv3 passes
# over a short bench.py run, as /opt/skills/guides/MI355X_MICROARCH.md prescribes.  usage: tools/pmc_step.sh <tag>
tag=$1
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/gpurun_out/pmc_${tag}_rd -o p -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/pmc_${tag}_rd.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/gpurun_out/pmc_${tag}_wr -o p -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/pmc_${tag}_wr.log 2>&1
cd $R
python tools/rocpd_pmc.py $(ls gpurun_out/pmc_${tag}_rd/*/p_results.db gpurun_out/pmc_${tag}_rd/p_results.db 2>/dev/null | head -1) > gpurun_out/${tag}_pmc_fetch.txt
python tools/rocpd_pmc.py $(ls gpurun_out/pmc_${tag}_wr/*/p_results.db gpurun_out/pmc_${tag}_wr/p_results.db 2>/dev/null | head -1) > gpurun_out/${tag}_pmc_write.txt
rm -rf gpurun_out/pmc_${tag}_rd gpurun_out/pmc_${tag}_wr
