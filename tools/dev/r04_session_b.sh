#!/bin/bash
# round-4 GPU session B: packed-fp32 reproducer, new parity tests, fake-peer exchange A/B, two-core host A/B, full bench line
set -uo pipefail
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
bash tools/dev/pk_f32_repro.sh 200 > gpurun_out/r04b_pk_repro.txt 2>&1
cat gpurun_out/r04b_pk_repro.txt
python -m pytest tests/test_model_gpu.py -m gpu -q -x -k "loss_curve or headline" > gpurun_out/r04b_tests.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r04b_tests.log
tail -15 gpurun_out/r04b_tests.log
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-exact-f32 --no-kernel-timing --no-extensions"
{
for k in 1 2; do
  $B 2>/dev/null | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('plain                ms/step', round(d['ms_per_step'],2), round(d['median_ms_per_step'],2), 'host_cpu_ms', round(d['host_cpu_ms'],1))"
  MUVO_DP_FAKE_PEERS=8 $B 2>/dev/null | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('fake peers 8          ms/step', round(d['ms_per_step'],2), round(d['median_ms_per_step'],2), json.dumps(d.get('gradient_exchange')))"
  MUVO_BENCH_CORES=2 $B 2>/dev/null | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('2 host cores           ms/step', round(d['ms_per_step'],2), round(d['median_ms_per_step'],2), 'host_cpu_ms', round(d['host_cpu_ms'],1))"
  MUVO_BENCH_CORES=2 MUVO_DP_FAKE_PEERS=8 $B 2>/dev/null | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('2 cores + fake peers 8 ms/step', round(d['ms_per_step'],2), round(d['median_ms_per_step'],2), 'exposed', d['gradient_exchange']['exposed_ms_per_step'])"
  MUVO_SIDE_PRIORITY=s2=low $B 2>/dev/null | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('wgrad stream low prio  ms/step', round(d['ms_per_step'],2), round(d['median_ms_per_step'],2))"
done
} > gpurun_out/r04b_dp_ab.txt 2>&1
cat gpurun_out/r04b_dp_ab.txt
/usr/bin/time -v python bench.py > gpurun_out/r04b_bench.json 2> gpurun_out/r04b_bench.err; echo "bench rc=$?"
grep -E "Elapsed|Maximum resident" gpurun_out/r04b_bench.err
python -c "
import json
d=json.loads([l for l in open('gpurun_out/r04b_bench.json') if l.startswith('{')][-1])
print('ms/step', d['ms_per_step'], 'median', d['median_ms_per_step'])
print('extensions', json.dumps(d.get('extensions')))
print('voxel_class', d.get('voxel_class'))
print('side', d.get('side_streams'))
"
