#!/bin/bash
# round-4 GPU session C: packed-fp32 reproducers (standalone v2 + the library-based one), kernel tests on the rebuilt library,
# the full default bench line (extensions, exact-f32 with conv_gemm.hip built with packed fp32 again)
set -uo pipefail
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
bash tools/dev/pk_f32_repro.sh 200 > gpurun_out/r04c_pk_repro.txt 2>&1
{ echo "# library-based reproducer (tools/dev/coresidency_repro2.py conv 50): victim built standalone with hipcc defaults, aggressor = the library's bf16x3 convolution"
  timeout 600 python tools/dev/coresidency_repro2.py conv 50 2>&1 | grep -v "^/opt\|Warning" | tail -8; } >> gpurun_out/r04c_pk_repro.txt 2>&1
cat gpurun_out/r04c_pk_repro.txt
python -m pytest tests/test_kernels_gpu.py tests/test_abi.py -m gpu -q -x > gpurun_out/r04c_tests.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r04c_tests.log
tail -4 gpurun_out/r04c_tests.log
s=$(date +%s)
python bench.py > gpurun_out/r04c_bench.json 2> gpurun_out/r04c_bench.err; echo "bench rc=$? in $(( $(date +%s) - s )) s"
python -c "
import json
d=json.loads([l for l in open('gpurun_out/r04c_bench.json') if l.startswith('{')][-1])
print('ms/step', d['ms_per_step'], 'median', d['median_ms_per_step'])
print('exact_f32', d.get('exact_f32'))
print('extensions', json.dumps(d.get('extensions')))
print('voxel_class', d.get('voxel_class'))
print('side', d.get('side_streams'))
print('classes', {k: round(v['seconds']*500,2) for k,v in d['kernel_classes'].items()})
"
