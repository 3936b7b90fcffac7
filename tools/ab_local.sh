#!/bin/bash
# Build a variant of the library HERE (hipcc cross-compiles) into muvo_amd/build_ab/<tag>/ so that it travels to the GPU box:
#   tools/ab_local.sh <tag> "<extra -D flags>" file1.hip [file2.hip ...]     (the other objects come from muvo_amd/build)
# then on the GPU box:  MUVO_HIP_LIB=muvo_amd/build_ab/<tag>/libmuvo_hip.so python tools/layer_bench.py ...
set -euo pipefail
tag=$1; flags=$2; shift 2
cd "$(dirname "$0")/.."
out=muvo_amd/build_ab/$tag
mkdir -p $out
cp muvo_amd/build/*.o $out/
for f in "$@"; do
  b=$(basename $f .hip)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -munsafe-fp-atomics -Wno-unused-result -Wno-unused-value -ffp-contract=off -mllvm -instcombine-max-copied-from-constant-users=8000 $flags -c muvo_amd/csrc/$b.hip -o $out/$b.o
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $out/libmuvo_hip.so $out/*.o
rm -f $out/*.o
echo built $out/libmuvo_hip.so
