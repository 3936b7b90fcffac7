#!/bin/bash
# GPU box: build a second copy of the library with extra -D flags and run a command with it (MUVO_HIP_LIB).
# usage: tools/ab_build.sh "<-D flags>" <tag> -- <command...>
set -euo pipefail
flags=$1; tag=$2; shift 3
cd ${GRAFT_REPO_ROOT:?}
mkdir -p /tmp/ab_$tag
for f in muvo_amd/csrc/*.hip; do
  b=$(basename $f .hip)
  if [ "$b" = conv_bf3 ]; then
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -munsafe-fp-atomics -Wno-unused-result -Wno-unused-value -ffp-contract=off -mllvm -instcombine-max-copied-from-constant-users=8000 $flags -c $f -o /tmp/ab_$tag/$b.o
  else
    cp muvo_amd/build/$b.o /tmp/ab_$tag/$b.o
  fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/ab_$tag/libmuvo_hip.so /tmp/ab_$tag/*.o
MUVO_HIP_LIB=/tmp/ab_$tag/libmuvo_hip.so "$@"
