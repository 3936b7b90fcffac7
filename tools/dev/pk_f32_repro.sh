#!/bin/bash
# GPU box: builds tools/dev/pk_f32_repro.hip in three variants and runs each (see the header of the .hip file).
#   packed (-ffp-contract=off, the library's flags minus the workaround) | fma (default contraction) | control (no packed fp32 ops)
set -uo pipefail
here=$(cd "$(dirname "$0")" && pwd)
n=${1:-200}
for v in packed fma control; do
  case $v in
    packed)  flags="-ffp-contract=off" ;;
    fma)     flags="" ;;
    control) flags="-ffp-contract=off -Xclang -target-feature -Xclang -packed-fp32-ops" ;;
  esac
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 $flags "$here/pk_f32_repro.hip" -o /tmp/pk_repro_$v 2>/dev/null || { echo "$v: build failed"; continue; }
  for cfg in "150 1" "150 0" "64 1"; do
    echo -n "$v, aggressor (LDS KB, mode) = ($cfg): "
    timeout 300 /tmp/pk_repro_$v $n $cfg | tail -2 | tr '\n' ' '
    echo
  done
done
