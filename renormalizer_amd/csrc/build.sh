#!/bin/bash
# Builds libmpsengine.so (gfx950) in-tree.  Usage: renormalizer_amd/csrc/build.sh [extra hipcc flags]
set -euo pipefail
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
SRCS="mpse_core.hip mpse_gemm.hip mpse_contract.hip mpse_small.hip mpse_heff0.hip mpse_vec.hip mpse_qr.hip mpse_qr2.hip mpse_cholqr.hip mpse_svd.hip mpse_davidson.hip"
OBJS=""
PIDS=""
for s in $SRCS; do
  [ -f "$s" ] || continue
  o="build/${s%.hip}.o"
  mkdir -p build
  if [ ! -f "$o" ] || [ "$s" -nt "$o" ] || [ -n "$(find . -maxdepth 1 -name '*.h' -newer "$o" 2>/dev/null)" ] || [ ../../include/mpsengine.h -nt "$o" ]; then
    # -amdgpu-kernarg-preload-count: the first 16 dwords of plain (pointer / scalar) kernel arguments arrive in SGPRs with
    # the wave instead of through a scalar load at its start: the launches of the Krylov chain are short enough for that
    # load to show (+0.8 % headline, +2 % at D = 64, same box)
    $HIPCC --offload-arch=gfx950 -O3 -std=c++17 -fPIC -mllvm -amdgpu-kernarg-preload-count=16 -c "$s" -o "$o" "$@" &
    PIDS="$PIDS $!"
  fi
  OBJS="$OBJS $o"
done
for p in $PIDS; do wait "$p" || { echo "compile failed" >&2; exit 1; }; done
$HIPCC --offload-arch=gfx950 -shared -fPIC $OBJS -o libmpsengine.so
echo "built $(pwd)/libmpsengine.so"
