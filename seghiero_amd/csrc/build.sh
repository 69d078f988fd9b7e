#!/bin/bash
# Build libseghiero_hip.so for gfx950 (in-tree; cross-compiles without a GPU).
set -euo pipefail
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-function"
OBJS=()
pids=()
for f in conv_gemm conv_bf16x6 conv_x6p bn_elementwise pool_resample dwconv loss loss3 sgd; do
  if [ ! -f "$f.o" ] || [ "$f.hip" -nt "$f.o" ] || [ common.h -nt "$f.o" ] || [ loss_common.h -nt "$f.o" ] || [ conv_x6.h -nt "$f.o" ] || [ ../../include/seghiero_hip.h -nt "$f.o" ]; then
    $HIPCC $FLAGS -c "$f.hip" -o "$f.o" &
    pids+=($!)
  fi
  OBJS+=("$f.o")
done
for p in "${pids[@]:-}"; do [ -n "$p" ] && wait "$p"; done
$HIPCC --offload-arch=gfx950 -shared -fPIC -o ../libseghiero_hip.so "${OBJS[@]}"
echo "built $(cd .. && pwd)/libseghiero_hip.so"
