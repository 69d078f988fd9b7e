#!/bin/bash
# Build libseghiero_hip.so for gfx950 (in-tree; cross-compiles without a GPU).
set -euo pipefail
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-function"
OBJS=()
pids=()
stale() {   # object $1 older than source $2 or any shared header
  [ ! -f "$1" ] || [ "$2" -nt "$1" ] || [ common.h -nt "$1" ] || [ loss_common.h -nt "$1" ] || [ conv_x6.h -nt "$1" ] || [ ../../include/seghiero_hip.h -nt "$1" ]
}
# the pipelined conv kernels compile as six translation units (fprop, dgrad, four wgrad families: SH_X6P_PART), started first
for part in 3 4 5 6 1 2; do
  if stale "conv_x6p_$part.o" conv_x6p.hip; then
    $HIPCC $FLAGS -DSH_X6P_PART=$part -c conv_x6p.hip -o "conv_x6p_$part.o" &
    pids+=($!)
  fi
  OBJS+=("conv_x6p_$part.o")
done
for f in conv_b16 conv_gemm conv_bf16x6 bn_elementwise pool_resample dwconv loss loss3 sgd comm; do
  if stale "$f.o" "$f.hip"; then
    $HIPCC $FLAGS -c "$f.hip" -o "$f.o" &
    pids+=($!)
  fi
  OBJS+=("$f.o")
done
for p in "${pids[@]:-}"; do [ -n "$p" ] && wait "$p"; done          # (set -e: a failed compile stops the build here)
$HIPCC --offload-arch=gfx950 -shared -fPIC -o ../libseghiero_hip.so "${OBJS[@]}" -ldl
echo "built $(cd .. && pwd)/libseghiero_hip.so"
