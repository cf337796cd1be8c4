#!/bin/bash
# Builds libsavsr_hip.so for gfx950 in-tree (the .so is git-ignored but travels with gpurun).
set -euo pipefail
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="--offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wall -Wno-unused-function"
OBJS=()
for f in conv_mfma.hip osconv.hip elementwise.hip satu.hip tail.hip metrics.hip resize.hip; do
  o="${f%.hip}.o"
  if [ ! -f "$o" ] || [ "$f" -nt "$o" ] || [ common.hpp -nt "$o" ] || [ ../../include/savsr_hip.h -nt "$o" ]; then
    $HIPCC $FLAGS -c "$f" -o "$o" &
  fi
  OBJS+=("$o")
done
if [ ! -f api.o ] || [ api.cpp -nt api.o ] || [ common.hpp -nt api.o ] || [ ../../include/savsr_hip.h -nt api.o ]; then
  $HIPCC $FLAGS -x hip -c api.cpp -o api.o &
fi
wait
$HIPCC --offload-arch=gfx950 -shared -fPIC -o libsavsr_hip.so "${OBJS[@]}" api.o
echo "built $(pwd)/libsavsr_hip.so"
