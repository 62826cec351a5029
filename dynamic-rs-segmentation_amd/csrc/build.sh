#!/bin/bash
# Build the C-ABI library of HIP kernels for gfx950 (MI355X) in-tree.  hipcc cross-compiles without a GPU.
set -e
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
OUT=../libdrs_hip.so
FLAGS="--offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wall -Wno-unused-function"
for f in conv_mfma conv_split pointwise patches engine; do
  if [ -n "$DRS_FORCE_REBUILD" ] || [ ! -f $f.o ] || [ $f.hip -nt $f.o ] || [ drs_common.hpp -nt $f.o ] || [ ../../include/drs.h -nt $f.o ]; then
    $HIPCC $FLAGS -c $f.hip -o $f.o "$@"
  fi
done
$HIPCC --offload-arch=gfx950 -shared -fPIC -o $OUT conv_mfma.o conv_split.o pointwise.o patches.o engine.o
echo "built $(readlink -f $OUT)"
