#!/bin/bash
# Build the C-ABI libraries of HIP kernels for gfx950 (MI355X) in-tree.  hipcc cross-compiles without a GPU.
#
#   build.sh            compile what is out of date (a source, drs_common.hpp or include/drs.h newer than its object), link
#   build.sh force      compile every source (what __graft_entry__.build() does: the driver's "does it build" check compiles)
#   build.sh clean      remove objects and libraries
# (The CPU-only sanitizer build of the host code is a separate recipe, sanitize/build_host_asan.sh: nothing here or under tests/
# builds or runs it.)
#
# Two libraries come out of the same sources:
#   ../libdrs_hip.so       the product: exports exactly what include/drs.h declares
#   ../libdrs_hip_dev.so   -DDRS_DEV: additionally exports the development switches of include/drs_dev.h (kernel-form / cut /
#                          skip A/B switches, process-global); used by the A/B tools and by the tests that hold two kernel forms
#                          bitwise equal -- never by the package's product path
# Writes build_info.json (mode, sources compiled, seconds) next to this script.
set -e
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
MODE=${1:-incremental}
[ -n "$DRS_FORCE_REBUILD" ] && [ "$MODE" = incremental ] && MODE=force
SRCS="conv_mfma conv_split pointwise patches engine rccl_comm"
FLAGS="--offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wall -Wno-unused-function"
T0=$SECONDS

if [ "$MODE" = clean ]; then
  rm -rf obj obj_dev *.o ../libdrs_hip.so ../libdrs_hip_dev.so build_info.json
  echo "cleaned"
  exit 0
fi

mkdir -p obj obj_dev
COMPILED=""
PIDS=""
for f in $SRCS; do
  for v in obj obj_dev; do
    o=$v/$f.o
    if [ "$MODE" = force ] || [ ! -f $o ] || [ $f.hip -nt $o ] || [ drs_common.hpp -nt $o ] || [ ../../include/drs.h -nt $o ] || [ ../../include/drs_dev.h -nt $o ]; then
      DEF=""; [ $v = obj_dev ] && DEF="-DDRS_DEV"
      $HIPCC $FLAGS $DEF -c $f.hip -o $o &
      PIDS="$PIDS $!"
      COMPILED="$COMPILED \"$v/$f\","
    fi
  done
done
for p in $PIDS; do wait $p; done
$HIPCC --offload-arch=gfx950 -shared -fPIC -o ../libdrs_hip.so $(for f in $SRCS; do echo obj/$f.o; done) -ldl
$HIPCC --offload-arch=gfx950 -shared -fPIC -o ../libdrs_hip_dev.so $(for f in $SRCS; do echo obj_dev/$f.o; done) -ldl
T1=$SECONDS
printf '{"build_mode": "%s", "build_exercised": %s, "compiled": [%s], "seconds": %d}\n' "$MODE" \
  "$([ -n "$COMPILED" ] && echo true || echo false)" "${COMPILED%,}" "$((T1 - T0))" > build_info.json
echo "built $(readlink -f ../libdrs_hip.so) and libdrs_hip_dev.so ($(cat build_info.json))"
