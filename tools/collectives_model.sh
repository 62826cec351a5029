#!/bin/bash
# One rank's step (B = 16) with every collective of the step forced on at world 1 and given a WIRE TIME by a stand-in for RCCL
# (tools/ubench/nccl_latency_double.hip: each all-reduce holds its stream for alpha + bytes / beta, data untouched): what each form
# of the library-side collectives EXPOSES of that time.  An estimate under the stated model, not a measurement of xGMI.
#   bash tools/collectives_model.sh <subdir of gpurun_out> [S ...]
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$1; mkdir -p $O; shift
SIDES=${@:-64}
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -shared -fPIC $R/tools/ubench/nccl_latency_double.hip -o /tmp/libnccl_latency_double.so || exit 1
cd $R
for S in $SIDES; do
  echo "== B=16 S=$S, no collectives"
  python tools/bench_step.py B=16 S=$S steps=30 2>&1 | grep "ms/step" | grep -v "   "
  for model in "0 1000000" "8 200" "15 120" "25 80"; do
    set -- $model
    echo "== B=16 S=$S, wire time of an all-reduce = $1 us + bytes / $2 GB/s  (2.5 KB sum: $(python3 -c "print(round($1 + 2560 / ($2 * 1e3), 1))") us, 8.37 MB gradient: $(python3 -c "print(round($1 + 8366360 / ($2 * 1e3), 1))") us)"
    for form in "" "DRS_RCCL_BUCKETS=2" "DRS_RCCL_ASYNC=1"; do
      env DRS_RCCL_LIB=/tmp/libnccl_latency_double.so NCCL_DOUBLE_ALPHA_US=$1 NCCL_DOUBLE_GBS=$2 $form python tools/bench_step.py B=16 S=$S steps=30 comm=rccl 2>&1 | grep "ms/step" | grep -v "   " | sed 's/  host enqueue.*//'
    done
  done
done
