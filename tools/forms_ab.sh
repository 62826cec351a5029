/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -shared -fPIC tools/ubench/nccl_latency_double.hip -o /tmp/libnccl_latency_double.so
run() { python tools/bench_step.py "$@" 2>&1 | grep "ms/step" | grep -v "   " | sed "s/.*S=[0-9]*: //; s/  (.*//"; }
for S in 64 45; do
  for rep in 1 2; do
    echo "S=$S none prio0: $(DRS_WG_STREAM_PRIO=0 run B=16 S=$S steps=30)   none new: $(run B=16 S=$S steps=30)"
  done
  for form in "" "DRS_RCCL_BUCKETS=2" "DRS_RCCL_ASYNC=1"; do
    echo "S=$S real-RCCL-world-1 [$form] prio0: $(env DRS_WG_STREAM_PRIO=0 $form bash -c "$(declare -f run); run B=16 S=$S steps=30 comm=rccl")   new: $(env $form bash -c "$(declare -f run); run B=16 S=$S steps=30 comm=rccl")"
    for model in "0 1000000" "15 120"; do set -- $model
      echo "S=$S double a=$1 [$form] prio0: $(env DRS_WG_STREAM_PRIO=0 DRS_RCCL_LIB=/tmp/libnccl_latency_double.so NCCL_DOUBLE_ALPHA_US=$1 NCCL_DOUBLE_GBS=$2 $form bash -c "$(declare -f run); run B=16 S=$S steps=30 comm=rccl")   new: $(env DRS_RCCL_LIB=/tmp/libnccl_latency_double.so NCCL_DOUBLE_ALPHA_US=$1 NCCL_DOUBLE_GBS=$2 $form bash -c "$(declare -f run); run B=16 S=$S steps=30 comm=rccl")"
    done
  done
done
