/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -shared -fPIC tools/ubench/nccl_latency_double.hip -o /tmp/libnccl_latency_double.so
run() { python tools/bench_step.py "$@" 2>&1 | grep "ms/step" | grep -v "   " | sed "s/.*S=[0-9]*: //; s/  (.*//"; }
for S in 35 45 55 64 75; do
  for rep in 1 2; do
    echo "S=$S none  normal: $(DRS_WG_STREAM_PRIO=0 run B=16 S=$S steps=30)   low: $(DRS_WG_STREAM_PRIO=1 run B=16 S=$S steps=30)   high: $(DRS_WG_STREAM_PRIO=2 run B=16 S=$S steps=30)"
  done
  echo "S=$S double a=15 inline  normal: $(env DRS_WG_STREAM_PRIO=0 DRS_RCCL_LIB=/tmp/libnccl_latency_double.so NCCL_DOUBLE_ALPHA_US=15 NCCL_DOUBLE_GBS=120 bash -c "$(declare -f run); run B=16 S=$S steps=30 comm=rccl")   low: $(env DRS_WG_STREAM_PRIO=1 DRS_RCCL_LIB=/tmp/libnccl_latency_double.so NCCL_DOUBLE_ALPHA_US=15 NCCL_DOUBLE_GBS=120 bash -c "$(declare -f run); run B=16 S=$S steps=30 comm=rccl")   high: $(env DRS_WG_STREAM_PRIO=2 DRS_RCCL_LIB=/tmp/libnccl_latency_double.so NCCL_DOUBLE_ALPHA_US=15 NCCL_DOUBLE_GBS=120 bash -c "$(declare -f run); run B=16 S=$S steps=30 comm=rccl")"
done
