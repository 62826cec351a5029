// DEVELOPMENT AID (tools/collectives_model.sh) -- not part of the product, never loaded unless DRS_RCCL_LIB names it.
//
// A WORLD-1 stand-in for the five NCCL entry points the step engine binds that gives every all-reduce a WIRE TIME: the call enqueues,
// on the stream it is given, a one-wave kernel that sleeps for  alpha + bytes / beta  (NCCL_DOUBLE_ALPHA_US, NCCL_DOUBLE_GBS) and
// leaves the data alone (a sum over one rank).  The collective's stream is held for that long while the rest of the chip stays free
// -- how a latency-bound all-reduce looks to the streams of ONE rank -- so the forms of the library-side collectives (inline / two
// buckets / asynchronous) can be compared for what they EXPOSE of a given wire time.  An estimate under a stated model, nothing more:
// RCCL's own kernels also take CUs, and no rank ever waits for a slower one here.
//   hipcc --offload-arch=gfx950 -O2 -shared -fPIC tools/ubench/nccl_latency_double.hip -o libnccl_latency_double.so
#include <hip/hip_runtime.h>
#include <cstdlib>
#include <cstring>

namespace {
struct UniqueId { char internal[128]; };
double env_or(const char* name, double dflt) { const char* e = std::getenv(name); return e && *e ? std::atof(e) : dflt; }

__global__ void wire_time_kernel(long long ticks) {       // ticks of the 100 MHz constant clock
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
}
}  // namespace

extern "C" {
int ncclGetUniqueId(UniqueId* id) { std::memset(id, 0, sizeof *id); return 0; }
int ncclCommInitRank(void** comm, int world, UniqueId, int rank) {
  if (!comm || world != 1 || rank != 0) return 4;
  *comm = std::malloc(8);
  return *comm ? 0 : 2;
}
int ncclCommDestroy(void* comm) { std::free(comm); return 0; }
int ncclAllReduce(const void* send, void* recv, size_t count, int datatype, int op, void* comm, hipStream_t stream) {
  if (!comm || op != 0) return 4;
  static const double alpha_us = env_or("NCCL_DOUBLE_ALPHA_US", 15.0), gbs = env_or("NCCL_DOUBLE_GBS", 120.0);
  const size_t bytes = count * (datatype == 8 ? 8 : 4);
  if (send != recv && hipMemcpyAsync(recv, send, bytes, hipMemcpyDeviceToDevice, stream) != hipSuccess) return 1;
  const double us = alpha_us + (double)bytes / (gbs * 1e3);
  hipLaunchKernelGGL(wire_time_kernel, dim3(1), dim3(64), 0, stream, (long long)(us * 100.0));
  return hipGetLastError() == hipSuccess ? 0 : 1;
}
const char* ncclGetErrorString(int rc) { return rc ? "error (nccl_latency_double)" : "no error"; }
}
