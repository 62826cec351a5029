// DEVELOPMENT AID (tests/test_gpu_dp.py, tools/bench_step.py BENCH_RCCL_LIB=) -- not part of the product, never loaded unless a caller names it through drs_rccl_bind_library.
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

// NCCL_DOUBLE_POISON=1 (tests/test_gpu_dp.py): while a sum is "on the wire" its buffer holds NaNs -- saved first, restored last.  A
// consumer that is not ordered behind the collective's stream then reads NaNs, a producer that is not ordered in front of it gets its
// values overwritten by the stale copy: both show in the step's results, which real RCCL at world 1 (no launch at all) never would.
__global__ void save_and_poison_kernel(unsigned int* buf, unsigned int* keep, size_t words) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < words; i += (size_t)gridDim.x * blockDim.x) { keep[i] = buf[i]; buf[i] = 0xffffffffu; }
}
__global__ void restore_kernel(unsigned int* buf, const unsigned int* keep, size_t words) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < words; i += (size_t)gridDim.x * blockDim.x) buf[i] = keep[i];
}
struct Comm { unsigned int* keep; size_t keep_words; };
}  // namespace

extern "C" {
int ncclGetUniqueId(UniqueId* id) { std::memset(id, 0, sizeof *id); return 0; }
int ncclCommInitRank(void** comm, int world, UniqueId, int rank) {
  if (!comm || world != 1 || rank != 0) return 4;
  Comm* c = (Comm*)std::calloc(1, sizeof(Comm));
  *comm = c;
  return c ? 0 : 2;
}
int ncclCommDestroy(void* comm) {
  Comm* c = (Comm*)comm;
  if (c && c->keep) (void)hipFree(c->keep);
  std::free(c);
  return 0;
}
int ncclAllReduce(const void* send, void* recv, size_t count, int datatype, int op, void* comm, hipStream_t stream) {
  if (!comm || op != 0) return 4;
  static const double alpha_us = env_or("NCCL_DOUBLE_ALPHA_US", 15.0), gbs = env_or("NCCL_DOUBLE_GBS", 120.0);
  const size_t bytes = count * (datatype == 8 ? 8 : 4);
  if (send != recv && hipMemcpyAsync(recv, send, bytes, hipMemcpyDeviceToDevice, stream) != hipSuccess) return 1;
  const double us = alpha_us + (double)bytes / (gbs * 1e3);
  static const bool poison = env_or("NCCL_DOUBLE_POISON", 0.0) != 0.0;
  Comm* c = (Comm*)comm;
  const size_t words = bytes / 4;
  if (poison && words) {      // (one sum at a time per communicator: its launches are in stream order, and the engine never has two sums of ONE communicator in flight)
    if (c->keep_words < words) {
      if (c->keep) { (void)hipDeviceSynchronize(); (void)hipFree(c->keep); c->keep = nullptr; }
      if (hipMalloc((void**)&c->keep, (words > (4u << 20) ? words : (4u << 20)) * 4) != hipSuccess) return 1;
      c->keep_words = words > (4u << 20) ? words : (4u << 20);
    }
    const int blocks = (int)((words + 255) / 256 < 1024 ? (words + 255) / 256 : 1024);
    hipLaunchKernelGGL(save_and_poison_kernel, dim3(blocks), dim3(256), 0, stream, (unsigned int*)recv, c->keep, words);
    hipLaunchKernelGGL(wire_time_kernel, dim3(1), dim3(64), 0, stream, (long long)(us * 100.0));
    hipLaunchKernelGGL(restore_kernel, dim3(blocks), dim3(256), 0, stream, (unsigned int*)recv, (const unsigned int*)c->keep, words);
  } else {
    hipLaunchKernelGGL(wire_time_kernel, dim3(1), dim3(64), 0, stream, (long long)(us * 100.0));
  }
  return hipGetLastError() == hipSuccess ? 0 : 1;
}
const char* ncclGetErrorString(int rc) { return rc ? "error (nccl_latency_double)" : "no error"; }
}
