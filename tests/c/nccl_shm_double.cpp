// TEST INFRASTRUCTURE -- not part of the product, never loaded unless a test names it through drs_rccl_bind_library.
//
// A stand-in for the five NCCL entry points the step engine binds (dynamic-rs-segmentation_amd/csrc/rccl_comm.hip:
// ncclGetUniqueId, ncclCommInitRank, ncclCommDestroy, ncclAllReduce, ncclGetErrorString) for a box with ONE GPU, where RCCL
// itself refuses two ranks on one device: the ranks are processes that share the device, a communicator is a POSIX
// shared-memory segment, and a sum all-reduce is "wait for the stream, copy this rank's operand to its slot, barrier, add the
// slots in RANK order, barrier, copy the sum back" -- host-synchronous, the same bits on every rank.  With it the tests drive the
// world > 1 code of the library-side collectives (engine.hip: inline / two buckets / asynchronous forms, the two-stream backward
// pass beside them) that otherwise only ever runs at world 1 here (tests/test_gpu_dp.py).
//
// What it checks and what it cannot: the operand is read after hipStreamSynchronize(stream) of the stream the engine names, so
// a sum issued on a stream that is NOT ordered behind its producer reads a stale operand here as it would under RCCL; the result
// lands with a blocking copy, so a consumer that is not ordered behind the collective's stream is NOT caught (RCCL would leave
// that a race).
//
//   g++ -O1 -shared -fPIC -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include tests/c/nccl_shm_double.cpp -L/opt/rocm/lib -lamdhip64 -lrt -o libnccl_shm_double.so
#include <hip/hip_runtime_api.h>

#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <thread>
#include <vector>

#include <fcntl.h>
#include <sys/mman.h>
#include <unistd.h>

namespace {

constexpr size_t SLOT_BYTES = 16u << 20;       // one rank's operand per round trip (longer operands go in pieces)
constexpr int MAX_WORLD = 8;
constexpr double LIMIT_S = 120.0;              // a rank that never arrives fails the call instead of hanging the test

struct Header {
  std::atomic<int> arrived;
  std::atomic<int> generation;
  std::atomic<int> attached;
};

struct Comm {
  int world, rank;
  Header* h;
  unsigned char* slots;
  size_t bytes;
  char name[96];
};

struct UniqueId { char internal[128]; };

bool barrier(Comm* c) {
  const int gen = c->h->generation.load(std::memory_order_acquire);
  if (c->h->arrived.fetch_add(1, std::memory_order_acq_rel) + 1 == c->world) {
    c->h->arrived.store(0, std::memory_order_relaxed);
    c->h->generation.fetch_add(1, std::memory_order_acq_rel);
    return true;
  }
  const auto t0 = std::chrono::steady_clock::now();
  int spins = 0;
  while (c->h->generation.load(std::memory_order_acquire) == gen) {
    if (++spins > 2000) std::this_thread::sleep_for(std::chrono::microseconds(50));
    else std::this_thread::yield();
    if ((spins & 1023) == 0 && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > LIMIT_S) return false;
  }
  return true;
}

template <class T>
void add_slots(const Comm* c, size_t n, T* out) {
  for (size_t i = 0; i < n; ++i) {
    T s = reinterpret_cast<const T*>(c->slots)[i];                       // rank 0 first, then 1, 2, ...: one order on every rank
    for (int r = 1; r < c->world; ++r) s += reinterpret_cast<const T*>(c->slots + (size_t)r * SLOT_BYTES)[i];
    out[i] = s;
  }
}

}  // namespace

extern "C" {

int ncclGetUniqueId(UniqueId* id) {
  static std::atomic<int> counter{0};
  std::memset(id, 0, sizeof *id);
  unsigned seed = (unsigned)std::chrono::steady_clock::now().time_since_epoch().count();
  std::snprintf(id->internal, sizeof id->internal, "/drs_nccl_double_%d_%d_%08x", (int)getpid(), counter.fetch_add(1), seed);
  return 0;
}

int ncclCommInitRank(void** comm, int world, UniqueId id, int rank) {
  if (!comm || world < 1 || world > MAX_WORLD || rank < 0 || rank >= world) return 4;      // ncclInvalidArgument
  Comm* c = new (std::nothrow) Comm;
  if (!c) return 2;
  c->world = world; c->rank = rank;
  std::memset(c->name, 0, sizeof c->name);
  std::strncpy(c->name, id.internal, sizeof c->name - 1);
  c->bytes = 4096 + (size_t)world * SLOT_BYTES;
  const int fd = shm_open(c->name, O_CREAT | O_RDWR, 0600);
  if (fd < 0 || ftruncate(fd, (off_t)c->bytes) != 0) { if (fd >= 0) close(fd); delete c; return 2; }      // (a new segment reads as zeros: the header starts at 0 / 0 / 0)
  void* p = mmap(nullptr, c->bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  close(fd);
  if (p == MAP_FAILED) { delete c; return 2; }
  c->h = reinterpret_cast<Header*>(p);
  c->slots = reinterpret_cast<unsigned char*>(p) + 4096;
  c->h->attached.fetch_add(1);
  if (!barrier(c)) { munmap(p, c->bytes); delete c; return 1; }          // collective, as ncclCommInitRank is
  *comm = c;
  return 0;
}

int ncclCommDestroy(void* comm) {
  Comm* c = reinterpret_cast<Comm*>(comm);
  if (!c) return 0;
  if (c->h->attached.fetch_sub(1) == 1) shm_unlink(c->name);             // the last one out removes the name
  munmap(c->h, c->bytes);
  delete c;
  return 0;
}

// datatype: 2 ncclInt32, 7 ncclFloat32, 8 ncclFloat64 (the three the step engine uses); op: 0 ncclSum
int ncclAllReduce(const void* send, void* recv, size_t count, int datatype, int op, void* comm, hipStream_t stream) {
  Comm* c = reinterpret_cast<Comm*>(comm);
  const size_t es = datatype == 8 ? 8 : 4;
  if (!c || !send || !recv || op != 0 || (datatype != 2 && datatype != 7 && datatype != 8)) return 4;
  if (hipStreamSynchronize(stream) != hipSuccess) return 1;
  std::vector<unsigned char> out(SLOT_BYTES);
  for (size_t off = 0; off < count; off += SLOT_BYTES / es) {
    const size_t n = count - off < SLOT_BYTES / es ? count - off : SLOT_BYTES / es;
    if (hipMemcpy(c->slots + (size_t)c->rank * SLOT_BYTES, reinterpret_cast<const unsigned char*>(send) + off * es, n * es, hipMemcpyDeviceToHost) != hipSuccess) return 1;
    if (!barrier(c)) return 1;
    if (datatype == 7) add_slots<float>(c, n, reinterpret_cast<float*>(out.data()));
    else if (datatype == 8) add_slots<double>(c, n, reinterpret_cast<double*>(out.data()));
    else add_slots<int32_t>(c, n, reinterpret_cast<int32_t*>(out.data()));
    if (!barrier(c)) return 1;                                             // every rank has read every slot: they may be rewritten
    if (hipMemcpy(reinterpret_cast<unsigned char*>(recv) + off * es, out.data(), n * es, hipMemcpyHostToDevice) != hipSuccess) return 1;
  }
  return 0;
}

const char* ncclGetErrorString(int rc) {
  return rc == 0 ? "no error" : rc == 4 ? "invalid argument (nccl_shm_double)" : rc == 2 ? "system error (nccl_shm_double: shared memory)"
                                                                               : "a rank did not arrive / HIP error (nccl_shm_double)";
}

}  // extern "C"
