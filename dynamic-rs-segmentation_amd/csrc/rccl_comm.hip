// Library-side collectives: RCCL (= NCCL on ROCm, over xGMI inside a node) bound at run time with dlopen, so that the sums of a
// data-parallel training step are issued from the step engine itself (engine.hip) instead of crossing C -> ctypes -> Python ->
// torch.distributed ~23 times a step (`drs_allreduce(handle, comm)`; the reference is single-process, isprs:1707).
//
// The library does not link librccl: a single-GPU host never needs it.  `librccl.so.1` is looked up among the objects the
// process has already loaded (a PyTorch host has loaded its own copy) and then on the loader path / under /opt/rocm/lib;
// drs_rccl_bind_library(path), called by the host before anything else of this file, names another NCCL-API library to bind instead
// (an explicit call of the host program: no environment variable substitutes the collectives library).
// A communicator made here belongs to the library's copy of RCCL; a host that links RCCL itself may hand in its own ncclComm_t
// (drs_net_set_rccl takes opaque pointers) provided both sides resolve to the same loaded librccl.
#include "drs_common.hpp"
#include "../../include/drs.h"

#include <dlfcn.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>

namespace {

struct UniqueId { char internal[128]; };      // ncclUniqueId (rccl.h: NCCL_UNIQUE_ID_BYTES = 128)
typedef int (*get_unique_id_fn)(UniqueId*);
typedef int (*comm_init_rank_fn)(void**, int, UniqueId, int);
typedef int (*comm_destroy_fn)(void*);
typedef int (*all_reduce_fn)(const void*, void*, size_t, int, int, void*, hipStream_t);
typedef const char* (*error_string_fn)(int);

struct Rccl {
  void* handle = nullptr;
  get_unique_id_fn get_unique_id = nullptr;
  comm_init_rank_fn comm_init_rank = nullptr;
  comm_destroy_fn comm_destroy = nullptr;
  all_reduce_fn all_reduce = nullptr;
  error_string_fn error_string = nullptr;
  bool ok = false;
};

Rccl g_rccl;
std::once_flag g_once;
std::mutex g_named_mutex;
char g_named[1024] = {0};      // drs_rccl_bind_library: the NCCL-API library to bind instead of the process's / the loader's librccl

void load_rccl() {
  const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  // A library named through drs_rccl_bind_library (another build of RCCL; the shared-memory stand-in the tests drive the world > 1
  // code of the step engine with on a one-GPU box, tests/c/nccl_shm_double.cpp).  Named and not loadable: no fall-back to the
  // copies below -- the caller asked for THAT library.
  void* h = nullptr;
  if (g_named[0]) {
    h = dlopen(g_named, RTLD_NOW | RTLD_LOCAL);
    if (!h) { std::fprintf(stderr, "libdrs_hip: drs_rccl_bind_library(%s): could not be loaded: %s\n", g_named, dlerror()); return; }
  }
  if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_NOLOAD);          // the copy the host process already runs, if any
  for (int i = 0; !h && i < 3; ++i) h = dlopen(names[i], RTLD_NOW | RTLD_LOCAL);
  if (!h) return;
  g_rccl.handle = h;
  g_rccl.get_unique_id = (get_unique_id_fn)dlsym(h, "ncclGetUniqueId");
  g_rccl.comm_init_rank = (comm_init_rank_fn)dlsym(h, "ncclCommInitRank");
  g_rccl.comm_destroy = (comm_destroy_fn)dlsym(h, "ncclCommDestroy");
  g_rccl.all_reduce = (all_reduce_fn)dlsym(h, "ncclAllReduce");
  g_rccl.error_string = (error_string_fn)dlsym(h, "ncclGetErrorString");
  g_rccl.ok = g_rccl.get_unique_id && g_rccl.comm_init_rank && g_rccl.comm_destroy && g_rccl.all_reduce;
}

const Rccl* rccl() {
  std::call_once(g_once, load_rccl);
  return g_rccl.ok ? &g_rccl : nullptr;
}

int report(const Rccl* r, const char* what, int rc) {
  if (rc == 0) return DRS_OK;
  std::fprintf(stderr, "libdrs_hip: %s failed: %s (ncclResult %d)\n", what, r && r->error_string ? r->error_string(rc) : "?", rc);
  return DRS_ERR_HIP;
}

}  // namespace

// used by engine.hip: one sum all-reduce, in place, on `stream`; dtype as in drs_net_buffer_info (0 f32, 1 f64, 3 i32)
__attribute__((visibility("hidden"))) int drs_rccl_all_reduce_sum(void* comm, void* ptr, size_t count, int dtype, hipStream_t stream) {
  const Rccl* r = rccl();
  if (!r || !comm || !ptr) return DRS_ERR_ARG;
  static const int kType[4] = {7 /* ncclFloat32 */, 8 /* ncclFloat64 */, -1, 2 /* ncclInt32 */};
  if (dtype < 0 || dtype > 3 || kType[dtype] < 0) return DRS_ERR_ARG;
  if (count == 0) return DRS_OK;
  return report(r, "ncclAllReduce", r->all_reduce(ptr, ptr, count, kType[dtype], 0 /* ncclSum */, comm, stream));
}

extern "C" {

// Bind `path` (an NCCL-API shared library: ncclGetUniqueId, ncclCommInitRank, ncclCommDestroy, ncclAllReduce) for the library-side
// collectives instead of librccl.  Only before the first use of any drs_rccl_* / drs_net_set_rccl entry point: once a library is
// bound (this one or librccl) the call is rejected (DRS_ERR_ARG), and so is a path that cannot be loaded.
int drs_rccl_bind_library(const char* path) {
  if (!path || !*path || std::strlen(path) >= sizeof g_named) return DRS_ERR_ARG;
  {
    std::lock_guard<std::mutex> lock(g_named_mutex);
    if (g_rccl.handle || g_named[0]) return DRS_ERR_ARG;
    std::strcpy(g_named, path);
  }
  return rccl() ? DRS_OK : DRS_ERR_ARG;
}

int drs_rccl_available(void) { return rccl() ? 1 : 0; }

int drs_rccl_unique_id(unsigned char* id128) {
  const Rccl* r = rccl();
  if (!r || !id128) return DRS_ERR_ARG;
  UniqueId id;
  std::memset(&id, 0, sizeof id);
  const int rc = report(r, "ncclGetUniqueId", r->get_unique_id(&id));
  if (rc == DRS_OK) std::memcpy(id128, id.internal, 128);
  return rc;
}

int drs_rccl_comm_create(int world, int rank, const unsigned char* id128, void** comm) {
  const Rccl* r = rccl();
  if (!r || !id128 || !comm || world < 1 || rank < 0 || rank >= world) return DRS_ERR_ARG;
  UniqueId id;
  std::memcpy(id.internal, id128, 128);
  *comm = nullptr;
  return report(r, "ncclCommInitRank", r->comm_init_rank(comm, world, id, rank));
}

int drs_rccl_comm_destroy(void* comm) {
  const Rccl* r = rccl();
  if (!r) return DRS_ERR_ARG;
  if (!comm) return DRS_OK;
  return report(r, "ncclCommDestroy", r->comm_destroy(comm));
}

int drs_rccl_all_reduce(void* comm, void* dev_ptr, size_t count, int dtype, void* stream) {
  return drs_rccl_all_reduce_sum(comm, dev_ptr, count, dtype, (hipStream_t)stream);
}

}  // extern "C"
