// Multi-GPU pixel gather behind the C-ABI (dyn_gather_tiles; the counterpart of nn.DataParallel's gather of the rendered
// chunk outputs, reference ibrnet/model.py:134-159 + render_image.py:60-217).  Rays shard across ranks as contiguous tiles with no
// data-path collective; the one exchange step of a frame is an all-gather of every rank's packed [tile, C] pixel rows over RCCL / xGMI.
//
// RCCL is resolved at run time from the process image (dlsym; dlopen("librccl.so") only if nothing is loaded yet): a PyTorch host has
// already loaded its own librccl, and a communicator is only valid inside the library instance that made it, so the kernels' library
// must not link a second copy.  A host without PyTorch creates the communicator through dyn_comm_* below (or passes its own ncclComm_t).
#include <dlfcn.h>
#include <stdlib.h>
#include <string.h>

#include "dyn_host.h"

struct DynNcclId {  // ncclUniqueId (rccl.h: NCCL_UNIQUE_ID_BYTES = 128), passed BY VALUE to ncclCommInitRank
  char internal[128];
};
namespace {
typedef int (*fn_all_gather)(const void*, void*, size_t, int, void*, hipStream_t);
typedef int (*fn_get_unique_id)(void*);
typedef int (*fn_comm_init_rank)(void**, int, DynNcclId, int);
typedef int (*fn_comm_destroy)(void*);
typedef int (*fn_comm_count)(const void*, int*);
typedef const char* (*fn_error_string)(int);
struct Rccl {
  fn_all_gather all_gather;
  fn_get_unique_id get_unique_id;
  fn_comm_init_rank comm_init_rank;
  fn_comm_destroy comm_destroy;
  fn_comm_count comm_count, comm_user_rank;
  fn_error_string error_string;
  bool ok;
};

// ONE source for the whole table: a communicator is only valid inside the library instance that made it, so the entry points must not mix two
// RCCL copies (a host that exposes only part of NCCL globally, a preloaded shim).  If ncclAllGather is already visible in the process image, the
// library that defines it is found with dladdr and re-opened with RTLD_NOLOAD, and every symbol comes from that handle; otherwise every symbol
// comes from the librccl this file opens itself.
void* rccl_handle() {
  static void* handle = [] () -> void* {
    if (void* p = dlsym(RTLD_DEFAULT, "ncclAllGather")) {
      Dl_info info;
      if (dladdr(p, &info) && info.dli_fname) {
        if (void* h = dlopen(info.dli_fname, RTLD_NOW | RTLD_NOLOAD)) return h;
      }
    }
    const char* env = getenv("DYNIBAR_RCCL_LIB");
    const char* names[] = {env, "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so"};
    for (const char* n : names) {
      if (!n) continue;
      if (void* h = dlopen(n, RTLD_NOW | RTLD_GLOBAL)) return h;
    }
    return nullptr;
  }();
  return handle;
}
void* rccl_symbol(const char* name) {
  void* h = rccl_handle();
  return h ? dlsym(h, name) : nullptr;
}

const Rccl& rccl() {
  static Rccl r = [] {
    Rccl x;
    x.all_gather = (fn_all_gather)rccl_symbol("ncclAllGather");
    x.get_unique_id = (fn_get_unique_id)rccl_symbol("ncclGetUniqueId");
    x.comm_init_rank = (fn_comm_init_rank)rccl_symbol("ncclCommInitRank");
    x.comm_destroy = (fn_comm_destroy)rccl_symbol("ncclCommDestroy");
    x.comm_count = (fn_comm_count)rccl_symbol("ncclCommCount");
    x.comm_user_rank = (fn_comm_count)rccl_symbol("ncclCommUserRank");
    x.error_string = (fn_error_string)rccl_symbol("ncclGetErrorString");
    x.ok = x.all_gather && x.get_unique_id && x.comm_init_rank && x.comm_destroy && x.comm_count && x.comm_user_rank;
    return x;
  }();
  return r;
}

int rccl_fail(const char* what, int rc) {
  const Rccl& r = rccl();
  dyn_set_error("%s: RCCL error %d (%s)", what, rc, r.error_string ? r.error_string(rc) : "?");
  return DYN_E_LAUNCH;
}
}  // namespace

#define DYN_NEED_RCCL(what) DYN_REQUIRE(rccl().ok, what ": RCCL is not loaded in this process and librccl.so could not be opened (set DYNIBAR_RCCL_LIB)")

extern "C" int dyn_comm_available(void) { return rccl().ok ? 1 : 0; }

extern "C" int dyn_comm_unique_id(void* id128) {
  DYN_REQUIRE(id128 != nullptr, "dyn_comm_unique_id: null pointer");
  DYN_NEED_RCCL("dyn_comm_unique_id");
  const int rc = rccl().get_unique_id(id128);
  return rc == 0 ? 0 : rccl_fail("dyn_comm_unique_id", rc);
}

extern "C" int dyn_comm_init_rank(void** comm, int nranks, const void* id128, int rank) {
  DYN_REQUIRE(comm && id128 && nranks >= 1 && rank >= 0 && rank < nranks, "dyn_comm_init_rank: bad argument");
  DYN_NEED_RCCL("dyn_comm_init_rank");
  DynNcclId id;
  memcpy(&id, id128, sizeof(id));
  const int rc = rccl().comm_init_rank(comm, nranks, id, rank);
  return rc == 0 ? 0 : rccl_fail("dyn_comm_init_rank", rc);
}

extern "C" int dyn_comm_destroy(void* comm) {
  if (comm == nullptr) return 0;
  DYN_NEED_RCCL("dyn_comm_destroy");
  const int rc = rccl().comm_destroy(comm);
  return rc == 0 ? 0 : rccl_fail("dyn_comm_destroy", rc);
}

extern "C" int dyn_comm_size_rank(void* comm, int* nranks, int* rank) {
  DYN_REQUIRE(comm && nranks && rank, "dyn_comm_size_rank: null pointer");
  DYN_NEED_RCCL("dyn_comm_size_rank");
  int rc = rccl().comm_count(comm, nranks);
  if (rc == 0) rc = rccl().comm_user_rank(comm, rank);
  return rc == 0 ? 0 : rccl_fail("dyn_comm_size_rank", rc);
}

// recv [nranks][rows_per_rank][cols] <- every rank's send [rows_per_rank][cols] (fp32), in rank order, on `stream`.  Tiles are padded to the
// common size by the caller (render_image.ray_tile: sizes differ by at most one ray), so the collective is one equal-count ncclAllGather.
extern "C" int dyn_gather_tiles(const float* send, float* recv, long rows_per_rank, int cols, void* comm, void* stream) {
  DYN_REQUIRE(send && recv && comm && rows_per_rank > 0 && cols > 0, "dyn_gather_tiles: bad argument");
  DYN_NEED_RCCL("dyn_gather_tiles");
  const int rc = rccl().all_gather(send, recv, (size_t)rows_per_rank * cols, /* ncclFloat32 */ 7, comm, (hipStream_t)stream);
  return rc == 0 ? 0 : rccl_fail("dyn_gather_tiles", rc);
}
