// dfx_comm.cpp -- the multi-GPU exchange step behind the C ABI (include/dfx.h, "multi-GPU exchange"): RCCL over xGMI, called directly.
//
// The reference has no multi-GPU path; its factors are consumed one by one on the host (core/gtsam/photometric_factor.cpp:105-180).
// SURVEY section 8e defines the exchange of the sharded design: every rank evaluates a contiguous shard of the pair list and then either
//   reduce  : assembles its pairs into the keyframe graph's flat block-sparse system (dfx_graph_assemble_async) and ONE ncclReduce /
//             ncclAllReduce sums the ranks' buffers (1.3 MB for 64 keyframes / 1024 pairs at CS = 32), or
//   gather  : all-gathers the ranks' result items (4152 bytes per pair) so that the host can emit one HessianFactor per pair.
// Both are latency-bound single collectives; xGMI is point to point, so RCCL's ring over the 7 links is the transport.
//
// librccl is resolved at run time (dlopen): libdfx.so has no link-time dependency on it, a process that already has an RCCL loaded
// (PyTorch ships its own copy) keeps using that one instead of a second copy with its own state, and single-GPU users never load it.
// DFX_RCCL_LIB names a specific library (tests/cpp/comm_test.cpp runs two ranks over a host-memory stand-in through this hook).
#include "../../include/dfx.h"
#include "dfx_rccl_abi.hpp"

#include <dlfcn.h>
#include <hip/hip_runtime_api.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>

// from dfx_api.cpp
extern "C" int dfx_internal_fail(int code, const char* msg);
extern "C" void* dfx_internal_exchange_stream(dfx_ctx* ctx);
extern "C" int dfx_internal_ctx_device(dfx_ctx* ctx);
extern "C" void* dfx_internal_main_stream(dfx_ctx* ctx);
extern "C" int dfx_internal_note_write(dfx_ctx* ctx, void* ptr);

namespace {

// the subset of rccl.h this file needs: dfx_rccl_abi.hpp (checked against the real header by tests/cpp/rccl_abi_check.cpp)
using NcclUniqueId = dfx_rccl::UniqueId;
using NcclComm = dfx_rccl::Comm;
enum { kNcclSuccess = dfx_rccl::kSuccess, kNcclSum = dfx_rccl::kSum, kNcclUint8 = dfx_rccl::kUint8, kNcclFloat = dfx_rccl::kFloat };
struct Rccl {
  void* handle = nullptr;
  dfx_rccl::GetUniqueId_t GetUniqueId = nullptr;
  dfx_rccl::CommInitRank_t CommInitRank = nullptr;
  dfx_rccl::CommDestroy_t CommDestroy = nullptr;
  dfx_rccl::Reduce_t Reduce = nullptr;
  dfx_rccl::AllReduce_t AllReduce = nullptr;
  dfx_rccl::AllGather_t AllGather = nullptr;
  dfx_rccl::Broadcast_t Broadcast = nullptr;
  dfx_rccl::GetErrorString_t GetErrorString = nullptr;
  std::string path;
};
Rccl g_rccl;
std::mutex g_rccl_mu;

int fail(int code, const std::string& msg) { return dfx_internal_fail(code, msg.c_str()); }

int load_rccl() {
  std::lock_guard<std::mutex> lk(g_rccl_mu);
  if (g_rccl.handle) return DFX_OK;
  void* h = nullptr;
  std::string tried;
  auto attempt = [&](const char* name, int flags) {
    if (h) return;
    h = dlopen(name, flags);
    if (h) g_rccl.path = name; else { tried += name; tried += (flags & RTLD_NOLOAD) ? " (loaded?) " : " "; }
  };
  if (const char* ev = std::getenv("DFX_RCCL_LIB"); ev && *ev) {   // an empty value counts as unset (dlopen("") would hand back the main program)
    attempt(ev, RTLD_NOW | RTLD_LOCAL);
    if (!h) return fail(DFX_E_INVALID, std::string("DFX_RCCL_LIB=") + ev + " cannot be loaded: " + dlerror());
  }
  // an RCCL the process has already loaded wins (one copy, one set of state); then the system's
  attempt("librccl.so", RTLD_NOW | RTLD_NOLOAD);
  attempt("librccl.so.1", RTLD_NOW | RTLD_NOLOAD);
  attempt("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
  attempt("librccl.so", RTLD_NOW | RTLD_LOCAL);
  attempt("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_LOCAL);
  if (!h) return fail(DFX_E_HIP, "librccl not found (tried: " + tried + "); set DFX_RCCL_LIB");
  Rccl r;
  r.handle = h; r.path = g_rccl.path;
  bool ok = true;
  auto sym = [&](const char* n) { void* p = dlsym(h, n); if (!p) ok = false; return p; };
  r.GetUniqueId = reinterpret_cast<dfx_rccl::GetUniqueId_t>(sym("ncclGetUniqueId"));
  r.CommInitRank = reinterpret_cast<dfx_rccl::CommInitRank_t>(sym("ncclCommInitRank"));
  r.CommDestroy = reinterpret_cast<dfx_rccl::CommDestroy_t>(sym("ncclCommDestroy"));
  r.Reduce = reinterpret_cast<dfx_rccl::Reduce_t>(sym("ncclReduce"));
  r.AllReduce = reinterpret_cast<dfx_rccl::AllReduce_t>(sym("ncclAllReduce"));
  r.AllGather = reinterpret_cast<dfx_rccl::AllGather_t>(sym("ncclAllGather"));
  r.Broadcast = reinterpret_cast<dfx_rccl::Broadcast_t>(sym("ncclBroadcast"));
  r.GetErrorString = reinterpret_cast<dfx_rccl::GetErrorString_t>(sym("ncclGetErrorString"));
  if (!ok) return fail(DFX_E_HIP, r.path + " lacks an RCCL entry point this library needs");
  g_rccl = r;
  return DFX_OK;
}

int nccl_fail(const char* what, int rc) {
  return fail(DFX_E_HIP, std::string(what) + " failed: " + (g_rccl.GetErrorString ? g_rccl.GetErrorString(rc) : "?") + " (" + g_rccl.path + ")");
}

}  // namespace

struct dfx_comm {
  NcclComm comm = nullptr;
  int rank = 0, world = 1, device = -1;
};

extern "C" {

DFX_API int dfx_comm_get_unique_id(void* id_out) {
  if (!id_out) return fail(DFX_E_INVALID, "dfx_comm_get_unique_id: null argument");
  int rc;
  if ((rc = load_rccl())) return rc;
  NcclUniqueId id;
  const int e = g_rccl.GetUniqueId(&id);
  if (e != kNcclSuccess) return nccl_fail("ncclGetUniqueId", e);
  std::memcpy(id_out, &id, sizeof(id));
  return DFX_OK;
}

DFX_API int dfx_comm_create(dfx_ctx* ctx, const void* id, int rank, int world, dfx_comm** out) {
  if (!id || !out) return fail(DFX_E_INVALID, "dfx_comm_create: null argument");
  *out = nullptr;
  if (world < 1 || rank < 0 || rank >= world) return fail(DFX_E_INVALID, "rank " + std::to_string(rank) + " of world " + std::to_string(world));
  int rc;
  if ((rc = load_rccl())) return rc;
  NcclUniqueId uid;
  std::memcpy(&uid, id, sizeof(uid));
  dfx_comm* c = new dfx_comm();
  c->rank = rank; c->world = world;
  // ncclCommInitRank binds the communicator to the CALLING THREAD's current device.  dfx_internal_ctx_device() makes the context's device current
  // (hipSetDevice, like every entry point of the library that takes a context) -- the device whose streams the collectives are enqueued on --
  // whatever device the caller's thread had current; checked here rather than assumed
  c->device = ctx ? dfx_internal_ctx_device(ctx) : -1;
  if (c->device >= 0) {
    int cur = -1;
    if (hipGetDevice(&cur) != hipSuccess || cur != c->device) { delete c; return fail(DFX_E_HIP, "dfx_comm_create: device " + std::to_string(c->device) + " of the context is not current"); }
  }
  const int e = g_rccl.CommInitRank(&c->comm, world, uid, rank);
  if (e != kNcclSuccess) { delete c; return nccl_fail("ncclCommInitRank", e); }
  *out = c;
  return DFX_OK;
}

DFX_API void dfx_comm_destroy(dfx_comm* c) {
  if (!c) return;
  if (c->comm && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(c->comm);
  delete c;
}

DFX_API int dfx_comm_rank(const dfx_comm* c) { return c ? c->rank : -1; }
DFX_API int dfx_comm_world(const dfx_comm* c) { return c ? c->world : 0; }

DFX_API int dfx_shard_range(int n, int rank, int world, int* first, int* count) {
  if (!first || !count || n < 0 || world < 1 || rank < 0 || rank >= world) return fail(DFX_E_INVALID, "dfx_shard_range: bad argument");
  const int base = n / world, rem = n % world;
  *first = rank * base + (rank < rem ? rank : rem);
  *count = base + (rank < rem ? 1 : 0);
  return DFX_OK;
}

DFX_API int dfx_graph_reduce_async(dfx_ctx* ctx, dfx_comm* c, const dfx_graph* graph, float* sys_dev, int root) {
  if (!graph) return fail(DFX_E_INVALID, "dfx_graph_reduce_async: null argument");
  return dfx_comm_reduce_f32_async(ctx, c, sys_dev, dfx_graph_system_floats(graph), root);
}

DFX_API int dfx_comm_reduce_f32_async(dfx_ctx* ctx, dfx_comm* c, float* buf_dev, size_t n, int root) {
  float* const sys_dev = buf_dev;
  if (!c || !sys_dev || n == 0) return fail(DFX_E_INVALID, "dfx_comm_reduce_f32_async: null argument");
  if (root >= c->world) return fail(DFX_E_INVALID, "root " + std::to_string(root) + " outside the world of " + std::to_string(c->world));
  void* stream = ctx ? dfx_internal_exchange_stream(ctx) : nullptr;
  const int e = root < 0 ? g_rccl.AllReduce(sys_dev, sys_dev, n, kNcclFloat, kNcclSum, c->comm, stream)
                         : g_rccl.Reduce(sys_dev, sys_dev, n, kNcclFloat, kNcclSum, root, c->comm, stream);
  if (e != kNcclSuccess) return nccl_fail(root < 0 ? "ncclAllReduce" : "ncclReduce", e);
  return DFX_OK;
}

DFX_API int dfx_items_all_gather_async(dfx_ctx* ctx, dfx_comm* c, const void* items_local_dev, size_t bytes_per_rank, void* items_all_dev) {
  if (!c || !items_local_dev || !items_all_dev) return fail(DFX_E_INVALID, "dfx_items_all_gather_async: null argument");
  if (bytes_per_rank == 0) return fail(DFX_E_INVALID, "dfx_items_all_gather_async: empty contribution");
  void* stream = ctx ? dfx_internal_exchange_stream(ctx) : nullptr;
  const int e = g_rccl.AllGather(items_local_dev, items_all_dev, bytes_per_rank, kNcclUint8, c->comm, stream);
  if (e != kNcclSuccess) return nccl_fail("ncclAllGather", e);
  return DFX_OK;
}

// Replication of a new keyframe's buffers ("replicate all keyframe pyramids on every GPU ... one-time broadcast per new keyframe";
// the buffers are those of core/mapping/keyframe.h:46-56).  Unlike the exchange of a step's RESULTS, this one is ordered on the context's
// MAIN stream: it rewrites inputs of the next launches there, behind whatever still reads the old content.
DFX_API int dfx_comm_broadcast_async(dfx_ctx* ctx, dfx_comm* c, void* buf_dev, size_t bytes, int root) {
  if (!c || !buf_dev || bytes == 0) return fail(DFX_E_INVALID, "dfx_comm_broadcast_async: null argument");
  if (root < 0 || root >= c->world) return fail(DFX_E_INVALID, "root " + std::to_string(root) + " outside the world of " + std::to_string(c->world));
  void* stream = ctx ? dfx_internal_main_stream(ctx) : nullptr;
  int rc;
  if ((rc = load_rccl())) return rc;
  if (ctx && c->rank != root && (rc = dfx_internal_note_write(ctx, buf_dev))) return rc;   // a valid0 map with a shadow: the bits forget
  const int e = g_rccl.Broadcast(buf_dev, buf_dev, bytes, kNcclUint8, root, c->comm, stream);
  if (e != kNcclSuccess) return nccl_fail("ncclBroadcast", e);
  return DFX_OK;
}

}  // extern "C"
