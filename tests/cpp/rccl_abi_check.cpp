// rccl_abi_check -- pre-flight of the multi-GPU exchange that needs no second GPU: the hand-declared RCCL subset the product calls through
// dlsym (deepfactors_amd/csrc/dfx_rccl_abi.hpp) against the REAL header <rccl/rccl.h>, at compile time, and against the real library at run time.
//
//   compile time  per entry point: same number of parameters as the header's prototype, every parameter and the result ABI-equivalent position by
//                 position (pointer <-> pointer, C enum / int <-> int of the same size, the 128-byte unique id by value <-> the same), and the
//                 enumerator values dfx_comm.cpp passes (ncclSuccess, ncclSum, ncclUint8, ncclFloat) equal to the header's
//   run time      dlopen of the real librccl (argv[1], else the names dfx_comm.cpp tries) and dlsym of every entry point; with a GPU:
//                 ncclGetUniqueId + ncclCommInitRank(world 1) + ncclAllReduce of 1024 floats through the product's own typedefs
//
// This file is TEST code: it is the only translation unit of the repository that includes rccl.h.
#include <rccl/rccl.h>

#include <dlfcn.h>

#include <cstdio>
#include <cstring>
#include <type_traits>

#include "../../deepfactors_amd/csrc/dfx_rccl_abi.hpp"

namespace {

template <class Real, class Ours>
constexpr bool abi_same() {
  using R = std::remove_cv_t<Real>;
  using O = std::remove_cv_t<Ours>;
  if constexpr (std::is_pointer_v<R> || std::is_pointer_v<O>) return std::is_pointer_v<R> && std::is_pointer_v<O>;
  else if constexpr (std::is_class_v<R> || std::is_class_v<O>)
    return std::is_class_v<R> && std::is_class_v<O> && sizeof(R) == sizeof(O) && alignof(R) == alignof(O) && std::is_trivially_copyable_v<R> && std::is_trivially_copyable_v<O>;
  else return (std::is_integral_v<R> || std::is_enum_v<R>) && (std::is_integral_v<O> || std::is_enum_v<O>) && sizeof(R) == sizeof(O);
}

template <class Real, class Ours> struct abi_equiv : std::false_type {};
template <class RR, class... RA, class OR, class... OA>
struct abi_equiv<RR (*)(RA...), OR (*)(OA...)> {
  static constexpr bool arity = sizeof...(RA) == sizeof...(OA);
  template <bool B = arity> static constexpr std::enable_if_t<B, bool> args() { return (abi_same<RA, OA>() && ...); }
  template <bool B = arity> static constexpr std::enable_if_t<!B, bool> args() { return false; }
  static constexpr bool value = arity && abi_same<RR, OR>() && args();
};

// the header's prototypes, as function-pointer types
static_assert(abi_equiv<decltype(&ncclGetUniqueId), dfx_rccl::GetUniqueId_t>::value, "ncclGetUniqueId");
static_assert(abi_equiv<decltype(&ncclCommInitRank), dfx_rccl::CommInitRank_t>::value, "ncclCommInitRank");
static_assert(abi_equiv<decltype(&ncclCommDestroy), dfx_rccl::CommDestroy_t>::value, "ncclCommDestroy");
static_assert(abi_equiv<decltype(&ncclReduce), dfx_rccl::Reduce_t>::value, "ncclReduce");
static_assert(abi_equiv<decltype(&ncclAllReduce), dfx_rccl::AllReduce_t>::value, "ncclAllReduce");
static_assert(abi_equiv<decltype(&ncclAllGather), dfx_rccl::AllGather_t>::value, "ncclAllGather");
static_assert(abi_equiv<decltype(&ncclBroadcast), dfx_rccl::Broadcast_t>::value, "ncclBroadcast");
static_assert(abi_equiv<decltype(&ncclGetErrorString), dfx_rccl::GetErrorString_t>::value, "ncclGetErrorString");
// ... and the ORDER of the same-class parameters, which the equivalence above cannot see (datatype / op / root are all ints to it):
// the header's exact prototypes, spelled with its own types
static_assert(std::is_same_v<decltype(&ncclReduce), ncclResult_t (*)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, int, ncclComm_t, hipStream_t)>, "ncclReduce order");
static_assert(std::is_same_v<decltype(&ncclAllReduce), ncclResult_t (*)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t)>, "ncclAllReduce order");
static_assert(std::is_same_v<decltype(&ncclAllGather), ncclResult_t (*)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t)>, "ncclAllGather order");
static_assert(std::is_same_v<decltype(&ncclBroadcast), ncclResult_t (*)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t)>, "ncclBroadcast order");
static_assert(std::is_same_v<decltype(&ncclCommInitRank), ncclResult_t (*)(ncclComm_t*, int, ncclUniqueId, int)>, "ncclCommInitRank order");
// enumerators
static_assert(int(ncclSuccess) == dfx_rccl::kSuccess && int(ncclSum) == dfx_rccl::kSum && int(ncclUint8) == dfx_rccl::kUint8 && int(ncclFloat) == dfx_rccl::kFloat &&
              int(ncclFloat32) == dfx_rccl::kFloat, "enumerator values");
static_assert(sizeof(ncclUniqueId) == sizeof(dfx_rccl::UniqueId) && NCCL_UNIQUE_ID_BYTES == 128, "unique id");
static_assert(std::is_pointer_v<ncclComm_t> && std::is_pointer_v<hipStream_t>, "handles are pointers");

}  // namespace

int main(int argc, char** argv) {
  const char* names[] = {argc > 1 ? argv[1] : nullptr, "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  void* h = nullptr;
  const char* used = nullptr;
  for (const char* n : names) {
    if (!n || h) continue;
    h = dlopen(n, RTLD_NOW | RTLD_LOCAL);
    if (h) used = n;
  }
  if (!h) { std::printf("rccl_abi_check: compile-time checks passed; no librccl could be loaded (%s)\n", dlerror()); return 3; }
  const char* syms[] = {"ncclGetUniqueId", "ncclCommInitRank", "ncclCommDestroy", "ncclReduce", "ncclAllReduce", "ncclAllGather", "ncclBroadcast", "ncclGetErrorString"};
  for (const char* s : syms)
    if (!dlsym(h, s)) { std::printf("rccl_abi_check: %s lacks %s\n", used, s); return 1; }
  auto err = reinterpret_cast<dfx_rccl::GetErrorString_t>(dlsym(h, "ncclGetErrorString"));
  if (!err(dfx_rccl::kSuccess) || !std::strlen(err(dfx_rccl::kSuccess))) { std::printf("rccl_abi_check: ncclGetErrorString(0) is empty\n"); return 1; }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess) ndev = 0;
  if (ndev > 0) {   // a world of one through the product's own typedefs: id by value, enum values, argument order all exercised for real
    auto get_id = reinterpret_cast<dfx_rccl::GetUniqueId_t>(dlsym(h, "ncclGetUniqueId"));
    auto init = reinterpret_cast<dfx_rccl::CommInitRank_t>(dlsym(h, "ncclCommInitRank"));
    auto destroy = reinterpret_cast<dfx_rccl::CommDestroy_t>(dlsym(h, "ncclCommDestroy"));
    auto allreduce = reinterpret_cast<dfx_rccl::AllReduce_t>(dlsym(h, "ncclAllReduce"));
    auto reduce = reinterpret_cast<dfx_rccl::Reduce_t>(dlsym(h, "ncclReduce"));
    auto gather = reinterpret_cast<dfx_rccl::AllGather_t>(dlsym(h, "ncclAllGather"));
    auto bcast = reinterpret_cast<dfx_rccl::Broadcast_t>(dlsym(h, "ncclBroadcast"));
    dfx_rccl::UniqueId id;
    dfx_rccl::Comm comm = nullptr;
    int e;
    if ((e = get_id(&id)) || (e = init(&comm, 1, id, 0))) { std::printf("rccl_abi_check: init failed: %s\n", err(e)); return 1; }
    const size_t n = 1024;
    float *a = nullptr, *b = nullptr, host[n];
    for (size_t i = 0; i < n; ++i) host[i] = float(i) * 0.5f;
    hipStream_t st;
    if (hipMalloc(&a, n * 4) || hipMalloc(&b, n * 4) || hipStreamCreate(&st) || hipMemcpy(a, host, n * 4, hipMemcpyHostToDevice)) return 1;
    if ((e = allreduce(a, a, n, dfx_rccl::kFloat, dfx_rccl::kSum, comm, st)) || (e = reduce(a, a, n, dfx_rccl::kFloat, dfx_rccl::kSum, 0, comm, st)) ||
        (e = gather(a, b, n * 4, dfx_rccl::kUint8, comm, st)) || (e = bcast(b, b, n * 4, dfx_rccl::kUint8, 0, comm, st))) {
      std::printf("rccl_abi_check: collective failed: %s\n", err(e));
      return 1;
    }
    float back[n];
    if (hipStreamSynchronize(st) || hipMemcpy(back, b, n * 4, hipMemcpyDeviceToHost)) return 1;
    for (size_t i = 0; i < n; ++i)
      if (back[i] != host[i]) { std::printf("rccl_abi_check: world-1 round trip changed element %zu: %g -> %g\n", i, host[i], back[i]); return 1; }
    destroy(comm);
    (void)hipFree(a); (void)hipFree(b); (void)hipStreamDestroy(st);
  }
  std::printf("rccl_abi_check OK (%s, %d device(s)%s)\n", used, ndev, ndev > 0 ? ", world-1 collectives through the product's typedefs" : "");
  return 0;
}
