// dfx_rccl_abi.hpp -- the subset of rccl.h that dfx_comm.cpp calls through dlsym, declared by hand so that libdfx.so carries no build- or
// link-time dependency on RCCL (ROCm 7.x: /opt/rocm/include/rccl/rccl.h:40-43,52,187,220,260,339,448-466,550,591,611,678).
//
// Hand-written declarations can drift from the header they stand for.  tests/cpp/rccl_abi_check.cpp includes BOTH this file and the real
// <rccl/rccl.h> and static_asserts, per entry point, that every parameter of the prototype below is ABI-equivalent to the header's (same
// position, size and class: pointer / integer-or-enum / the 128-byte id by value) and that the enumerator values are the header's; then it
// dlopens the real librccl and resolves each symbol.  It is built by __graft_entry__.build() and run by the CPU and the GPU test suites.
#pragma once
#include <cstddef>

namespace dfx_rccl {

struct UniqueId { char internal[128]; };   // ncclUniqueId
typedef void* Comm;                        // ncclComm_t
typedef void* Stream;                      // hipStream_t
enum : int { kSuccess = 0 /* ncclSuccess */, kSum = 0 /* ncclSum */, kUint8 = 1 /* ncclUint8 */, kFloat = 7 /* ncclFloat */ };

// result, datatype, reduction operator and rank arguments are C enums / ints in rccl.h: passed as int
typedef int (*GetUniqueId_t)(UniqueId*);
typedef int (*CommInitRank_t)(Comm*, int nranks, UniqueId id, int rank);
typedef int (*CommDestroy_t)(Comm);
typedef int (*Reduce_t)(const void* send, void* recv, size_t count, int datatype, int op, int root, Comm, Stream);
typedef int (*AllReduce_t)(const void* send, void* recv, size_t count, int datatype, int op, Comm, Stream);
typedef int (*AllGather_t)(const void* send, void* recv, size_t sendcount, int datatype, Comm, Stream);
typedef int (*Broadcast_t)(const void* send, void* recv, size_t count, int datatype, int root, Comm, Stream);
typedef const char* (*GetErrorString_t)(int);

}  // namespace dfx_rccl
