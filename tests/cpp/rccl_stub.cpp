// rccl_stub.cpp -- TEST INFRASTRUCTURE: a host-memory stand-in for the eight RCCL entry points libdfx resolves at run time
// (deepfactors_amd/csrc/dfx_comm.cpp), so that the exchange step behind the C ABI can run as two "ranks" (threads of one process) on a
// box without a GPU: tests/cpp/comm_test.cpp, loaded through DFX_RCCL_LIB.  Buffers are host pointers, streams are ignored, every
// collective is blocking (rendezvous of all ranks of the communicator).  Sums are formed in rank order, like a ring reduce would not:
// the test compares against the same order.
#include <condition_variable>
#include <cstddef>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

namespace {
struct Group {
  int n = 0;
  std::mutex mu;
  std::condition_variable cv;
  int arrived = 0, generation = 0;
  std::vector<const void*> slot;
  void barrier() {
    std::unique_lock<std::mutex> lk(mu);
    const int g = generation;
    if (++arrived == n) { arrived = 0; ++generation; cv.notify_all(); }
    else cv.wait(lk, [&] { return generation != g; });
  }
};
struct Comm { Group* g; int rank; };
std::mutex g_mu;
std::map<std::string, Group*> g_groups;
int g_ids = 0;
size_t elem(int dt) { return dt == 7 ? 4 : 1; }   // ncclFloat / ncclUint8 are all the library uses
}  // namespace

extern "C" {
struct ncclUniqueId { char internal[128]; };

int ncclGetUniqueId(ncclUniqueId* id) {
  std::lock_guard<std::mutex> lk(g_mu);
  std::memset(id, 0, sizeof(*id));
  std::snprintf(id->internal, sizeof(id->internal), "dfx-stub-%d", ++g_ids);
  return 0;
}
int ncclCommInitRank(void** comm, int nranks, ncclUniqueId id, int rank) {
  std::lock_guard<std::mutex> lk(g_mu);
  Group*& g = g_groups[std::string(id.internal, sizeof(id.internal))];
  if (!g) { g = new Group(); g->n = nranks; g->slot.assign(nranks, nullptr); }
  if (g->n != nranks || rank < 0 || rank >= nranks) return 4;   // ncclInvalidArgument
  *comm = new Comm{ g, rank };
  return 0;
}
int ncclCommDestroy(void* comm) { delete static_cast<Comm*>(comm); return 0; }
const char* ncclGetErrorString(int rc) { return rc == 0 ? "no error" : "stub error"; }

static int reduce_impl(const void* send, void* recv, size_t count, int dt, int root, void* comm) {
  if (dt != 7) return 4;
  Comm* c = static_cast<Comm*>(comm);
  Group* g = c->g;
  g->slot[c->rank] = send;
  g->barrier();
  std::vector<float> tmp;
  if (root < 0 || c->rank == root) {
    tmp.assign(count, 0.f);
    for (int r = 0; r < g->n; ++r) { const float* p = static_cast<const float*>(g->slot[r]); for (size_t i = 0; i < count; ++i) tmp[i] += p[i]; }
  }
  g->barrier();   // every contribution has been read: in-place results may be written now
  if (!tmp.empty()) std::memcpy(recv, tmp.data(), count * sizeof(float));
  g->barrier();
  return 0;
}
int ncclReduce(const void* send, void* recv, size_t count, int dt, int op, int root, void* comm, void*) { return op == 0 ? reduce_impl(send, recv, count, dt, root, comm) : 4; }
int ncclAllReduce(const void* send, void* recv, size_t count, int dt, int op, void* comm, void*) { return op == 0 ? reduce_impl(send, recv, count, dt, -1, comm) : 4; }
int ncclBroadcast(const void* send, void* recv, size_t count, int dt, int root, void* comm, void*) {
  Comm* c = static_cast<Comm*>(comm);
  Group* g = c->g;
  if (root < 0 || root >= g->n) return 4;
  g->slot[c->rank] = send;
  g->barrier();
  if (c->rank != root) std::memcpy(recv, g->slot[root], count * elem(dt));
  g->barrier();
  return 0;
}
int ncclAllGather(const void* send, void* recv, size_t sendcount, int dt, void* comm, void*) {
  Comm* c = static_cast<Comm*>(comm);
  Group* g = c->g;
  g->slot[c->rank] = send;
  g->barrier();
  const size_t bytes = sendcount * elem(dt);
  for (int r = 0; r < g->n; ++r) std::memcpy(static_cast<char*>(recv) + (size_t)r * bytes, g->slot[r], bytes);
  g->barrier();
  return 0;
}
}
