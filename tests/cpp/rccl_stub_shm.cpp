// rccl_stub_shm.cpp -- TEST INFRASTRUCTURE: the host-memory stand-in of rccl_stub.cpp for ranks that are PROCESSES (tests/test_bench_fake_main.py runs
// bench.py's main() on two gloo ranks without a GPU, with the shipped C-ABI exchange -- dfx_comm_* -- underneath): the communicator is a POSIX
// shared-memory segment named by the unique id; every rank copies its contribution into its slot, a sense-reversing barrier on atomics in the
// segment collects the ranks, the root sums the slots in rank order.  Buffers are host pointers, streams are ignored, collectives block.
// Loaded through DFX_RCCL_LIB like the thread stub.
#include <fcntl.h>
#include <sched.h>
#include <sys/mman.h>
#include <unistd.h>

#include <atomic>
#include <cstddef>
#include <cstdio>
#include <cstring>

namespace {
constexpr size_t kSlot = 32u << 20;   // bytes per rank (sparse: untouched pages cost nothing)
constexpr int kMaxRanks = 8;
struct Hdr { std::atomic<int> arrived, generation; };
struct Comm { char name[128]; Hdr* hdr; char* data; int rank, n; size_t bytes; };
std::atomic<int> g_ids{ 0 };
size_t elem(int dt) { return dt == 7 ? 4 : 1; }
void barrier(Comm* c) {
  const int g = c->hdr->generation.load();
  if (c->hdr->arrived.fetch_add(1) + 1 == c->n) { c->hdr->arrived.store(0); c->hdr->generation.fetch_add(1); }
  else while (c->hdr->generation.load() == g) sched_yield();
}
}  // namespace

extern "C" {
struct ncclUniqueId { char internal[128]; };

int ncclGetUniqueId(ncclUniqueId* id) {
  std::memset(id, 0, sizeof(*id));
  std::snprintf(id->internal, sizeof(id->internal), "/dfx-stub-%d-%d", (int)getpid(), ++g_ids);
  return 0;
}
int ncclCommInitRank(void** comm, int nranks, ncclUniqueId id, int rank) {
  if (nranks < 1 || nranks > kMaxRanks || rank < 0 || rank >= nranks) return 4;
  Comm* c = new Comm();
  std::memcpy(c->name, id.internal, sizeof(c->name));
  c->name[sizeof(c->name) - 1] = 0;
  c->rank = rank; c->n = nranks; c->bytes = 4096 + kSlot * (size_t)nranks;
  const int fd = shm_open(c->name, O_CREAT | O_RDWR, 0600);
  if (fd < 0 || ftruncate(fd, (off_t)c->bytes) != 0) { delete c; return 2; }
  void* p = mmap(nullptr, c->bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  close(fd);
  if (p == MAP_FAILED) { delete c; return 2; }
  c->hdr = static_cast<Hdr*>(p); c->data = static_cast<char*>(p) + 4096;
  *comm = c;
  barrier(c);   // like ncclCommInitRank: returns once every rank is there
  return 0;
}
int ncclCommDestroy(void* comm) {
  Comm* c = static_cast<Comm*>(comm);
  barrier(c);
  munmap(c->hdr, c->bytes);
  if (c->rank == 0) shm_unlink(c->name);
  delete c;
  return 0;
}
const char* ncclGetErrorString(int rc) { return rc == 0 ? "no error" : "stub error"; }

static int reduce_impl(const void* send, void* recv, size_t count, int dt, int root, void* comm) {
  Comm* c = static_cast<Comm*>(comm);
  if (dt != 7 || count * 4 > kSlot) return 4;
  std::memcpy(c->data + kSlot * (size_t)c->rank, send, count * 4);
  barrier(c);
  if (root < 0 || c->rank == root) {
    float* out = static_cast<float*>(recv);
    for (size_t i = 0; i < count; ++i) {
      float s = 0.f;
      for (int r = 0; r < c->n; ++r) s += reinterpret_cast<const float*>(c->data + kSlot * (size_t)r)[i];
      out[i] = s;
    }
  }
  barrier(c);   // every slot has been read: the next collective may overwrite them
  return 0;
}
int ncclReduce(const void* send, void* recv, size_t count, int dt, int op, int root, void* comm, void*) { return op == 0 ? reduce_impl(send, recv, count, dt, root, comm) : 4; }
int ncclAllReduce(const void* send, void* recv, size_t count, int dt, int op, void* comm, void*) { return op == 0 ? reduce_impl(send, recv, count, dt, -1, comm) : 4; }
int ncclBroadcast(const void* send, void* recv, size_t count, int dt, int root, void* comm, void*) {
  Comm* c = static_cast<Comm*>(comm);
  const size_t bytes = count * elem(dt);
  if (root < 0 || root >= c->n || bytes > kSlot) return 4;
  if (c->rank == root) std::memcpy(c->data, send, bytes);
  barrier(c);
  if (c->rank != root) std::memcpy(recv, c->data, bytes);
  barrier(c);
  return 0;
}
int ncclAllGather(const void* send, void* recv, size_t sendcount, int dt, void* comm, void*) {
  Comm* c = static_cast<Comm*>(comm);
  const size_t bytes = sendcount * elem(dt);
  if (bytes > kSlot) return 4;
  std::memcpy(c->data + kSlot * (size_t)c->rank, send, bytes);
  barrier(c);
  for (int r = 0; r < c->n; ++r) std::memcpy(static_cast<char*>(recv) + (size_t)r * bytes, c->data + kSlot * (size_t)r, bytes);
  barrier(c);
  return 0;
}
}
