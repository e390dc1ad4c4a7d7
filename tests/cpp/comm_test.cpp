// comm_test.cpp -- the exchange step of the multi-GPU path through the C ABI (include/dfx.h: dfx_comm_*, dfx_shard_range,
// dfx_comm_reduce_f32_async, dfx_items_all_gather_async), world size 2, WITHOUT a GPU: the two ranks are threads, and libdfx resolves its
// RCCL entry points from tests/cpp/librccl_stub.so (DFX_RCCL_LIB; host-memory collectives).  What a C++ mapper does per Gauss-Newton
// round is modelled end to end on synthetic items: shard the pair list, "evaluate" the own shard, (a) gather mode -- all-gather the padded
// shards and check that every rank holds every pair's item in pair order, (b) reduce mode -- scatter-add the own items into a flat system
// and reduce onto rank 0 / onto all ranks, and check the sum against the single-process result.  Exit code 0 = pass.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include "../../include/dfx.h"

#define REQUIRE(c) do { if (!(c)) { std::printf("FAILED %s:%d: %s  [%s]\n", __FILE__, __LINE__, #c, dfx_last_error()); std::exit(1); } } while (0)

static const int kPairs = 37, kWorld = 2, kItemFloats = 11, kNodes = 9;

static void fake_item(int pair, float* it) { for (int i = 0; i < kItemFloats; ++i) it[i] = (float)std::sin(0.37 * pair + 1.3 * i) * (1 + pair % 5); }
// "assembly": pair p adds its item to node p % kNodes and node (p + 1) % kNodes of a flat [kNodes][kItemFloats] system
static void scatter(int pair, const float* it, std::vector<float>& sys) {
  for (int i = 0; i < kItemFloats; ++i) { sys[(pair % kNodes) * kItemFloats + i] += it[i]; sys[((pair + 1) % kNodes) * kItemFloats + i] += 0.5f * it[i]; }
}

static void rank_main(int rank, const unsigned char* id, std::vector<std::vector<float>>* reduced, std::vector<std::vector<float>>* gathered) {
  dfx_comm* comm = nullptr;
  REQUIRE(dfx_comm_create(nullptr, id, rank, kWorld, &comm) == DFX_OK);
  REQUIRE(dfx_comm_rank(comm) == rank && dfx_comm_world(comm) == kWorld);
  int first, count;
  REQUIRE(dfx_shard_range(kPairs, rank, kWorld, &first, &count) == DFX_OK);
  // ---- gather mode: shards padded to the largest shard
  const int per = (kPairs + kWorld - 1) / kWorld;
  std::vector<float> mine((size_t)per * kItemFloats, 0.f), all((size_t)per * kWorld * kItemFloats, -1.f);
  for (int l = 0; l < count; ++l) fake_item(first + l, &mine[(size_t)l * kItemFloats]);
  REQUIRE(dfx_items_all_gather_async(nullptr, comm, mine.data(), mine.size() * sizeof(float), all.data()) == DFX_OK);
  (*gathered)[rank] = all;
  // ---- reduce mode: onto rank 0, then onto every rank
  std::vector<float> sys((size_t)kNodes * kItemFloats, 0.f);
  for (int l = 0; l < count; ++l) scatter(first + l, &mine[(size_t)l * kItemFloats], sys);
  std::vector<float> sys2 = sys;
  REQUIRE(dfx_comm_reduce_f32_async(nullptr, comm, sys.data(), sys.size(), 0) == DFX_OK);
  REQUIRE(dfx_comm_reduce_f32_async(nullptr, comm, sys2.data(), sys2.size(), -1) == DFX_OK);
  (*reduced)[rank] = sys;
  (*reduced)[kWorld + rank] = sys2;
  REQUIRE(dfx_comm_reduce_f32_async(nullptr, comm, sys.data(), sys.size(), kWorld) == DFX_E_INVALID);   // root outside the world
  // ---- keyframe replication: rank 1's buffer lands on every rank, bytes and all
  std::vector<unsigned char> kfbuf(1000 + 37);
  for (size_t i = 0; i < kfbuf.size(); ++i) kfbuf[i] = (unsigned char)((rank * 131 + i * 7) & 0xff);
  REQUIRE(dfx_comm_broadcast_async(nullptr, comm, kfbuf.data(), kfbuf.size(), 1) == DFX_OK);
  for (size_t i = 0; i < kfbuf.size(); ++i) REQUIRE(kfbuf[i] == (unsigned char)((1 * 131 + i * 7) & 0xff));
  REQUIRE(dfx_comm_broadcast_async(nullptr, comm, kfbuf.data(), kfbuf.size(), kWorld) == DFX_E_INVALID);
  REQUIRE(dfx_comm_broadcast_async(nullptr, comm, kfbuf.data(), 0, 0) == DFX_E_INVALID);
  dfx_comm_destroy(comm);
}

int main() {
  // shards: contiguous, disjoint, complete, sizes differ by at most one -- for a few shapes
  for (int n : { 0, 1, 5, 37, 1024 })
    for (int w : { 1, 2, 3, 8 }) {
      int next = 0, lo = n, hi = 0;
      for (int r = 0; r < w; ++r) {
        int f, c;
        REQUIRE(dfx_shard_range(n, r, w, &f, &c) == DFX_OK && f == next && c >= 0);
        next = f + c; lo = c < lo ? c : lo; hi = c > hi ? c : hi;
      }
      REQUIRE(next == n && hi - lo <= 1);
    }
  int f, c;
  REQUIRE(dfx_shard_range(5, 2, 2, &f, &c) == DFX_E_INVALID);
  unsigned char id[DFX_COMM_ID_BYTES];
  REQUIRE(dfx_comm_get_unique_id(id) == DFX_OK);
  std::vector<std::vector<float>> reduced(2 * kWorld), gathered(kWorld);
  std::vector<std::thread> th;
  for (int r = 0; r < kWorld; ++r) th.emplace_back(rank_main, r, id, &reduced, &gathered);
  for (auto& t : th) t.join();
  // single-process truth
  std::vector<float> want((size_t)kNodes * kItemFloats, 0.f), part[kWorld];
  const int per = (kPairs + kWorld - 1) / kWorld;
  for (int r = 0; r < kWorld; ++r) {
    part[r].assign(want.size(), 0.f);
    int first, count;
    dfx_shard_range(kPairs, r, kWorld, &first, &count);
    for (int l = 0; l < count; ++l) { float it[kItemFloats]; fake_item(first + l, it); scatter(first + l, it, part[r]); }
  }
  for (size_t i = 0; i < want.size(); ++i) want[i] = part[0][i] + part[1][i];   // the stub sums in rank order
  REQUIRE(reduced[0] == want);                                  // reduce onto rank 0 ...
  REQUIRE(reduced[1] == part[1]);                               // ... leaves the other rank's buffer as it was
  REQUIRE(reduced[kWorld + 0] == want && reduced[kWorld + 1] == want);   // all-reduce: every rank
  for (int r = 0; r < kWorld; ++r) {
    for (int p = 0; p < kPairs; ++p) {
      int owner = 0, first = 0, count = 0;
      for (owner = 0; owner < kWorld; ++owner) { dfx_shard_range(kPairs, owner, kWorld, &first, &count); if (p >= first && p < first + count) break; }
      float it[kItemFloats];
      fake_item(p, it);
      REQUIRE(std::memcmp(&gathered[r][((size_t)owner * per + (p - first)) * kItemFloats], it, sizeof(it)) == 0);
    }
  }
  std::printf("comm_test OK (%d pairs over %d ranks: gather, reduce, all-reduce, broadcast)\n", kPairs, kWorld);
  return 0;
}
