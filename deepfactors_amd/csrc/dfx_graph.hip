// dfx_graph.hip -- Gauss-Newton normal equations of a keyframe graph (the exchange buffer of the multi-GPU path).
//
// The reference hands each pair's 44x44 system to its own gtsam::HessianFactor, keyed by (pose0, pose1, code0)
// (core/gtsam/photometric_factor.cpp:105-180), and lets iSAM2 add the factors up; its graph links arbitrary
// keyframe -> frame pairs, in both directions (core/mapping/mapper.cpp:308-311).  Here the factors are summed on the
// device into a block-sparse system over N nodes (node = keyframe or frame: pose 6 + code CS unknowns, D = 6 + CS):
//     Hd[n]  D x D   diagonal block of node n: (pose0 | code0) x (pose0 | code0) of every pair whose keyframe is n
//                    (G11, G13, G33) plus the 6x6 (pose1, pose1) = G22 of every pair whose frame is n
//     Ho[p]  D x 6   off-diagonal block of pair p: rows = (pose | code) of its keyframe node, columns = pose of its frame
//                    node (G12 over G23^T); single writer
//     g[n]   D       g1 / g3 of the pairs out of n, g2 of the pairs into n          (Jtr as stored; the factor negates it)
// Assembly is a GATHER in fixed order (one workgroup per node walks the node's incident pairs in ascending pair index, sums
// in double, writes once): no atomics, no pre-zeroed buffer, the same bits for any launch shape, and -- when every rank holds
// every item (gather mode) -- for any world size.  A rank that only holds the items of pairs [first, first + n) writes the
// contribution of exactly these pairs (zeros elsewhere); the ranks' buffers are then summed by one RCCL reduce.
#include "dfx_kernels.hpp"

namespace dfx {

// packed upper-triangular index of entry (a, b) of an NP x NP symmetric matrix stored row-major over the upper triangle
__device__ __forceinline__ int tri_index(int a, int b, int NP) {
  const int lo = a < b ? a : b, hi = a < b ? b : a;
  return lo * NP - lo * (lo - 1) / 2 + (hi - lo);
}

// grid = n_nodes + n_pairs workgroups of 256 threads
template <int CS>
__global__ __launch_bounds__(256) void k_graph_assemble(const GraphDev G, const char* __restrict__ items, const size_t item_stride, const int first_pair,
                                                        const int n_local, float* __restrict__ sys) {
  constexpr int NP = 12 + CS, D = 6 + CS, NT = NP * (NP + 1) / 2;
  float* const Hd = sys;
  float* const Ho = sys + (size_t)G.n_nodes * D * D;
  float* const gv = Ho + (size_t)G.n_pairs * D * 6;
  auto item_of = [&](int p) -> const float* {   // null when pair p is not held by this rank
    const int l = p - first_pair;
    return (l >= 0 && l < n_local) ? reinterpret_cast<const float*>(items + (size_t)l * item_stride) : nullptr;
  };
  if ((int)blockIdx.x < G.n_nodes) {
    const int n = blockIdx.x;
    const int k0 = G.kf_begin[n], k1 = G.kf_begin[n + 1];   // pairs whose keyframe is n (ascending)
    const int f0 = G.fr_begin[n], f1 = G.fr_begin[n + 1];   // pairs whose frame is n (ascending)
    for (int e = threadIdx.x; e < D * D + D; e += 256) {
      double acc = 0.0;
      if (e < D * D) {
        const int r = e / D, c = e - r * D;
        const int ia = r < 6 ? r : r + 6, ib = c < 6 ? c : c + 6;   // node-local -> item parameter (pose0 0..5, code0 12..)
        const int t0 = tri_index(ia, ib, NP);
        for (int q = k0; q < k1; ++q) { const float* it = item_of(G.kf_pairs[q]); if (it) acc += (double)it[t0]; }
        if (r < 6 && c < 6) {
          const int t1 = tri_index(6 + r, 6 + c, NP);
          for (int q = f0; q < f1; ++q) { const float* it = item_of(G.fr_pairs[q]); if (it) acc += (double)it[t1]; }
        }
        Hd[(size_t)n * D * D + e] = (float)acc;
      } else {
        const int r = e - D * D;
        const int ia = r < 6 ? r : r + 6;
        for (int q = k0; q < k1; ++q) { const float* it = item_of(G.kf_pairs[q]); if (it) acc += (double)it[NT + ia]; }
        if (r < 6) for (int q = f0; q < f1; ++q) { const float* it = item_of(G.fr_pairs[q]); if (it) acc += (double)it[NT + 6 + r]; }
        gv[(size_t)n * D + r] = (float)acc;
      }
    }
  } else {
    const int p = blockIdx.x - G.n_nodes;
    const float* it = item_of(p);
    for (int e = threadIdx.x; e < D * 6; e += 256) {
      const int r = e / 6, c = e - r * 6;
      const int ia = r < 6 ? r : r + 6;
      Ho[(size_t)p * D * 6 + e] = it ? it[tri_index(ia, 6 + c, NP)] : 0.0f;
    }
  }
}

hipError_t launch_graph_assemble(int cs, const GraphDev& G, const void* items_dev, size_t item_stride, int first_pair, int n_local, float* sys_dev,
                                 hipStream_t stream) {
  const dim3 grid(G.n_nodes + G.n_pairs), block(256);
  switch (cs) {
    case 16: hipLaunchKernelGGL(k_graph_assemble<16>, grid, block, 0, stream, G, (const char*)items_dev, item_stride, first_pair, n_local, sys_dev); break;
    case 32: hipLaunchKernelGGL(k_graph_assemble<32>, grid, block, 0, stream, G, (const char*)items_dev, item_stride, first_pair, n_local, sys_dev); break;
    case 64: hipLaunchKernelGGL(k_graph_assemble<64>, grid, block, 0, stream, G, (const char*)items_dev, item_stride, first_pair, n_local, sys_dev); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

}  // namespace dfx
