// dfx_misc_kernels.hip -- the HBM/latency-bound siblings of the SfM step for gfx950:
//   SE3Aligner::RunStep / Warp   (reference sources/cuda/cu_se3aligner.cpp:37-176, lucas_kanade_se3.h:41-77)
//   SfmAligner::EvaluateError    (cu_sfmaligner.cpp:72-147, dense_sfm.h:79-119)
//   UpdateDepth, SobelGradients, GaussianBlurDown, SquaredError (cu_image_proc.cpp:57-277)
//   DepthAligner::RunStep        (cu_depthaligner.cpp:32-110)
// All reductions: lane = pixel, grid-stride over 64-pixel chunks, wave shuffle (64 wide) -> LDS across the
// 4 waves in fixed order -> one 32-float partial per workgroup -> k_finalize_rows sums the partials in double
// in fixed order.  Inlier counts travel as exact floats (< 2^24 per workgroup) and are summed in double.
#include "dfx_device.hpp"
#include "dfx_kernels.hpp"

namespace dfx {

constexpr int kT = 256;   // threads per workgroup (4 waves)

template <int N>
__device__ __forceinline__ void block_reduce_store(float (&v)[N], float* __restrict__ out_row) {
  static_assert(N <= kSimpleRow, "partial row too small");
  __shared__ float red[kT / 64][kSimpleRow];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int q = 0; q < N; ++q) {
    const float s = wave_sum(v[q]);
    if (lane == 0) red[wave][q] = s;
  }
  __syncthreads();
  if (threadIdx.x < kSimpleRow) {
    float s = 0.f;
    if (threadIdx.x < N) s = ((red[0][threadIdx.x] + red[1][threadIdx.x]) + red[2][threadIdx.x]) + red[3][threadIdx.x];
    out_row[threadIdx.x] = s;
  }
}

__device__ __forceinline__ Geo geo_from(const SimplePairDev& p) {
  Geo g;
#pragma unroll
  for (int q = 0; q < 9; ++q) g.R[q] = p.R[q];
  g.t[0] = p.t[0]; g.t[1] = p.t[1]; g.t[2] = p.t[2];
  g.fx = p.fx; g.fy = p.fy; g.u0 = p.u0; g.v0 = p.v0; g.w = p.w; g.h = p.h;
  return g;
}

// ---- the per-thread pixel walk shared by the reductions below -------------------------------------------------------
// Thread t of workgroup b visits pixels b * kT + t, + gridDim.x * kT, ...: (x, y) advance by a fixed step (no division per pixel), the
// depth / intensity of the NEXT pixel are loaded before the current one is processed (their latency hides under the warp + taps + sums
// of the current pixel), and K^-1 (x, y, 1) comes from the per-camera ray table staged in LDS (the same IEEE expressions the reference
// evaluates per pixel, pinhole_camera_impl.h:77-86, evaluated once per column / row on the host: dfx_api.cpp ray_table) instead of two
// IEEE divisions per pixel.  These operators are vector-ALU bound (exact-IEEE geometry: ~150-200 instructions per pixel), not HBM bound.
struct RayLds {
  const float* tab;   // LDS: [W] x rays, [H] y rays; null -> compute
  int W;
};
template <bool TAB>
__device__ __forceinline__ RayLds stage_ray_table(const float* __restrict__ ray_tab, float* lds_tab, int W, int H) {
  if (!TAB) return RayLds{ nullptr, W };
  for (int e = threadIdx.x; e < W + H; e += kT) lds_tab[e] = ray_tab[e];
  __syncthreads();
  return RayLds{ lds_tab, W };
}
constexpr int kRayLdsMax = 4096;   // floats of LDS a simple kernel spends on the table (W + H <= 4096; larger images compute the rays)

// Two software-pipeline stages per pixel: `issue` (the warp, then the bilinear tap LOADS of the pixel -- branch-free: a pixel without
// correspondence reads taps at (0, 0), in range and never used) runs one pixel AHEAD of `consume` (interpolation, Jacobian row, sums), so
// the taps' memory round trip -- a dependent load behind ~60 vector-ALU instructions of geometry -- hides under the arithmetic of the
// pixel before instead of stalling every iteration (round 3: 128 pairs of 640x480 262 -> see DESIGN.md 3.4; these operators were
// latency-bound at one round trip per pixel and wave, not HBM- or ALU-bound).
#ifndef DFX_WALK_PIPELINED
#define DFX_WALK_PIPELINED 1
#endif
struct TapLoads {       // the four taps of img1 [and of grad1] of one pixel, possibly still in flight
  f32x2_u ia, ib;       // img1 rows iy, iy + 1: (x, x + 1)
  f32x4_u8 ga, gb;      // grad1 rows iy, iy + 1: (gx, gy)(x), (gx, gy)(x + 1)
  float ax, ay;
};
template <bool GRAD>
__device__ __forceinline__ TapLoads issue_taps(const ImgRef& I1, const ImgRef& G1, const Corr& c) {
  const Taps tp = make_taps(c.u, c.v);
  const int ix = c.valid ? tp.ix : 0, iy = c.valid ? tp.iy : 0;
  TapLoads t;
  t.ax = tp.ax; t.ay = tp.ay;
  const char* r0 = I1.rowb(iy) + (size_t)ix * 4;
  t.ia = gload<f32x2_u>(r0);
  t.ib = gload<f32x2_u>(r0 + I1.pitch);
  if (GRAD) {
    const char* q0 = G1.rowb(iy) + (size_t)ix * 8;
    t.ga = gload<f32x4_u8>(q0);
    t.gb = gload<f32x4_u8>(q0 + G1.pitch);
  }
  return t;
}
__device__ __forceinline__ float taps_img(const TapLoads& t) { return lerp1(lerp1(t.ia.x, t.ia.y, t.ax), lerp1(t.ib.x, t.ib.y, t.ax), t.ay); }
__device__ __forceinline__ void taps_grad(const TapLoads& t, float& gx, float& gy) {
  gx = lerp1(lerp1(t.ga.x, t.ga.z, t.ax), lerp1(t.gb.x, t.gb.z, t.ax), t.ay);
  gy = lerp1(lerp1(t.ga.y, t.ga.w, t.ax), lerp1(t.gb.y, t.gb.w, t.ax), t.ay);
}

template <bool TAB, bool GRAD, typename F>
__device__ __forceinline__ void walk_pixels(const Geo& g, const SimplePairDev& p, const RayLds& rt, const int W, const int H, const float border,
                                            const float min_dpt, F&& consume /* (d, i0, const Corr&, const TapLoads&) */) {
  const ImgRef I0{ (const char*)p.img0, p.pitch_img0 }, D0{ (const char*)p.dpt0, p.pitch_dpt0 };
  const ImgRef I1{ (const char*)p.img1, p.pitch_img1 }, G1{ (const char*)p.grad1, p.pitch_grad1 };
  const unsigned npx = (unsigned)W * (unsigned)H;
  const unsigned step = gridDim.x * kT;
  unsigned i = blockIdx.x * kT + threadIdx.x;
  if (i >= npx) return;
  int y = (int)(i / (unsigned)W), x = (int)(i - (unsigned)y * (unsigned)W);
  const int sdy = (int)(step / (unsigned)W), sdx = (int)(step - (unsigned)sdy * (unsigned)W);
  auto corr = [&](int cx, int cy, float cd) {
    return TAB ? find_correspondence_ray<true>(g, rt.tab[cx], rt.tab[rt.W + cy], cd, border, min_dpt) : find_correspondence<true>(g, cx, cy, cd, border, min_dpt);
  };
  // the pixel after (ci, cx, cy) if `have` and there is one -- else the same pixel again (a harmless repeat: in range, never consumed)
  auto advance = [&](bool have, unsigned ci, int cx, int cy, unsigned& ni, int& nx, int& ny) -> bool {
    const bool more = have && ci + step < npx;
    int tx = cx + sdx, ty = cy + sdy;
    if (tx >= W) { tx -= W; ++ty; }
    ni = more ? ci + step : ci; nx = more ? tx : cx; ny = more ? ty : cy;
    return more;
  };
  float d = D0.at(x, y), i0 = I0.at(x, y);
  // (the SE3 step -- 184 vector-ALU instructions per pixel, its SIMDs 86 % busy with them -- gains nothing from the second stage and pays
  // for its state: 248 vs 256 us per 128 pairs; EvaluateError, 102 instructions per pixel: 155 -> 129 us.  profiles/r03_small_ops.txt)
  if constexpr (DFX_WALK_PIPELINED && !GRAD) {
  // pixel 0: geometry done, taps issued; pixel 1: depth / intensity in flight.  The loop is written out twice with the two pixel states
  // (A, B) swapping roles, so that no state is copied from "next" to "current" (the copies were 47 of the SE3 loop's 177 instructions).
  unsigned i1; int x1, y1;
  bool has1 = advance(true, i, x, y, i1, x1, y1);
  float dB = D0.at(x1, y1), i0B = I0.at(x1, y1);
  float dA = d, i0A = i0;
  Corr cA = corr(x, y, dA), cB;
  TapLoads tA = issue_taps<GRAD>(I1, G1, cA), tB;
  while (true) {
    unsigned i2; int x2, y2;
    const bool has2 = advance(has1, i1, x1, y1, i2, x2, y2);
    const float dN = D0.at(x2, y2), i0N = I0.at(x2, y2);   // depth / intensity of the pixel after next
    cB = corr(x1, y1, dB);                                  // stage 1 of the next pixel (B)
    tB = issue_taps<GRAD>(I1, G1, cB);
    consume(dA, i0A, cA, tA);                               // stage 2 of the current one (A)
    if (!has1) break;
    // second half: B is current, A takes the pixel after
    unsigned i3; int x3, y3;
    const bool has3 = advance(has2, i2, x2, y2, i3, x3, y3);
    dA = dN; i0A = i0N;
    const float dM = D0.at(x3, y3), i0M = I0.at(x3, y3);
    cA = corr(x2, y2, dA);
    tA = issue_taps<GRAD>(I1, G1, cA);
    consume(dB, i0B, cB, tB);
    if (!has2) break;
    dB = dM; i0B = i0M;
    i1 = i3; x1 = x3; y1 = y3; has1 = has3;
  }
  } else {
  while (true) {
    unsigned in; int xn, yn;
    const bool more = advance(true, i, x, y, in, xn, yn);
    const float dn = D0.at(xn, yn), i0n = I0.at(xn, yn);
    const Corr c = corr(x, y, d);
    const TapLoads t = issue_taps<GRAD>(I1, G1, c);
    consume(d, i0, c, t);
    if (!more) break;
    i = in; x = xn; y = yn; d = dn; i0 = i0n;
  }
  }
}

// ---- SE3 step: 21 JtJ + 6 Jtr + r^2 + inliers = 29 floats per lane -----------------------------------------
// Per-lane sums of one SE3 Gauss-Newton step over this thread's pixels (lucas_kanade_se3.h:41-77): shared by the blocking
// operator and by the device-resident tracker.
template <bool TAB>
__device__ __forceinline__ void se3_accumulate(const Geo& g, const SimplePairDev& p, const RayLds& rt, const int W, const int H, const float huber_delta,
                                               float (&acc)[29]) {

#pragma unroll
  for (int q = 0; q < 29; ++q) acc[q] = 0.f;
  walk_pixels<TAB, true>(g, p, rt, W, H, 1.0f, 0.0f, [&](float d, float i0, const Corr& c, const TapLoads& t) {
    if (c.valid) {
      float gx, gy;
      taps_grad(t, gx, gy);
      const float samp = taps_img(t);
      float J[6], D00, D02, D11, D12;
      pose_row(g, c, d, gx, gy, J, D00, D02, D11, D12);
      float r = i0 - samp;
      const float wgt = huber_weight(r, huber_delta);
      r *= wgt;
#pragma unroll
      for (int j = 0; j < 6; ++j) J[j] *= wgt;
      int k = 0;
#pragma unroll
      for (int a = 0; a < 6; ++a)
#pragma unroll
        for (int b = a; b < 6; ++b) acc[k++] += J[a] * J[b];
#pragma unroll
      for (int a = 0; a < 6; ++a) acc[21 + a] += J[a] * r;
      acc[27] += r * r;
      acc[28] += 1.0f;
    }
  });
}

template <bool TAB>
__global__ __launch_bounds__(kT) void k_se3_step(const SimplePairDev p, const int W, const int H, const float huber_delta,
                                                 float* __restrict__ partials) {
  __shared__ float ray_lds[TAB ? kRayLdsMax : 1];
  const Geo g = geo_from(p);
  const RayLds rt = stage_ray_table<TAB>(p.ray_tab, ray_lds, W, H);
  float acc[29];
  se3_accumulate<TAB>(g, p, rt, W, H, huber_delta, acc);
  block_reduce_store<29>(acc, partials + (size_t)blockIdx.x * kSimpleRow);
}

// ---- device-resident tracker (CameraTracker::TrackFrame, reference core/system/camera_tracker.cpp:42-71) -------------
// The pose lives in device memory between iterations: k_se3_step_dev reads it, k_track_update folds the workgroup
// partials (double, fixed order), solves the 6x6 normal equations by LDL^T in double and applies the reference's update
// (t += dt, R = exp(dw) R; lucas_kanade_se3.h:85-95).  A whole coarse-to-fine schedule is enqueued without host syncs.
struct TrackState {      // device-resident
  double R[9], t[3];     // pose_ck, kept in double across iterations
  float Rf[9], tf[3];    // fp32 copy consumed by the step kernel
  float last_residual; float last_inliers; int solver_failures; int iterations_done;
};

// blockIdx.y = candidate: Relocalize / the loop-closure geometry checks track ONE live frame against N keyframes
// (deepfactors.cpp:713-743, loop_detector.cpp:146-167); descriptors, states and partials are arrays over candidates.
template <bool TAB>
__global__ __launch_bounds__(kT) void k_se3_step_dev(const SimplePairDev* __restrict__ descs, const TrackState* __restrict__ states, const int W,
                                                     const int H, const float huber_delta, float* __restrict__ partials_all) {
  const SimplePairDev& p = descs[blockIdx.y];
  const TrackState* st = states + blockIdx.y;
  float* partials = partials_all + (size_t)blockIdx.y * gridDim.x * kSimpleRow;
  Geo g = geo_from(p);
#pragma unroll
  for (int q = 0; q < 9; ++q) g.R[q] = st->Rf[q];
  g.t[0] = st->tf[0]; g.t[1] = st->tf[1]; g.t[2] = st->tf[2];
  __shared__ float ray_lds[TAB ? kRayLdsMax : 1];
  const RayLds rt = stage_ray_table<TAB>(p.ray_tab, ray_lds, W, H);
  float acc[29];
  se3_accumulate<TAB>(g, p, rt, W, H, huber_delta, acc);
  block_reduce_store<29>(acc, partials + (size_t)blockIdx.x * kSimpleRow);
}

__global__ __launch_bounds__(1024) void k_track_update(const float* __restrict__ partials_all, const int nblocks, TrackState* __restrict__ states) {
  const float* partials = partials_all + (size_t)blockIdx.x * nblocks * kSimpleRow;
  TrackState* st = states + blockIdx.x;
  __shared__ double red[32][kSimpleRow];
  __shared__ double sum[kSimpleRow];
  const int e = threadIdx.x & 31, rg = threadIdx.x >> 5;
  double s = strided_sum_f64_wide<32, 32>(partials + e, rg, nblocks, kSimpleRow);
  red[rg][e] = s;
  __syncthreads();
  if (rg == 0) {
    s = 0.0;
#pragma unroll
    for (int q = 0; q < 32; ++q) s += red[q][e];
    sum[e] = s;
  }
  __syncthreads();
  if (threadIdx.x != 0) return;
  // Thread 0 solves.  Every loop below is fully unrolled so that the small matrices are REGISTERS (as dynamically indexed private
  // arrays they landed in scratch memory and made this kernel as slow as the step kernel itself; as LDS arrays every dependent
  // access paid an LDS round trip: 8.3 us for the kernel, most of a tracker iteration).
  // reference precision: the item is fp32 (JTJJrReductionItem<float,6>) before the solve (camera_tracker.cpp:59)
  double A[6][6], bvec[6];
  {
    int k = 0;
#pragma unroll
    for (int a = 0; a < 6; ++a)
#pragma unroll
      for (int b = a; b < 6; ++b) { const double v = (double)(float)sum[k++]; A[a][b] = v; A[b][a] = v; }
#pragma unroll
    for (int a = 0; a < 6; ++a) bvec[a] = (double)(float)sum[21 + a];
  }
  st->last_residual = (float)sum[27];
  st->last_inliers = (float)sum[28];
  st->iterations_done += 1;
  // LDL^T (no pivoting), forward / diagonal / backward substitution
  double L[6][6], D[6], yv[6], x[6];
  bool ok = sum[28] > 0.0;
#pragma unroll
  for (int j = 0; j < 6; ++j) {
    double d = A[j][j];
#pragma unroll
    for (int q = 0; q < j; ++q) d -= L[j][q] * L[j][q] * D[q];
    ok = ok && (fabs(d) > 0.0);
    D[j] = d;
#pragma unroll
    for (int i = j + 1; i < 6; ++i) {
      double v = A[i][j];
#pragma unroll
      for (int q = 0; q < j; ++q) v -= L[i][q] * L[j][q] * D[q];
      L[i][j] = v / d;
    }
  }
  if (!ok) { st->solver_failures += 1; return; }
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    double v = bvec[i];
#pragma unroll
    for (int q = 0; q < i; ++q) v -= L[i][q] * yv[q];
    yv[i] = v;
  }
#pragma unroll
  for (int i = 5; i >= 0; --i) {
    double v = yv[i] / D[i];
#pragma unroll
    for (int q = i + 1; q < 6; ++q) v -= L[q][i] * x[q];
    x[i] = v;
  }
  // update = -x ; t += update[0:3] ; R = exp(update[3:6]) * R   (lucas_kanade_se3.h:85-95)
  const double w0 = -x[3], w1 = -x[4], w2 = -x[5];
  const double th2 = w0 * w0 + w1 * w1 + w2 * w2, th = sqrt(th2);
  const double Ac = th < 1e-9 ? 1.0 - th2 / 6.0 : sin(th) / th, Bc = th < 1e-9 ? 0.5 - th2 / 24.0 : (1.0 - cos(th)) / th2;
  const double K[3][3] = { { 0, -w2, w1 }, { w2, 0, -w0 }, { -w1, w0, 0 } };
  double E[3][3], Rn[3][3], Ro[3][3];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      Ro[i][j] = st->R[i * 3 + j];
      double k2 = 0;
#pragma unroll
      for (int q = 0; q < 3; ++q) k2 += K[i][q] * K[q][j];
      E[i][j] = (i == j ? 1.0 : 0.0) + Ac * K[i][j] + Bc * k2;
    }
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      double v = 0;
#pragma unroll
      for (int q = 0; q < 3; ++q) v += E[i][q] * Ro[q][j];
      Rn[i][j] = v;
    }
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) { st->R[i * 3 + j] = Rn[i][j]; st->Rf[i * 3 + j] = (float)Rn[i][j]; }
#pragma unroll
  for (int i = 0; i < 3; ++i) { st->t[i] -= x[i]; st->tf[i] = (float)st->t[i]; }
}

size_t track_state_bytes() { return sizeof(TrackState); }

hipError_t launch_track_iteration(const SimplePairDev* descs_dev, int n, void* states_dev, int W, int H, float huber_delta, int blocks,
                                  float* partials_dev, hipStream_t stream) {
  const bool tab_ok = W + H <= kRayLdsMax;   // the per-camera ray table fits the kernels' LDS budget (every descriptor carries one)
  if (tab_ok) hipLaunchKernelGGL((k_se3_step_dev<true>), dim3(blocks, n), dim3(kT), 0, stream, descs_dev, (const TrackState*)states_dev, W, H, huber_delta, partials_dev); else hipLaunchKernelGGL((k_se3_step_dev<false>), dim3(blocks, n), dim3(kT), 0, stream, descs_dev, (const TrackState*)states_dev, W, H, huber_delta, partials_dev);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(k_track_update, dim3(n), dim3(1024), 0, stream, (const float*)partials_dev, blocks, (TrackState*)states_dev);
  return hipGetLastError();
}

void track_state_init(void* host_state, const double* R, const double* t) {
  TrackState* s = (TrackState*)host_state;
  for (int i = 0; i < 9; ++i) { s->R[i] = R[i]; s->Rf[i] = (float)R[i]; }
  for (int i = 0; i < 3; ++i) { s->t[i] = t[i]; s->tf[i] = (float)t[i]; }
  s->last_residual = 0.f; s->last_inliers = 0.f; s->solver_failures = 0; s->iterations_done = 0;
}
void track_state_read(const void* host_state, double* R, double* t, float* residual, float* inliers, int* failures, int* iters) {
  const TrackState* s = (const TrackState*)host_state;
  for (int i = 0; i < 9; ++i) R[i] = s->R[i];
  for (int i = 0; i < 3; ++i) t[i] = s->t[i];
  *residual = s->last_residual; *inliers = s->last_inliers; *failures = s->solver_failures; *iters = s->iterations_done;
}

// ---- SparseGeometricFactor::linearize (reference core/gtsam/sparse_geometric_factor.cpp:147-275) ---------------------------
// One lane per sampled point (N <= a few thousand): decode depth at the point in kf0, warp, decode kf1's depth at the
// nearest-neighbour pixel of the projection, residual err = dpt1 - (R p + t).z, Huber weight, one row
// [err_J_pose0 (6) | err_J_pose1 (6) | err_J_cde0 (CS) | err_J_cde1 (CS) | err] per point (zero row if no correspondence).
struct SparseGeoDev {
  float R[9], t[3], M[9], HM[9];
  float fx, fy, u0, v0, w, h;
  const float* prx0; const float* jac0; const float* prx1; const float* jac1; const float* dgrad1;
  uint32_t pitch_prx0, pitch_jac0, pitch_prx1, pitch_jac1, pitch_dgrad1;
  float huber_delta, avg_dpt;
};

template <int CS>
__global__ __launch_bounds__(64) void k_sparse_geometric(const SparseGeoDev P, const float* __restrict__ code0, const float* __restrict__ code1,
                                                         const int* __restrict__ pts, const int npts, float* __restrict__ rows) {
  constexpr int NC = 12 + 2 * CS + 1;
  const int i = blockIdx.x * 64 + threadIdx.x;
  if (i >= npts) return;
  float* row = rows + (size_t)i * NC;
  const int x = pts[2 * i], y = pts[2 * i + 1];
  Geo g;
#pragma unroll
  for (int q = 0; q < 9; ++q) g.R[q] = P.R[q];
  g.t[0] = P.t[0]; g.t[1] = P.t[1]; g.t[2] = P.t[2];
  g.fx = P.fx; g.fy = P.fy; g.u0 = P.u0; g.v0 = P.v0; g.w = P.w; g.h = P.h;
  const char* j0p = (const char*)P.jac0 + (size_t)y * P.pitch_jac0 + (size_t)x * CS * 4;
  float dot0 = 0.f;
  {
#pragma clang fp contract(off)
    for (int k = 0; k < CS; ++k) dot0 += gload<float>(j0p + 4 * k) * code0[k];   // sequential, like DepthFromCode
  }
  const float a = P.avg_dpt;
  const float d0 = a / (gload<float>((const char*)P.prx0 + (size_t)y * P.pitch_prx0 + (size_t)x * 4) + dot0) - a;
  const Corr c = find_correspondence(g, x, y, d0, 1.0f, 0.0f);
  if (!c.valid) {
    for (int k = 0; k < NC; ++k) row[k] = 0.f;
    return;
  }
  const int nx = (int)c.u, ny = (int)c.v;   // cast<int>: truncation (sparse_geometric_factor.cpp:207)
  const char* j1p = (const char*)P.jac1 + (size_t)ny * P.pitch_jac1 + (size_t)nx * CS * 4;
  float dot1 = 0.f;
  {
#pragma clang fp contract(off)
    for (int k = 0; k < CS; ++k) dot1 += gload<float>(j1p + 4 * k) * code1[k];
  }
  const float d1 = a / (gload<float>((const char*)P.prx1 + (size_t)ny * P.pitch_prx1 + (size_t)nx * 4) + dot1) - a;
  const float qz = 1.0f / c.iz;
  const float err = d1 - (c.vz + g.t[2]);
  (void)qz;
  const f32x2 dg = gload<f32x2>((const char*)P.dgrad1 + (size_t)ny * P.pitch_dgrad1 + (size_t)nx * 8);
  // gC = -(dgrad . C) with C = D [I | -hat(R p)]; third row of [I | -hat(R p)] is (0, 0, 1, v.y, -v.x, 0)
  float gC[6], D00, D02, D11, D12;
  pose_row(g, c, d0, dg.x, dg.y, gC, D00, D02, D11, D12);
  const float a10[6] = { gC[0], gC[1], 1.0f + gC[2], c.vy + gC[3], -c.vx + gC[4], gC[5] };
  float e0[6], e1[6];
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    e0[j] = a10[0] * P.M[j] + a10[1] * P.M[3 + j] + a10[2] * P.M[6 + j];
    e0[3 + j] = a10[3] * P.M[j] + a10[4] * P.M[3 + j] + a10[5] * P.M[6 + j];
    e1[j] = -e0[j];
    e1[3 + j] = -(a10[0] * P.HM[j] + a10[1] * P.HM[3 + j] + a10[2] * P.HM[6 + j]) - e0[3 + j];
  }
  const float apd0 = a + d0, apd1 = a + d1;
  const float dprx0 = -(apd0 * apd0) / a;
  const float pj0 = D00 * c.rrx + D02 * c.rrz, pj1 = D11 * c.rry + D12 * c.rrz;
  const float sc0 = (c.rrz - (dg.x * pj0 + dg.y * pj1)) * dprx0;
  const float sc1 = (apd1 * apd1) / a;   // -DepthJacobianPrx(dpt1)
  const float wgt = huber_weight(err, P.huber_delta);
#pragma unroll
  for (int j = 0; j < 6; ++j) { row[j] = e0[j] * wgt; row[6 + j] = e1[j] * wgt; }
  for (int k = 0; k < CS; ++k) {
    row[12 + k] = sc0 * gload<float>(j0p + 4 * k) * wgt;
    row[12 + CS + k] = sc1 * gload<float>(j1p + 4 * k) * wgt;
  }
  row[12 + 2 * CS] = err * wgt;
}

size_t sparse_geo_desc_bytes() { return sizeof(SparseGeoDev); }
void sparse_geo_fill(void* desc, const float* R, const float* t, const float* M, const float* HM, const float* cam6, const float* prx0,
                     uint32_t pp0, const float* jac0, uint32_t pj0, const float* prx1, uint32_t pp1, const float* jac1, uint32_t pj1,
                     const float* dgrad1, uint32_t pg1, float huber_delta, float avg_dpt) {
  SparseGeoDev* d = (SparseGeoDev*)desc;
  for (int i = 0; i < 9; ++i) { d->R[i] = R[i]; d->M[i] = M[i]; d->HM[i] = HM[i]; }
  for (int i = 0; i < 3; ++i) d->t[i] = t[i];
  d->fx = cam6[0]; d->fy = cam6[1]; d->u0 = cam6[2]; d->v0 = cam6[3]; d->w = cam6[4]; d->h = cam6[5];
  d->prx0 = prx0; d->jac0 = jac0; d->prx1 = prx1; d->jac1 = jac1; d->dgrad1 = dgrad1;
  d->pitch_prx0 = pp0; d->pitch_jac0 = pj0; d->pitch_prx1 = pp1; d->pitch_jac1 = pj1; d->pitch_dgrad1 = pg1;
  d->huber_delta = huber_delta; d->avg_dpt = avg_dpt;
}
hipError_t launch_sparse_geometric(int cs, const void* desc_host, const float* code0_dev, const float* code1_dev, const int* pts_dev, int npts,
                                   float* rows_dev, hipStream_t stream) {
  const SparseGeoDev& P = *(const SparseGeoDev*)desc_host;
  const int blocks = (npts + 63) / 64;
  switch (cs) {
    case 16: hipLaunchKernelGGL(k_sparse_geometric<16>, dim3(blocks), dim3(64), 0, stream, P, code0_dev, code1_dev, pts_dev, npts, rows_dev); break;
    case 32: hipLaunchKernelGGL(k_sparse_geometric<32>, dim3(blocks), dim3(64), 0, stream, P, code0_dev, code1_dev, pts_dev, npts, rows_dev); break;
    case 64: hipLaunchKernelGGL(k_sparse_geometric<64>, dim3(blocks), dim3(64), 0, stream, P, code0_dev, code1_dev, pts_dev, npts, rows_dev); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

// ---- SfM error: sum (w r)^2, inliers -----------------------------------------------------------------------
template <bool TAB>
__device__ __forceinline__ void sfm_error_accumulate(const Geo& g, const SimplePairDev& p, const RayLds& rt, const int W, const int H, const float huber_delta,
                                                     float (&acc)[2]) {

  acc[0] = acc[1] = 0.f;
  walk_pixels<TAB, false>(g, p, rt, W, H, 1.0f, 0.0f, [&](float, float i0, const Corr& c, const TapLoads& t) {   // dense_sfm.h:91: default border 1, min_dpt 0
    if (c.valid) {
      float r = i0 - taps_img(t);
      r *= huber_weight(r, huber_delta);
      acc[0] += r * r;
      acc[1] += 1.0f;
    }
  });
}

template <bool TAB>
__global__ __launch_bounds__(kT) void k_sfm_error(const SimplePairDev p, const int W, const int H, const float huber_delta,
                                                  float* __restrict__ partials) {
  __shared__ float ray_lds[TAB ? kRayLdsMax : 1];
  const Geo g = geo_from(p);
  const RayLds rt = stage_ray_table<TAB>(p.ray_tab, ray_lds, W, H);
  float acc[2];
  sfm_error_accumulate<TAB>(g, p, rt, W, H, huber_delta, acc);
  block_reduce_store<2>(acc, partials + (size_t)blockIdx.x * kSimpleRow);
}

// ---- batched forms (blockIdx.y = pair): PhotometricFactor::error over a factor set evaluates one pair per blocking call in the reference
// (core/gtsam/photometric_factor.cpp:61-81,197-216); a relocalisation / loop-closure check steps one live frame against many keyframes.
// Same per-pair arithmetic and reduction order as the single-pair kernels launched with the same number of workgroups.
template <bool TAB>
__global__ __launch_bounds__(kT) void k_sfm_error_batch(const SimplePairDev* __restrict__ descs, const int W, const int H, const float huber_delta,
                                                        float* __restrict__ partials_all) {
  __shared__ float ray_lds[TAB ? kRayLdsMax : 1];
  const SimplePairDev& p = descs[blockIdx.y];
  const Geo g = geo_from(p);
  const RayLds rt = stage_ray_table<TAB>(p.ray_tab, ray_lds, W, H);
  float acc[2];
  sfm_error_accumulate<TAB>(g, p, rt, W, H, huber_delta, acc);
  block_reduce_store<2>(acc, partials_all + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * kSimpleRow);
}

template <bool TAB>
__global__ __launch_bounds__(kT) void k_se3_step_batch(const SimplePairDev* __restrict__ descs, const int W, const int H, const float huber_delta,
                                                       float* __restrict__ partials_all) {
  __shared__ float ray_lds[TAB ? kRayLdsMax : 1];
  const SimplePairDev& p = descs[blockIdx.y];
  const Geo g = geo_from(p);
  const RayLds rt = stage_ray_table<TAB>(p.ray_tab, ray_lds, W, H);
  float acc[29];
  se3_accumulate<TAB>(g, p, rt, W, H, huber_delta, acc);
  block_reduce_store<29>(acc, partials_all + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * kSimpleRow);
}

// ---- Warp: render img1 into frame 0, SIGNED residual sum (cu_se3aligner.cpp:106) ----------------------------
__global__ __launch_bounds__(kT) void k_se3_warp(const SimplePairDev p, const int W, const int H, float* __restrict__ partials) {
  const Geo g = geo_from(p);
  const ImgRef I0{ (const char*)p.img0, p.pitch_img0 }, I1{ (const char*)p.img1, p.pitch_img1 };
  const ImgRef D0{ (const char*)p.dpt0, p.pitch_dpt0 };
  float acc[2] = { 0.f, 0.f };
  const int npx = W * H;
  for (int i = blockIdx.x * kT + threadIdx.x; i < npx; i += gridDim.x * kT) {
    const int y = i / W, x = i - y * W;
    const float d = D0.at(x, y);
    const float i0 = I0.at(x, y);
    const Corr c = find_correspondence(g, x, y, d, 1.0f, 0.0f);   // `depth <= 0 -> skip`, PixelValid(pix1, 1)
    float outv = 0.f;
    if (c.valid) {
      const Taps tp = make_taps(c.u, c.v);
      outv = sample_img(I1, tp);
      acc[0] += i0 - outv;
      acc[1] += 1.0f;
    }
    gstore<float>((char*)p.img2 + (size_t)y * p.pitch_img2 + (size_t)x * 4, outv);
  }
  block_reduce_store<2>(acc, partials + (size_t)blockIdx.x * kSimpleRow);
}

// ---- SquaredError ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kT) void k_squared_error(const float* __restrict__ a, const uint32_t pa, const float* __restrict__ b,
                                                      const uint32_t pb, const int W, const int H, float* __restrict__ partials) {
  float acc[1] = { 0.f };
  const int npx = W * H;
  for (int i = blockIdx.x * kT + threadIdx.x; i < npx; i += gridDim.x * kT) {
    const int y = i / W, x = i - y * W;
    const float d = reinterpret_cast<const float*>((const char*)a + (size_t)y * pa)[x] -
                    reinterpret_cast<const float*>((const char*)b + (size_t)y * pb)[x];
    acc[0] += d * d;
  }
  block_reduce_store<1>(acc, partials + (size_t)blockIdx.x * kSimpleRow);
}

// ---- finalize: out[e] = sum_b partials[b][e] in double, fixed order; layout-specific scatter -------------------
enum FinalKind { kFinalItem6 = 0, kFinalCorr = 1, kFinalScalar = 2 };

__global__ __launch_bounds__(1024) void k_finalize_rows(const float* __restrict__ partials_all, const int nblocks, const int kind,
                                                        char* __restrict__ out_all, const size_t out_stride) {
  // blockIdx.x = pair of a batched launch (0 for the single-pair operators)
  const float* partials = partials_all + (size_t)blockIdx.x * nblocks * kSimpleRow;
  char* out = out_all + (size_t)blockIdx.x * out_stride;
  __shared__ double red[32][kSimpleRow];
  const int e = threadIdx.x & 31, rg = threadIdx.x >> 5;   // 32 row groups
  static_assert(kMaxSimpleBlocks <= 32 * 32, "one load per row group and thread");
  double s = strided_sum_f64_wide<32, 32>(partials + e, rg, nblocks, kSimpleRow);
  red[rg][e] = s;
  __syncthreads();
  if (rg != 0) return;
  s = 0.0;
  for (int q = 0; q < 32; ++q) s += red[q][e];
  if (kind == kFinalItem6) {
    // JTJJrReductionItem<float,6>: 21 + 6 + 1 floats, then u64 inliers at byte 112
    if (e < 28) reinterpret_cast<float*>(out)[e] = (float)s;
    else if (e == 28) *reinterpret_cast<unsigned long long*>(out + 112) = (unsigned long long)(s + 0.5);
  } else if (kind == kFinalCorr) {
    if (e == 0) reinterpret_cast<float*>(out)[0] = (float)s;
    else if (e == 1) *reinterpret_cast<unsigned long long*>(out + 8) = (unsigned long long)(s + 0.5);
  } else {
    if (e == 0) reinterpret_cast<float*>(out)[0] = (float)s;
  }
}

// ---- UpdateDepth: dpt = a / (prx0 + jac . code) - a  (the code-Jacobian decoder, GEMV, 8 + 4 CS bytes / pixel) ---
// LPP = CS/4 lanes share one pixel (float4 each -> every wave-load is 1 KiB contiguous); 64 pixels per wave step so
// that the prx read and the depth write are single coalesced 256-byte accesses.
template <int CS>
__device__ __forceinline__ void update_depth_body(const float* __restrict__ code, const float* __restrict__ prx,
                                                  const uint32_t pitch_prx, const float* __restrict__ jac,
                                                  const uint32_t pitch_jac, const float avg_dpt, float* __restrict__ out,
                                                  const uint32_t pitch_out, const int W, const int H) {
  constexpr int LPP = CS / 4;        // lanes per pixel
  constexpr int PPL = 64 / LPP;      // pixels per wave-load
  constexpr int NSUB = 64 / PPL;     // wave-loads per 64-pixel chunk (== LPP)
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int q = lane % LPP, grp = lane / LPP;
  const f32x4 c4 = *reinterpret_cast<const f32x4*>(code + 4 * q);
  const int npx = W * H;
  const int nchunks = (npx + 63) >> 6;
  // Fast path (every image of the reference's pyramids): rows are whole chunks (W % 64 == 0), so a chunk lies in ONE image row and its
  // 64 * CS Jacobian floats are one contiguous run -- no per-load row arithmetic, no bounds branches, and the stream is read with the
  // non-temporal policy (read once; keeps the L2 for the 8 B/px of prx / depth traffic next to it).
  const bool rows_are_chunks = (W & 63) == 0;
  for (int chunk = blockIdx.x * (kT / 64) + wave; chunk < nchunks; chunk += gridDim.x * (kT / 64)) {
    const int base = chunk << 6;
    const int y0 = base / W, x0 = base - y0 * W;
    // sub-step s: lane group `grp` handles pixel base + grp*NSUB + s  -> after NSUB steps lane l owns pixel base + l
    f32x4 v[NSUB];
    if (rows_are_chunks) {
      const f32x4* row = reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>((const char*)jac + (size_t)y0 * pitch_jac) + (size_t)x0 * CS) + q;
#pragma unroll
      for (int s = 0; s < NSUB; ++s) v[s] = __builtin_nontemporal_load(row + (grp * NSUB + s) * LPP);
    } else {
#pragma unroll
      for (int s = 0; s < NSUB; ++s) {
        const int off = grp * NSUB + s;
        int x = x0 + off, y = y0;
        while (x >= W) { x -= W; ++y; }
        f32x4 t = f32x4{ 0.f, 0.f, 0.f, 0.f };
        if (base + off < npx) t = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>((const char*)jac + (size_t)y * pitch_jac) + (size_t)x * CS + 4 * q);
        v[s] = t;
      }
    }
    float mine = 0.f;
#pragma unroll
    for (int s = 0; s < NSUB; ++s) {
      float d = v[s].x * c4.x + v[s].y * c4.y + v[s].z * c4.z + v[s].w * c4.w;
#pragma unroll
      for (int m = 1; m < LPP; m <<= 1) d += __shfl_xor(d, m, 64);
      if (q == s) mine = d;   // lane l = grp*LPP + q keeps pixel grp*NSUB + q = l (NSUB == LPP)
    }
    int x = x0 + lane, y = y0;
    if (!rows_are_chunks) while (x >= W) { x -= W; ++y; }
    if (rows_are_chunks || base + lane < npx) {
      const float p0 = reinterpret_cast<const float*>((const char*)prx + (size_t)y * pitch_prx)[x];
      const float pr = p0 + mine;
      reinterpret_cast<float*>((char*)out + (size_t)y * pitch_out)[x] = avg_dpt / pr - avg_dpt;
    }
  }
}

template <int CS>
__global__ __launch_bounds__(kT) void k_update_depth(const float* __restrict__ code, const float* __restrict__ prx,
                                                     const uint32_t pitch_prx, const float* __restrict__ jac,
                                                     const uint32_t pitch_jac, const float avg_dpt, float* __restrict__ out,
                                                     const uint32_t pitch_out, const int W, const int H) {
  update_depth_body<CS>(code, prx, pitch_prx, jac, pitch_jac, avg_dpt, out, pitch_out, W, H);
}

// n decode jobs of one image size in ONE launch (grid.y = job): Mapper::UpdateMap re-decodes every changed keyframe per level
// (core/mapping/mapper.cpp:860-888), dfx_sfm_linearize_batch decodes every distinct keyframe of a batch of pairs.
template <int CS>
__global__ __launch_bounds__(kT) void k_update_depth_batch(const DepthJobDev* __restrict__ jobs, const float avg_dpt, const int W, const int H) {
  const DepthJobDev& j = jobs[blockIdx.y];
  update_depth_body<CS>(j.code, j.prx, j.pitch_prx, j.jac, j.pitch_jac, avg_dpt, j.out, j.pitch_out, W, H);
}

// ---- Sobel / 8 with clamped borders (cu_image_proc.cpp:57-92) ---------------------------------------------------
__global__ __launch_bounds__(kT) void k_sobel(const float* __restrict__ img, const uint32_t pitch, float* __restrict__ grad,
                                              const uint32_t gpitch, const int W, const int H) {
  const int x = blockIdx.x * 64 + (threadIdx.x & 63);
  const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (x >= W || y >= H) return;
  const int xm = max(x - 1, 0), xp = min(x + 1, W - 1), ym = max(y - 1, 0), yp = min(y + 1, H - 1);
  const float* r0 = reinterpret_cast<const float*>((const char*)img + (size_t)ym * pitch);
  const float* r1 = reinterpret_cast<const float*>((const char*)img + (size_t)y * pitch);
  const float* r2 = reinterpret_cast<const float*>((const char*)img + (size_t)yp * pitch);
  const float a = r0[xm], b = r0[x], c = r0[xp], d = r1[xm], f = r1[xp], g = r2[xm], h = r2[x], i = r2[xp];
  // same tap order as the reference loop (py outer, px inner), zero taps skipped
  float sx = 0.f, sy = 0.f;
  sx += a * -1.f; sy += a * -1.f;
  sy += b * -2.f;
  sx += c * 1.f;  sy += c * -1.f;
  sx += d * -2.f;
  sx += f * 2.f;
  sx += g * -1.f; sy += g * 1.f;
  sy += h * 2.f;
  sx += i * 1.f;  sy += i * 1.f;
  f32x2 o = { sx / 8.f, sy / 8.f };
  *reinterpret_cast<f32x2*>(reinterpret_cast<float*>((char*)grad + (size_t)y * gpitch) + 2 * x) = o;
}

// ---- 5x5 binomial blur + decimate (cu_image_proc.cpp:134-164) ----------------------------------------------------
__global__ __launch_bounds__(kT) void k_blur_down(const float* __restrict__ in, const uint32_t pitch, const int W, const int H,
                                                  float* __restrict__ out, const uint32_t opitch, const int OW, const int OH) {
  const int x = blockIdx.x * 64 + (threadIdx.x & 63);
  const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (x >= OW || y >= OH) return;
  const float B[5] = { 1.f, 4.f, 6.f, 4.f, 1.f };
  float sum = 0.f, wall = 0.f;
#pragma unroll
  for (int py = 0; py < 5; ++py) {
    const int ny = min(max(2 * y + py - 2, 0), H - 1);
    const float* r = reinterpret_cast<const float*>((const char*)in + (size_t)ny * pitch);
#pragma unroll
    for (int px = 0; px < 5; ++px) {
      const int nx = min(max(2 * x + px - 2, 0), W - 1);
      const float k = B[px] * B[py];
      sum += r[nx] * k;
      wall += k;
    }
  }
  reinterpret_cast<float*>((char*)out + (size_t)y * opitch)[x] = sum / wall;
}

// ---------------------------------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------------------------------
hipError_t launch_se3_step(const SimplePairDev& p, int W, int H, float huber_delta, int blocks, float* partials_dev,
                           void* item_dev, hipStream_t stream) {
  const bool tab_ok = W + H <= kRayLdsMax;   // the per-camera ray table fits the kernels' LDS budget (every descriptor carries one)
  if (tab_ok) hipLaunchKernelGGL((k_se3_step<true>), dim3(blocks), dim3(kT), 0, stream, p, W, H, huber_delta, partials_dev); else hipLaunchKernelGGL((k_se3_step<false>), dim3(blocks), dim3(kT), 0, stream, p, W, H, huber_delta, partials_dev);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(k_finalize_rows, dim3(1), dim3(1024), 0, stream, (const float*)partials_dev, blocks, (int)kFinalItem6, (char*)item_dev, (size_t)0);
  return hipGetLastError();
}

hipError_t launch_sfm_error(const SimplePairDev& p, int W, int H, float huber_delta, int blocks, float* partials_dev,
                            void* corr_item_dev, hipStream_t stream) {
  const bool tab_ok = W + H <= kRayLdsMax;   // the per-camera ray table fits the kernels' LDS budget (every descriptor carries one)
  if (tab_ok) hipLaunchKernelGGL((k_sfm_error<true>), dim3(blocks), dim3(kT), 0, stream, p, W, H, huber_delta, partials_dev); else hipLaunchKernelGGL((k_sfm_error<false>), dim3(blocks), dim3(kT), 0, stream, p, W, H, huber_delta, partials_dev);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(k_finalize_rows, dim3(1), dim3(1024), 0, stream, (const float*)partials_dev, blocks, (int)kFinalCorr, (char*)corr_item_dev, (size_t)0);
  return hipGetLastError();
}

hipError_t launch_sfm_error_batch(const SimplePairDev* descs_dev, int n, int W, int H, float huber_delta, int blocks, float* partials_dev,
                                  void* corr_items_dev, hipStream_t stream) {
  const bool tab_ok = W + H <= kRayLdsMax;   // the per-camera ray table fits the kernels' LDS budget (every descriptor carries one)
  if (tab_ok) hipLaunchKernelGGL((k_sfm_error_batch<true>), dim3(blocks, n), dim3(kT), 0, stream, descs_dev, W, H, huber_delta, partials_dev); else hipLaunchKernelGGL((k_sfm_error_batch<false>), dim3(blocks, n), dim3(kT), 0, stream, descs_dev, W, H, huber_delta, partials_dev);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(k_finalize_rows, dim3(n), dim3(1024), 0, stream, (const float*)partials_dev, blocks, (int)kFinalCorr, (char*)corr_items_dev, (size_t)16);
  return hipGetLastError();
}

hipError_t launch_se3_step_batch(const SimplePairDev* descs_dev, int n, int W, int H, float huber_delta, int blocks, float* partials_dev,
                                 void* items_dev, hipStream_t stream) {
  const bool tab_ok = W + H <= kRayLdsMax;   // the per-camera ray table fits the kernels' LDS budget (every descriptor carries one)
  if (tab_ok) hipLaunchKernelGGL((k_se3_step_batch<true>), dim3(blocks, n), dim3(kT), 0, stream, descs_dev, W, H, huber_delta, partials_dev); else hipLaunchKernelGGL((k_se3_step_batch<false>), dim3(blocks, n), dim3(kT), 0, stream, descs_dev, W, H, huber_delta, partials_dev);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(k_finalize_rows, dim3(n), dim3(1024), 0, stream, (const float*)partials_dev, blocks, (int)kFinalItem6, (char*)items_dev, (size_t)120);
  return hipGetLastError();
}

hipError_t launch_se3_warp(const SimplePairDev& p, int W, int H, int blocks, float* partials_dev, void* corr_item_dev,
                           hipStream_t stream) {
  hipLaunchKernelGGL(k_se3_warp, dim3(blocks), dim3(kT), 0, stream, p, W, H, partials_dev);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(k_finalize_rows, dim3(1), dim3(1024), 0, stream, (const float*)partials_dev, blocks, (int)kFinalCorr, (char*)corr_item_dev, (size_t)0);
  return hipGetLastError();
}

hipError_t launch_squared_error(const float* a, uint32_t pitch_a, const float* b, uint32_t pitch_b, int W, int H, int blocks,
                                float* partials_dev, float* out_dev, hipStream_t stream) {
  hipLaunchKernelGGL(k_squared_error, dim3(blocks), dim3(kT), 0, stream, a, pitch_a, b, pitch_b, W, H, partials_dev);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(k_finalize_rows, dim3(1), dim3(1024), 0, stream, (const float*)partials_dev, blocks, (int)kFinalScalar, (char*)out_dev, (size_t)0);
  return hipGetLastError();
}

hipError_t launch_update_depth(int cs, const float* code_dev, const float* prx_orig, uint32_t pitch_prx, const float* jac,
                               uint32_t pitch_jac, float avg_dpt, float* dpt_out, uint32_t pitch_out, int W, int H,
                               hipStream_t stream) {
  const int nchunks = (W * H + 63) / 64;
  int blocks = (nchunks + 3) / 4;
  if (blocks > 2048) blocks = 2048;
  if (blocks < 1) blocks = 1;
  switch (cs) {
    case 16: hipLaunchKernelGGL(k_update_depth<16>, dim3(blocks), dim3(kT), 0, stream, code_dev, prx_orig, pitch_prx, jac, pitch_jac, avg_dpt, dpt_out, pitch_out, W, H); break;
    case 32: hipLaunchKernelGGL(k_update_depth<32>, dim3(blocks), dim3(kT), 0, stream, code_dev, prx_orig, pitch_prx, jac, pitch_jac, avg_dpt, dpt_out, pitch_out, W, H); break;
    case 64: hipLaunchKernelGGL(k_update_depth<64>, dim3(blocks), dim3(kT), 0, stream, code_dev, prx_orig, pitch_prx, jac, pitch_jac, avg_dpt, dpt_out, pitch_out, W, H); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

hipError_t launch_update_depth_batch(int cs, const DepthJobDev* jobs_dev, int njobs, float avg_dpt, int W, int H, hipStream_t stream) {
  const int nchunks = (W * H + 63) / 64;
  int blocks = (nchunks + 3) / 4;
  const int cap = (8192 + njobs - 1) / njobs;   // ~8 workgroups per CU over the whole batch
  if (blocks > cap) blocks = cap;
  if (blocks < 1) blocks = 1;
  const dim3 grid(blocks, njobs);
  switch (cs) {
    case 16: hipLaunchKernelGGL(k_update_depth_batch<16>, grid, dim3(kT), 0, stream, jobs_dev, avg_dpt, W, H); break;
    case 32: hipLaunchKernelGGL(k_update_depth_batch<32>, grid, dim3(kT), 0, stream, jobs_dev, avg_dpt, W, H); break;
    case 64: hipLaunchKernelGGL(k_update_depth_batch<64>, grid, dim3(kT), 0, stream, jobs_dev, avg_dpt, W, H); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

hipError_t launch_sobel(const float* img, uint32_t pitch, float* grad, uint32_t gpitch, int W, int H, hipStream_t stream) {
  hipLaunchKernelGGL(k_sobel, dim3((W + 63) / 64, (H + 3) / 4), dim3(kT), 0, stream, img, pitch, grad, gpitch, W, H);
  return hipGetLastError();
}

hipError_t launch_blur_down(const float* in, uint32_t pitch, int W, int H, float* out, uint32_t opitch, int OW, int OH,
                            hipStream_t stream) {
  hipLaunchKernelGGL(k_blur_down, dim3((OW + 63) / 64, (OH + 3) / 4), dim3(kT), 0, stream, in, pitch, W, H, out, opitch, OW, OH);
  return hipGetLastError();
}

}  // namespace dfx
