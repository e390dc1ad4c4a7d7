// dfx_kernels.hpp -- launcher declarations shared between the kernel TUs and the C-ABI TU.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace dfx {

// Device-side descriptor of one keyframe->frame pair (argument list of SfmAligner::RunStep,
// cu_sfmaligner.h:76-86, after the host has folded RelativePose + its Jacobians, cu_sfmaligner.cpp:166).
struct SfmPairDev {
  float R[9], t[3];   // pose_10
  float M[9];         // Ra^T, Ra = R(pose1):  pose10_J_pose0 = blkdiag(M, M)            (warping.h:128-134)
  float HM[9];        // hat(Ra^T (ta - tb)) Ra^T: pose10_J_pose1 = [[-M, -HM], [0, -M]]  (warping.h:119-126)
  float fx, fy, u0, v0, w, h;
  const float* img0;
  const float* img1;
  const float* dpt0;
  float* valid0;      // may be null
  const float* jac;   // [H][W*CS]
  const float* grad1; // [H][W][2]
  const float* ray_tab;   // [W] (x - u0) / fx, then [H + kRayTabSlack] (y - v0) / fy   (SfmAligner::RunStep only)
  unsigned long long* valid0_shadow;   // library-owned valid0 images: one bit per pixel (linear index), set = "holds 1.0"; else null
  uint32_t pitch_img0, pitch_img1, pitch_dpt0, pitch_valid0, pitch_jac, pitch_grad1;   // bytes
  // batches whose pairs differ in image size (several pyramid levels in ONE launch): the pair's own size, its share of the 1-D grid and
  // the index of its first workgroup partial (`blkmap` of launch_sfm_step); unused (0) by launches of one image size
  uint32_t w_px, h_px, nblk, blk0;
};
constexpr int kRayTabSlack = 80;   // rows a lane past the last pixel may index (<= 64 / W + 1), zero-filled

struct SfmParamsDev {
  float huber_delta, avg_dpt, min_dpt, border;
  unsigned launch_id;   // non-zero, unique per step launch of the process: stamps valid0 shadows that need a rebuild (dfx_sfm_step.hip)
};

// Device-side view of a keyframe graph (dfx_graph): CSR lists of the pairs incident to each node, ascending pair index.
struct GraphDev {
  int n_nodes, n_pairs;
  const int* kf_begin;   // [n_nodes + 1] into kf_pairs: pairs whose KEYFRAME (pose0, code0) is the node
  const int* kf_pairs;
  const int* fr_begin;   // [n_nodes + 1] into fr_pairs: pairs whose FRAME (pose1) is the node
  const int* fr_pairs;
};

// Graph assembly inside the reduction tail of a batched step (k_sfm_tail_b3): the last pair to arrive at a node gathers the node
struct TailGraphDev {
  GraphDev G;
  const int* pair_nodes;   // [n_pairs][2] = (keyframe node, frame node)
  int first_pair, n_local;
  float* sys;              // null: no assembly
  unsigned* node_cnt;      // [n_nodes], zero between launches (the assembling workgroup rewinds its counter)
};

// Dynamic schedule of the batched SfM step (k_sfm_step<..., DYN>): per-pair item queues popped by wave-workers
struct DynDev {
  unsigned* qhead;      // [npairs] next item of each pair; rewound by k_sfm_finalize
  int items_per_pair;   // columns (W / 64) x row bands
  int rows_per_item;    // image rows per item (<= 32)
  int npairs;
  int team;             // waves per pair = partials per pair
  unsigned vs_magic;    // 2^32 / (W / 64) + 1: chunk id -> image row by one multiply-high
};

struct alignas(16) DepthJobDev {   // one UpdateDepth of a batch (k_update_depth_batch): dpt = a / (prx + jac . code) - a
  float code[64];        // read as float4 vectors: the struct (and every slot of a descriptor array) is 16-byte aligned
  const float* prx;
  const float* jac;
  float* out;
  uint32_t pitch_prx, pitch_jac, pitch_out, _pad;
};
static_assert(sizeof(DepthJobDev) % 16 == 0 && alignof(DepthJobDev) == 16, "float4 loads of DepthJobDev::code need 16-byte slots");

// Wave-uniform constants of the FAST geometry of the pixel reductions (SE3 step, EvaluateError; dfx_misc_kernels.hip `row_walk`), derived
// once per pair from (R, t, camera) on the host when a descriptor is filled; when the pose lives on the device (the tracker) every workgroup of
// k_se3_step_dev rebuilds the pose-dependent part from the state it reads (fast_band in dfx_misc_kernels.hip: a dozen scalar-unit instructions).
// Image coordinates are centred on (w/2, h/2) so that PixelValid(border = 1) is the symmetric test |u_c| < w/2 - 1, evaluated without a
// division as  m = |X| - hw * Z < 0  with  X = fx q.x + (u0 - w/2) q.z,  Z = q.z.  The inlier set stays EXACTLY the reference's
// (pinhole_camera_impl.h:105-108 on the values of warping.h:204-241): a pixel whose margin lies within E = e1 |d| + e2 of zero -- a
// rigorous bound on the difference between the fast and the reference-order arithmetic -- is re-evaluated in the reference's
// operation order (find_correspondence_ray), wave-uniformly and rarely (a band of ~1e-3 pixels along the view border).
struct FastGeo {
  float KR[9];      // rows: fx R_0 + cu R_2, fy R_1 + cv R_2, R_2        (cu = u0 - w/2, cv = v0 - h/2)
  float Kt[3];      // fx t_0 + cu t_2, fy t_1 + cv t_2, t_2
  float cu, cv;
  float hw, hh;     // w/2 - 1, h/2 - 1
  float fcx, fcy;   // w/2 - floor(w/2), h/2 - floor(h/2): tap coordinate = centred coordinate + fc, relative to pixel (icx, icy)
  float du, dv;     // w/2 - u0, h/2 - v0: u - u0 = u_c + du
  int icx, icy;     // floor(w/2), floor(h/2)
  float e1, e2;     // ambiguity band of the margin: E = e1 |d| + e2
};
// Camera-only part: what a kernel needs to rebuild (KR, Kt, e1, e2) when the pose changes on the device.
struct FastCam {
  float rxmax, rymax;   // max |K^-1 (x, y, 1)| over the image
  float gscale;         // 32 * 2^-24 * (fx + fy + 2 (w + h) + 2 (|u0| + |v0|)): see derive_fast_geo
};
#if defined(__HIPCC__) || defined(__CUDACC__)
#define DFX_HD __host__ __device__
#else
#define DFX_HD
#endif
DFX_HD inline double dfx_absd(double v) { return v < 0 ? -v : v; }
DFX_HD inline double dfx_maxd(double a, double b) { return a > b ? a : b; }
DFX_HD inline double dfx_floord(double v) { const long long i = (long long)v; return (double)(i > v ? i - 1 : i); }
DFX_HD inline void derive_fast_cam(float fx, float fy, float u0, float v0, float w, float h, int W, int H, FastCam* c) {
  const double rx = dfx_maxd(dfx_absd(0.0 - u0), dfx_absd((double)(W - 1) - u0)) / dfx_absd((double)fx);
  const double ry = dfx_maxd(dfx_absd(0.0 - v0), dfx_absd((double)(H - 1) - v0)) / dfx_absd((double)fy);
  c->rxmax = (float)(rx * 1.000001);
  c->rymax = (float)(ry * 1.000001);
  // Error budget of the margin (homogeneous units): both evaluations of fx q.x + (u0 - bound) q.z differ from the real value by at most
  // ~8 * 2^-24 * (fx + |u0| + w) * S, S = sum of the absolute terms of a component of q = R p + t (<= rho |d| + tau); the division and
  // the rounded "+ u0" of the reference order add 3 * 2^-24 * (2 w + |u0|) * S.  32 * 2^-24 * G * S with G below is > 2x that sum.
  const double G = dfx_absd((double)fx) + dfx_absd((double)fy) + 2.0 * ((double)w + (double)h) + 2.0 * (dfx_absd((double)u0) + dfx_absd((double)v0));
  c->gscale = (float)(32.0 * 5.9604644775390625e-08 * G * 1.000001);
}
DFX_HD inline void derive_fast_geo(const double* R, const double* t, float fx, float fy, float u0, float v0, float w, float h,
                                   const FastCam& c, FastGeo* g) {
  const double cu = (double)u0 - 0.5 * (double)w, cv = (double)v0 - 0.5 * (double)h;
  for (int j = 0; j < 3; ++j) {
    g->KR[j] = (float)((double)fx * R[j] + cu * R[6 + j]);
    g->KR[3 + j] = (float)((double)fy * R[3 + j] + cv * R[6 + j]);
    g->KR[6 + j] = (float)R[6 + j];
  }
  g->Kt[0] = (float)((double)fx * t[0] + cu * t[2]);
  g->Kt[1] = (float)((double)fy * t[1] + cv * t[2]);
  g->Kt[2] = (float)t[2];
  g->cu = (float)cu; g->cv = (float)cv;
  g->hw = (float)(0.5 * (double)w - 1.0); g->hh = (float)(0.5 * (double)h - 1.0);
  const double fw = dfx_floord(0.5 * (double)w), fh = dfx_floord(0.5 * (double)h);
  g->fcx = (float)(0.5 * (double)w - fw); g->fcy = (float)(0.5 * (double)h - fh);
  g->icx = (int)fw; g->icy = (int)fh;
  g->du = (float)(-cu); g->dv = (float)(-cv);
  double rho = 0.0, tau = 0.0;
  for (int i = 0; i < 3; ++i) {
    rho = dfx_maxd(rho, dfx_absd(R[3 * i]) * c.rxmax + dfx_absd(R[3 * i + 1]) * c.rymax + dfx_absd(R[3 * i + 2]));
    tau = dfx_maxd(tau, dfx_absd(t[i]));
  }
  g->e1 = (float)((double)c.gscale * rho * 1.000001);
  g->e2 = (float)((double)c.gscale * tau * 1.000001);
}

struct SimplePairDev {   // SE3Aligner / EvaluateError / Warp
  float R[9], t[3];
  float fx, fy, u0, v0, w, h;
  FastGeo fg;           // of (R, t): SE3 step / EvaluateError; the device-resident tracker keeps its own copy beside the pose
  FastCam fc;
  const float* img0;
  const float* img1;
  const float* dpt0;
  const float* grad1;   // null for error / warp
  float* img2;          // warp output or null
  const float* ray_tab; // per-camera ray table ([W] (x - u0) / fx, then [H ...] (y - v0) / fy) or null: the rays are then computed per pixel
  uint32_t pitch_img0, pitch_img1, pitch_dpt0, pitch_grad1, pitch_img2;
};

// z-space size of the SfM step partials: one 256-float block for the 29 (P,P) sums + the packed 16x16 MFMA blocks
// X(b,b'), Pm(b) + two 4x4x1 blocks per Dd(q) (see dfx_sfm_step.hip)
inline int sfm_nacc(int ncb) { return ncb * (ncb - 1) / 2 + ncb + 2 * ((ncb + 1) / 2); }
inline int sfm_zdim(int ncb) { return (1 + sfm_nacc(ncb)) * 256; }

// A blocking call's result lands in pinned, device-mapped host memory, written by the call's LAST kernel.  With a DoneFlag that kernel also stores the call's
// sequence number behind the result (system-scope release), and the host polls that word instead of waiting for the stream: hipStreamSynchronize learns of
// the end of a kernel 5.4 us after a polling host reads the flag (tools/ubench/sync_latency.cpp: 11.8 against 6.3 us for one kernel, 15.5 against 10.0 for two;
// a command-processor write behind the kernel, hipStreamWriteValue32, only gains 2.4) -- a quarter of an SE3Aligner::RunStep.  Kernels of several workgroups
// count arrivals in `counter` (device memory, zero between calls); flag == nullptr: no signal (batched and *_async entries).
struct DoneFlag { uint32_t* flag = nullptr; uint32_t seq = 0; unsigned* counter = nullptr; };
#if defined(__HIPCC__)
// the result was stored by lanes of the calling wave only
__device__ __forceinline__ void signal_done_wave(const DoneFlag& d) {
  if (!d.flag) return;
  __builtin_amdgcn_s_waitcnt(0);                    // this wave's stores have completed
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");     // system scope
  if ((threadIdx.x & 63) == 0) __hip_atomic_store(d.flag, d.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
// every thread of every workgroup of the grid calls this once its stores of the result are issued (the protocol of k_sfm_tail_b3's arrival counters)
__device__ __forceinline__ void signal_done_grid(const DoneFlag& d, const unsigned total_wgs) {
  if (!d.flag) return;
  __builtin_amdgcn_s_waitcnt(0);
  __syncthreads();
  if (threadIdx.x != 0) return;
  if (total_wgs > 1) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");   // into the count; the last arriver's system-scope release below carries every workgroup's stores to the host
    const unsigned got = __hip_atomic_fetch_add(d.counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u;
    if (got != total_wgs) return;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    __hip_atomic_store(d.counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // rewound for the next call
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
  __hip_atomic_store(d.flag, d.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
#endif

// All launchers enqueue on `stream` and return the HIP status of the launch.
// ev_begin/ev_end (optional) bracket the step kernel only.
hipError_t launch_sfm_step(int cs, const SfmPairDev* pairs_dev, int npairs, int W, int H, const SfmParamsDev& prm,
                           int blocks_per_pair, float* partials_dev, void* items_dev, size_t item_stride,
                           hipStream_t stream, bool jac_dense, int prec, hipEvent_t ev_begin = nullptr, hipEvent_t ev_end = nullptr,
                           const SfmPairDev* one_host = nullptr,   // one_host (npairs == 1): the descriptor travels in the kernel arguments
                           const DynDev* dyn = nullptr, int dyn_grid = 0,    // dyn: the dynamic schedule (workgroups = dyn_grid, partials = [pair][team])
                           bool valid0_shadows = false,                      // every non-null valid0 of the batch carries a shadow (SfmPairDev::valid0_shadow)
                           hipStream_t fin_stream = nullptr, hipEvent_t ev_mid = nullptr,   // deferred tail: the finalize kernel runs on fin_stream behind ev_mid
                           const unsigned* blkmap_dev = nullptr, int total_blocks = 0,    // pairs of several image sizes: 1-D grid, workgroup g serves pair
                                                                                           // blkmap[g] >> 16 as its block blkmap[g] & 0xffff (of SfmPairDev::nblk);
                                                                                           // W, H = the largest width / height (ray-table LDS); blocks_per_pair unused
                           const TailGraphDev* tail_graph = nullptr, int node_wgs = 0,    // graph assembly inside the reduction tail (k_sfm_tail_b3) where the
                           bool* assembled = nullptr,                                     // launch has one: *assembled tells; node_wgs = 0 or the graph's nodes
                           DoneFlag* done = nullptr,   // one_host launches: the finalize kernel signals *done; cleared (flag = nullptr) when the launch cannot
                           double* split_scratch = nullptr, unsigned* split_cnt = nullptr);   // one_host, bf16 split: kSplitScratchBytes of device scratch + 64 zeroed
                                                                                              // counters -> the four-workgroups-per-tile finalize kernel
size_t sfm_step_partials_bytes(int cs, int npairs, int blocks_per_pair);
// system layout (floats): Hd [n_nodes][D][D], Ho [n_pairs][D][6], g [n_nodes][D]; contributions of the pairs [first_pair, first_pair + n_local)
hipError_t launch_graph_assemble(int cs, const GraphDev& G, const void* items_dev, size_t item_stride, int first_pair, int n_local, float* sys_dev,
                                 hipStream_t stream);

hipError_t launch_se3_step(const SimplePairDev& p, int W, int H, float huber_delta, int blocks, float* partials_dev,
                           void* item_dev, hipStream_t stream, const DoneFlag& done = DoneFlag{});
hipError_t launch_sfm_error(const SimplePairDev& p, int W, int H, float huber_delta, int blocks, float* partials_dev,
                            void* corr_item_dev, hipStream_t stream, const DoneFlag& done = DoneFlag{});
// batched forms: descs_dev[n], partials [n][blocks][kSimpleRow], results packed (16 bytes per dfx_corr_item, 120 per JTJJrReductionItem<float,6>)
// ev_begin / ev_end (optional) bracket the reduction kernel only
hipError_t launch_sfm_error_batch(const SimplePairDev* descs_dev, int n, int W, int H, float huber_delta, int blocks, float* partials_dev,
                                  void* corr_items_dev, hipStream_t stream, hipEvent_t ev_begin = nullptr, hipEvent_t ev_end = nullptr);
hipError_t launch_se3_step_batch(const SimplePairDev* descs_dev, int n, int W, int H, float huber_delta, int blocks, float* partials_dev,
                                 void* items_dev, hipStream_t stream, hipEvent_t ev_begin = nullptr, hipEvent_t ev_end = nullptr);
hipError_t launch_se3_warp(const SimplePairDev& p, int W, int H, int blocks, float* partials_dev, void* corr_item_dev,
                           hipStream_t stream, const DoneFlag& done = DoneFlag{});
hipError_t launch_update_depth(int cs, const float* code_dev, const float* prx_orig, uint32_t pitch_prx, const float* jac,
                               uint32_t pitch_jac, float avg_dpt, float* dpt_out, uint32_t pitch_out, int W, int H,
                               hipStream_t stream);
hipError_t launch_update_depth_batch(int cs, const DepthJobDev* jobs_dev, int njobs, float avg_dpt, int W, int H, hipStream_t stream);
hipError_t launch_sobel(const float* img, uint32_t pitch, float* grad, uint32_t gpitch, int W, int H, hipStream_t stream);
hipError_t launch_blur_down(const float* in, uint32_t pitch, int W, int H, float* out, uint32_t opitch, int OW, int OH,
                            hipStream_t stream);
hipError_t launch_squared_error(const float* a, uint32_t pitch_a, const float* b, uint32_t pitch_b, int W, int H, int blocks,
                                float* partials_dev, float* out_dev, hipStream_t stream, const DoneFlag& done = DoneFlag{});
hipError_t launch_depth_aligner_step(int cs, const SfmPairDev* pair_host, int W, int H, float avg_dpt, int blocks,
                                     float* partials_dev, void* item_dev, hipStream_t stream, bool jac_dense, int prec, DoneFlag* done = nullptr,
                                     double* split_scratch = nullptr, unsigned* split_cnt = nullptr);
constexpr size_t kSplitScratchBytes = 64 * 4 * 512 * sizeof(double);   // b3_tiles(4) = 15 tiles <= 64, four groups, 256 + 256 doubles each

// device-resident tracker
size_t track_state_bytes();
void track_state_init(void* host_state, const double* R, const double* t);
void track_state_read(const void* host_state, double* R, double* t, float* residual, float* inliers, int* failures, int* iters);
// one Gauss-Newton iteration of `n` independent trackers at one pyramid level = one launch: the update from the previous evaluation (partials_prev
// [n][blocks_prev][kSimpleRow] at states_in[n]; blocks_prev 0 = none) -> states_out[n], then the evaluation at it -> partials_dev [n][blocks][kSimpleRow];
// launch_track_final applies the last evaluation
hipError_t launch_track_iteration(const SimplePairDev* descs_dev, int n, const void* states_in, void* states_out, const float* partials_prev, int blocks_prev,
                                  int W, int H, float huber_delta, int blocks, float* partials_dev, hipStream_t stream,
                                  const SimplePairDev* one_host = nullptr,    // n == 1: the level's descriptor by value (descs_dev unused) and, for the first
                                  const void* state0_host = nullptr);         // evaluation of a frame (blocks_prev == 0), the initial state by value
hipError_t launch_track_final(int n, const void* states_in, void* states_out, const float* partials_prev, int blocks_prev, hipStream_t stream,
                              const DoneFlag& done = DoneFlag{});   // (n == 1)

// SparseGeometricFactor::linearize, n factors per launch (descriptors in device-visible memory; codes inside the descriptor, points and rows device pointers)
size_t sparse_geo_desc_bytes();
// [A | b]^T [A | b] of every factor's rows (upper triangle, row-major, NC (NC + 1) / 2 floats per factor, NC = 12 + 2 CS + 1); descs_dev as launched above
hipError_t launch_rows_gram(int cs, const void* descs_dev, int n_factors, float* gram_dev, hipStream_t stream);
void sparse_geo_fill(void* desc, const float* R, const float* t, const float* M, const float* HM, const float* cam6, const float* code0, const float* code1, int cs,
                     const float* prx0, uint32_t pp0, const float* jac0, uint32_t pj0, const float* prx1, uint32_t pp1, const float* jac1, uint32_t pj1,
                     const float* dgrad1, uint32_t pg1, const int* pts_dev, int npts, int W, int H, float* rows_dev, float huber_delta, float avg_dpt);
hipError_t launch_sparse_geometric_batch(int cs, const void* descs_dev, int n_factors, int max_points, hipStream_t stream);

// one pyramid level of one frame (k_pyr_level): Sobel gradient of `in` (grad may be null) + blur-down into `out` (may be null: last level)
struct PyrLevelDev {
  const float* in; float* grad; float* out;
  uint32_t pitch_in, pitch_grad, pitch_out;
  int W, H, OW, OH;
};
// word / seq: the first workgroup of a build's first launch stores `seq` into the host-visible `word` ("build seq is running": see pyr_mirror_descs); null = no signal
struct PyrStart { uint32_t* word = nullptr; uint32_t seq = 0; };
// rows_ok: every frame's rows are aligned for the row-streaming kernel (image 8-byte, gradient 16-byte pointer and pitch); else the LDS-tile kernel
// mirror / mirror_count: the launch's first workgroup copies `mirror_count` descriptors from descs_dev (the pinned staging slot of a build's first launch) to
// `mirror` (device memory) for the launches behind it; null = no copy
hipError_t launch_pyr_level(const PyrLevelDev* descs_dev, int n, int W, int H, hipStream_t stream, bool rows_ok, PyrLevelDev* mirror = nullptr, int mirror_count = 0,
                            PyrStart st = PyrStart{}, int cus = 0);   // cus: compute units of the device (0: unknown), for the launch shape
int pyr_rows_per_segment(int H, int nstrips, int wpg, int gps, int n, int cus);
// the row-streaming launch of a level: waves per workgroup (the strips side by side, at most 8), workgroups per row segment, rows per segment
void pyr_rows_shape(int W, int H, int n, int cus, int* wpg, int* gps, int* R);
// levels k0 .. L-1 of every frame in one launch: `nb` bands (workgroups) per frame, each with its rows of those levels in LDS (descs: level-major [L][n]).
// pyr_tail_plan: Hs / Ws = the levels' sizes, nl = L - k0; returns the bands per frame (0: does not qualify), the band height at the last level and the LDS bytes.
constexpr size_t kPyrTailMaxLds = 144 * 1024;
int pyr_tail_plan(const int* Hs, const int* Ws, int nl, int n, int* rows_per, size_t* lds_bytes);
hipError_t launch_pyr_tail(const PyrLevelDev* descs_dev, int n, int k0, int L, int nb, int rows_per, size_t lds_bytes, hipStream_t stream);

constexpr int kSimpleRow = 32;       // floats per block partial of the VALU reduction kernels
constexpr int kMaxSimpleBlocks = 1024;

}  // namespace dfx
