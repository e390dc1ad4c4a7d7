/* dfx.h -- C ABI of libdfx.so: MI355X-native (gfx950, HIP) dense photometric alignment kernels that
 * replace the CUDA hot path of jczarnowski/DeepFactors behind its own operator interface.
 *
 * Every entry point names the reference interface it replaces (paths relative to the reference
 * repository root).  The reference has no C ABI; its path sits behind C++ templates explicitly
 * instantiated in libdf_cuda.so (sources/cuda/CMakeLists.txt:51-63).  include/dfx_shim.hpp holds
 * the header-compatible C++17 classes (df::SfmAligner<float,CS>, df::SE3Aligner<float>,
 * df::UpdateDepth) a maintainer drops in; they forward here.  See INTEGRATION.md.
 *
 * Conventions
 *  - All image pointers are DEVICE pointers (HIP) to fp32 data, row-pitched like VisionCore's
 *    Buffer2DView: element (x, y) lives at (char*)ptr + y*pitch_bytes + x*sizeof(elem).
 *    `grad` images hold interleaved (gx, gy) float pairs (Eigen::Matrix<float,1,2>), `prx_jac`
 *    images have w = W*CS floats per row (mapping/keyframe.h:52).
 *  - Poses are Sophus::SE3f by value: unit quaternion (x, y, z, w) + translation.
 *  - Every function returns 0 on success, a negative DFX_E_* code otherwise, and never throws;
 *    dfx_last_error() returns the thread-local message (the shim rethrows it, mirroring
 *    CudaCheckLastError, sources/cuda/launch_utils.h:26-32).
 *  - "No overlap" is signalled in-band like the reference: inliers == 0 (photometric_factor.cpp:213-216).
 *  - Synchronous calls block until the result is on the host, like the reference (cudaDeviceSynchronize
 *    + blocking copy, cu_sfmaligner.cpp:175-183).  *_async / *_batch variants only enqueue on the
 *    context's stream and leave results in device memory.
 */
#ifndef DFX_H_
#define DFX_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#if defined(_WIN32)
#define DFX_API
#else
#define DFX_API __attribute__((visibility("default")))
#endif

#define DFX_OK 0
#define DFX_E_INVALID (-1)   /* bad argument (null pointer, unsupported code size, misaligned jacobian ...) */
#define DFX_E_HIP (-2)       /* HIP runtime error; message in dfx_last_error() */
#define DFX_E_NOGPU (-3)     /* no usable HIP device / built without device code for it */

typedef struct dfx_ctx dfx_ctx; /* device id + stream + scratch; one per host thread */

/* vc::Image2DView<float,TargetDeviceCUDA> (VisionCore; used at cu_sfmaligner.h:74-75) */
typedef struct dfx_img {
  void* ptr;          /* device pointer */
  size_t pitch_bytes; /* row pitch */
  uint32_t w, h;      /* in ELEMENTS of the image's own type (floats; float pairs for grad; W*CS for prx_jac) */
} dfx_img;

/* Sophus::SE3f */
typedef struct dfx_se3 {
  float q[4]; /* x y z w */
  float t[3];
} dfx_se3;

/* df::PinholeCamera<float> (common/algorithm/pinhole_camera.h): fx fy u0 v0 width height */
typedef struct dfx_cam {
  float fx, fy, u0, v0, w, h;
} dfx_cam;

/* df::DenseSfmParams (common/algorithm/dense_sfm.h:36-43).  ocl_th is unused by the reference and dropped. */
typedef struct dfx_sfm_params {
  float huber_delta; /* 0.1 */
  float avg_dpt;     /* 2.0 */
  float min_dpt;     /* 0.0 */
  int32_t valid_border; /* 2 */
  /* SfmAlignerParams::step_blocks (cu_sfmaligner.h:44, SetStepThreadsBlocks cu_sfmaligner.cpp:196-203): workgroups per pair of
   * the step kernel for THIS call; 0 = the context's value (dfx_sfm_set_step_blocks), which defaults to automatic.  Passed per
   * call so that two aligners sharing a context cannot change each other's launch shape (the reference's __constant__
   * sfm_params has exactly that aliasing, cu_sfmaligner.cpp:34,111).  Threads per workgroup are fixed at 256 (4 waves). */
  int32_t step_blocks;
} dfx_sfm_params;

/* df::CorrespondenceReductionItem<float> (cuda/reduction_items.h:35-71): 16 bytes */
typedef struct dfx_corr_item {
  float residual;
  uint32_t _pad;
  uint64_t inliers;
} dfx_corr_item;

/* Result item: the MEMBERS of df::JTJJrReductionItem<float,NP> (cuda/reduction_items.h:77-143) in dfx's own packed layout --
 *   float JtJ[NP*(NP+1)/2]   packed upper triangle, row-major ((0,0),(0,1)..(0,NP-1),(1,1)..)
 *   float Jtr[NP]
 *   float residual
 *   (pad to 8)  uint64_t inliers
 * NP = 6 (SE3Aligner, 120 bytes), 12+CS (SfmAligner, 4152 bytes for CS=32), CS (DepthAligner).  This is NOT the reference struct's
 * in-memory layout (Eigen aligns Jtr to 16 bytes there: sizeof is 4160 for NP = 44), so never memcpy an item into the reference
 * struct: read it through the dfx_item_* accessors below (the C++ shim's FromRaw does).
 * Parameter order for SfmAligner: [pose0 (tx,ty,tz,wx,wy,wz), pose1 (6), code0 (CS)]. */
static inline size_t dfx_item_jtj_len(int np) { return (size_t)np * (size_t)(np + 1) / 2; }
static inline size_t dfx_item_inliers_offset(int np) {
  return ((dfx_item_jtj_len(np) + (size_t)np + 1) * sizeof(float) + 7) & ~(size_t)7;
}
static inline size_t dfx_item_size(int np) { return dfx_item_inliers_offset(np) + sizeof(uint64_t); }
static inline const float* dfx_item_jtj(const void* item) { return (const float*)item; }
static inline const float* dfx_item_jtr(const void* item, int np) { return (const float*)item + dfx_item_jtj_len(np); }
static inline float dfx_item_residual(const void* item, int np) { return ((const float*)item)[dfx_item_jtj_len(np) + np]; }
static inline uint64_t dfx_item_inliers(const void* item, int np) {
  return *(const uint64_t*)((const char*)item + dfx_item_inliers_offset(np));
}

/* ---- context ------------------------------------------------------------------------------- */
/* Replaces cuda::Init / per-aligner scratch buffers (cu_sfmaligner.cpp:101-114, cu_se3aligner.cpp:120).
 * `device` < 0 = the calling thread's current HIP device.  `stream` is the hipStream_t every call of this context enqueues on;
 * NULL = the device's default stream (what the reference uses).  The caller's producers of the input images must be ordered
 * with that stream.  Every image handed to a call must live on the context's device. */
DFX_API int dfx_ctx_create(int device, void* stream, dfx_ctx** out);
/* Re-binds the context to another stream of its device (waits for the work already enqueued on the old one). */
DFX_API int dfx_ctx_set_stream(dfx_ctx* ctx, void* stream);
/* Deferred tail (no reference counterpart; the reference finishes every RunStep with a second kernel + sync on one stream,
 * cu_sfmaligner.cpp:60-69,175-183).  A batched SfM step is a ~1 ms streaming kernel followed by a reduction tail of two short dependent
 * kernels (finalize, graph assembly: ~30 us of launch boundaries and latency-bound work at 128 pairs).  With a tail stream set, the
 * *_async entries enqueue the step kernel on the context's stream and the tail on `tail_stream` (another stream of the same device,
 * owned by the caller), so the tail of launch k runs beside the step kernel of launch k + 1; the library alternates two halves of its
 * scratch and orders them with events.  Consequences for the caller:
 *   - items / assembled systems of *_async calls (dfx_sfm_step_batch_async, dfx_sfm_linearize_batch_async, dfx_graph_assemble_async) are
 *     complete on the TAIL stream: consume them there, or call dfx_tail_join first (the context's stream then waits for every tail
 *     enqueued so far -- a stream-side wait, the host does not block);
 *   - output buffers handed to consecutive launches are written from the tail stream in launch order, so ONE items buffer / system can be
 *     reused by consecutive launches, but a consumer on another stream must have been ordered before the next launch overwrites them;
 *   - blocking entry points are unaffected (their tail runs on the context's stream); dfx_sync waits for both streams.
 * NULL switches the mode off.  Useful for a stream of independent batches (several windows in flight, throughput runs); a single
 * Gauss-Newton loop needs each result before the next launch and gains nothing. */
DFX_API int dfx_set_tail_stream(dfx_ctx* ctx, void* tail_stream);
DFX_API int dfx_tail_join(dfx_ctx* ctx);
DFX_API int dfx_ctx_device(dfx_ctx* ctx);
DFX_API void dfx_ctx_destroy(dfx_ctx* ctx);
DFX_API const char* dfx_last_error(void);
DFX_API const char* dfx_version(void);
/* Waits for everything enqueued on the context's stream (and on its tail stream, if one is set). */
DFX_API int dfx_sync(dfx_ctx* ctx);
/* Context-wide default of dfx_sfm_params.step_blocks (workgroups per pair of the step kernel); 0 = automatic (sized from the
 * CU count and the batch).  A non-zero dfx_sfm_params.step_blocks overrides it per call. */
DFX_API int dfx_sfm_set_step_blocks(dfx_ctx* ctx, int blocks_per_pair);
/* The automatic choice for a batch of `npairs` pairs of w x h (distinct_jacobians: every pair streams its own prx_jac), without launching
 * anything.  A pair's sums are bit-reproducible for a launch SHAPE (workgroups per pair); the automatic shape depends on the batch size, so the
 * same pair inside a 1024-pair launch and inside a 128-pair shard differs in the last bits.  A sharded job ("1/2/4/8-GPU results
 * bit-identical per pair") asks for the shape of the WHOLE pair list here and pins it on every rank through dfx_sfm_params.step_blocks:
 * tests/test_gpu_shard_invariance.py. */
/* Preconditions of that claim: the static schedule (a run under DFX_SCHEDULE_DYNAMIC sums whatever its queues hand a wave), ONE image size (a batch of several
 * sizes takes its shape from the whole batch's pixel count), and ONE value for all ranks -- the figure depends on the local CU count, so ranks on unlike GPUs
 * must take it from one of them (rank 0 computes, the others receive it with the pair list).  A pinned step_blocks forces the static schedule on that call. */
DFX_API int dfx_sfm_auto_step_blocks(dfx_ctx* ctx, int cs, uint32_t w, uint32_t h, int npairs, int distinct_jacobians, int* blocks_out);
DFX_API int dfx_device_cu_count(dfx_ctx* ctx);
/* How the JtJ/Jtr outer products of the SfM / DepthAligner step are evaluated on the matrix cores (fp32 in, fp32 out).  A context
 * starts in DFX_MFMA_AUTO.
 *  DFX_MFMA_AUTO       (default) the library's choice per code size, by measurement on MI355X: the exact bf16 split
 *                      from DFX_AUTO_BF16X3_MIN_CS on, i.e. at every supported code size.  Headline sweep (640x480, CS = 32, 128 pairs):
 *                      0.75 of the 8 TB/s roofline; the same sweep pinned to DFX_MFMA_F32_CHAIN: 0.69.
 *  DFX_MFMA_BF16X3     every fp32 entry is split EXACTLY into three bf16 pieces (x = h + m + l, round-to-nearest-even through
 *                      v_cvt_pk_bf16_f32) and the products are summed as hh + hm + mh + hl + lh + mm on v_mfma_f32_16x16x32_bf16
 *                      with fp32 accumulation; of the dropped terms ml and lm are each at most 2^-24 of a product (|m| <= 2^-8 |x|,
 *                      |l| <= 2^-16 |x|) and ll 2^-32: together at most 2^-23 in the worst case, ~2^-26 on average, i.e. the order of the
 *                      rounding of an fp32 multiply (2^-24) -- measured error against the fp64 oracle equals the chain's (tests/test_gpu_bf16x3.py).  Same inlier
 *                      sets, same valid0 writes, bit-reproducible for a launch shape like the chain; its bits differ from the chain's.
 *  DFX_MFMA_F32_CHAIN  pinned by a caller that wants the fmaf-chain bits: v_mfma_f32_16x16x4_f32 -- bitwise an fp32 fmaf chain over
 *                      the pixels of a wave.  Matrix-bound at CS = 64 (0.57 - 0.59 of the roofline against 0.72 - 0.77 for the split). */
#define DFX_MFMA_F32_CHAIN 0
#define DFX_MFMA_BF16X3 1
#define DFX_MFMA_AUTO 2
#define DFX_AUTO_BF16X3_MIN_CS 16
DFX_API int dfx_set_mfma_mode(dfx_ctx* ctx, int mode);
/* *mode = the evaluation mode (DFX_MFMA_F32_CHAIN / DFX_MFMA_BF16X3) the context's last SfM / DepthAligner step resolved to. */
DFX_API int dfx_last_mfma_mode(dfx_ctx* ctx, int* mode);
/* No environment variable steers a context: evaluation mode, schedule, wait mode and descriptor paths are set through dfx_set_mfma_mode /
 * dfx_set_schedule / dfx_set_result_wait / dfx_ctx_configure only, so a drop-in build cannot change result bits behind its caller's back.
 * (Rounds 3-5 read DFX_MFMA, DFX_SCHEDULE, DFX_POLL_RESULT, DFX_*_DESC_ZEROCOPY and five tuning aids from the environment; removed in round 6.
 * The one variable left is DFX_RCCL_LIB, the path of the RCCL library to load -- deployment, not numerics; see "multi-GPU exchange".) */
/* Launch schedule of the batched SfM step (no reference counterpart: the reference has one fixed 11 x 32 grid, cu_sfmaligner.cpp:60).
 *  DFX_SCHEDULE_AUTO     (default) the library's choice -- today always the static partition.
 *  DFX_SCHEDULE_STATIC   the static partition: bit-reproducible for a given launch shape, like the reference's fixed grid.
 *  DFX_SCHEDULE_DYNAMIC  opt-in: whenever the launch is structurally able to (W % 64 == 0, unpadded Jacobian rows, no explicit
 *                        step_blocks) the batch runs on resident wave-workers that pop work items from per-pair queues, so no slot
 *                        idles behind the hardware's uneven wave progress.  Which items a wave sums is decided at run time: results
 *                        are reproducible to fp32 re-association (~1e-7 relative), not bit for bit.  Worth trying for >= 128 pairs
 *                        (a pair's team is then <= 32 waves): -1.5 % to +4.5 % kernel time depending on the box; with fewer pairs the
 *                        queue heads are contended and it is up to 3x SLOWER than the static launch.
 * dfx_last_schedule: *dynamic = 1 when the context's last batched SfM step ran on the queues. */
#define DFX_SCHEDULE_AUTO 0
#define DFX_SCHEDULE_STATIC 1
#define DFX_SCHEDULE_DYNAMIC 2
DFX_API int dfx_set_schedule(dfx_ctx* ctx, int mode);
DFX_API int dfx_last_schedule(dfx_ctx* ctx, int* dynamic);
/* How the context's blocking calls wait (see "How the blocking single-result entries return" below): DFX_WAIT_POLL = the host polls a word in mapped memory
 * (the default), DFX_WAIT_STREAM = hipStreamSynchronize. */
#define DFX_WAIT_STREAM 0
#define DFX_WAIT_POLL 1
DFX_API int dfx_set_result_wait(dfx_ctx* ctx, int mode);
/* Per-context switches that survived their A/B (value 0 / 1; an unknown option or value is an error).  Neither changes a result bit
 * (tests/test_gpu_desc_paths.py): they choose where a batched launch reads its descriptor array from.
 *  DFX_OPT_SIMPLE_DESC_ZEROCOPY (default 1)  batched SE3 step / EvaluateError / UpdateDepth: the kernels read the descriptors straight
 *                                            out of the pinned, mapped staging slot instead of a device copy (-7 / -3.5 us per call of 128 pairs).
 *                                            (The pyramid build always does for its first launch, which mirrors them to device memory for the later ones.)
 *  DFX_OPT_STEP_DESC_ZEROCOPY   (default 0)  the same for the batched SfM step: -8 us outside the kernel, +4 us inside it (3840 long-lived
 *                                            workgroups read 1.5 MB over PCIe); pays only when the tail runs on its own stream */
#define DFX_OPT_SIMPLE_DESC_ZEROCOPY 1
#define DFX_OPT_STEP_DESC_ZEROCOPY 2
DFX_API int dfx_ctx_configure(dfx_ctx* ctx, int option, int value);
/* Measurement hook (no reference counterpart; the reference times with std::clock around blocking calls,
 * tools/kernel_benchmark.cpp:145-180): when enabled, every SfM step launch -- and every batched SE3-step / EvaluateError launch
 * (dfx_se3_step_batch*, dfx_sfm_error_batch*) -- is bracketed by HIP events on the context's stream: around the step (reduction)
 * kernel only, excluding the finalize kernel and copies.
 * dfx_profile_read waits for the stream, returns the number of bracketed launches and their summed
 * duration in milliseconds since the last read, and resets the counters. */
DFX_API int dfx_set_profiling(dfx_ctx* ctx, int enable);
DFX_API int dfx_profile_read(dfx_ctx* ctx, int* n_launches, double* total_ms);
/* Same, plus the shortest and the longest bracketed launch (either may be NULL). */
DFX_API int dfx_profile_read_ex(dfx_ctx* ctx, int* n_launches, double* total_ms, double* min_ms, double* max_ms);
/* Debug: copies the first `bytes` of the workgroup-partials scratch of the last launch to the host. */
DFX_API int dfx_debug_read_partials(dfx_ctx* ctx, void* host, size_t bytes);

/* ---- device images owned through the library (what include/dfx_host.hpp's keyframe store is made of; replaces the device side of
 * cuda/synced_pyramid.h:30-217 / vc::Image2DManaged).  elem_bytes = 4 (float images; prx_jac has w = W*CS) or 8 (gradients).  Rows
 * are 16-byte aligned and otherwise unpadded, so a Jacobian allocated here takes the dense-stream kernel variant.  upload /
 * download block like the reference's copyFrom; fill only enqueues. */
/* valid0 maps allocated HERE get a shadow (1 bit per pixel, "known to hold 1.0", created when the image is first handed to an SfM step as
 * valid0, kept consistent by dfx_img_fill_f32 / dfx_img_upload / every kernel the library launches into the image): the step kernel then
 * reads 8 bytes per 64-pixel chunk of it instead of the map's 256 to learn that nothing is left to write -- the map is a write-only output
 * of the path (dense_sfm.h:161; all ones from the keyframe build on, mapper.cpp:937), and that read was 2.7 % of the kernel's HBM traffic.
 * Contract: the memory of a library-owned image is written only through this API (the whole image view, as returned).  valid0 maps in
 * foreign memory (hipMalloc of the caller, a VisionCore buffer, a torch tensor) have writers the library cannot see; they are read every
 * step as before (same results, 4 B/px more traffic). */
DFX_API int dfx_img_alloc(dfx_ctx* ctx, uint32_t w, uint32_t h, size_t elem_bytes, dfx_img* out);
DFX_API int dfx_img_free(dfx_ctx* ctx, dfx_img* img);
DFX_API int dfx_img_upload(dfx_ctx* ctx, const dfx_img* dst, const void* host, size_t host_pitch_bytes, size_t elem_bytes);
DFX_API int dfx_img_download(dfx_ctx* ctx, const dfx_img* src, void* host, size_t host_pitch_bytes, size_t elem_bytes);
DFX_API int dfx_img_fill_f32(dfx_ctx* ctx, const dfx_img* dst, float value);
/* Page-locked host memory for LARGE results (no reference counterpart: its host buffers are pageable and its results are 100-byte items).  The rows of a
 * whole graph's SparseGeometricFactors are 150 MB per round at 1024 factors; into pageable memory that copy runs at ~5 GB/s and costs more than every kernel
 * of the round together, into a buffer from here at link speed.  Any host pointer is accepted by the blocking entries; this only makes the copy fast. */
DFX_API int dfx_host_alloc(dfx_ctx* ctx, size_t bytes, void** out);
DFX_API int dfx_host_free(dfx_ctx* ctx, void* ptr);
/* Debug / tests: copies the valid0 shadow of a library-owned image to the host (one uint64 per 64 pixels of the linear index y*w + x,
 * bit = pixel known to hold 1.0).  *n_words = 0 when the image has no shadow (foreign memory, or never used as valid0). */
DFX_API int dfx_debug_read_valid0_shadow(dfx_ctx* ctx, const dfx_img* img, uint64_t* host_words, size_t cap_words, size_t* n_words);

/* ---- SE3Aligner<float> (cuda/cu_se3aligner.h:52-72) -----------------------------------------
 * Accuracy statement shared by RunStep, EvaluateError and the tracker (the "row walk" kernels): the inlier set is exactly the reference's (pixels whose
 * validity margin is within a rigorous error bound of zero are re-evaluated in the reference's operation order); the sampled position differs from the
 * reference's in two documented ways, both far below the stated tolerance: the projection is evaluated with fused multiply-adds and one reciprocal
 * (~3e-5 pixel), and a tap coordinate less than 2^-13 pixel BELOW an integer is taken as that integer (at the exact identity u = x + rounding noise; the
 * fp64 oracle has u = x there).  The camera of a call must have the size of the images it is used with (w, h of dfx_cam = image width, height: the camera
 * of that pyramid level, camera_pyramid.h:41-46); other combinations are refused.
 * Run-ahead of the *_batch_async entries: each call holds one of 8 pinned descriptor slots until its kernels have run, so the 9th call in flight blocks
 * the host until the first has finished (bounded run-ahead; the blocking forms are unaffected). */
/* RunStep (cu_se3aligner.cpp:153-176): out_item = JTJJrReductionItem<float,6> on the HOST (120 bytes). */
DFX_API int dfx_se3_step(dfx_ctx* ctx, const dfx_se3* pose_10, const dfx_cam* cam, const dfx_img* img0,
                         const dfx_img* img1, const dfx_img* dpt0, const dfx_img* grad1, float huber_delta,
                         void* out_item);
/* Warp (cu_se3aligner.cpp:125-151): renders img1 into frame 0 (img2), signed residual sum + inliers. */
DFX_API int dfx_se3_warp(dfx_ctx* ctx, const dfx_se3* pose_10, const dfx_cam* cam, const dfx_img* img0,
                         const dfx_img* img1, const dfx_img* dpt0, const dfx_img* img2_out, dfx_corr_item* out);

/* How the blocking single-result entries return (dfx_se3_step, dfx_se3_warp, dfx_sfm_error, dfx_sfm_step / dfx_sfm_step_batch and dfx_sfm_linearize_batch
 * with n = 1, dfx_depth_aligner_step, dfx_squared_error, dfx_track_frame): the call's last kernel stores the result into pinned, device-mapped host memory
 * and, behind it (system-scope release), the call's sequence number; the host polls that word and copies the result out -- it does NOT wait for the stream
 * to report idle, which the runtime learns 3-6 us later (profiles/r05_poll_result.txt; dfx_set_result_wait switches per context).  Everything the call enqueued in front of that kernel has completed
 * when the call returns; the stream itself may show busy for a few more microseconds.  A stream error or a launch that never writes the word is detected
 * (the stream is queried every few hundred microseconds of polling).  dfx_set_result_wait(ctx, DFX_WAIT_STREAM) restores hipStreamSynchronize. */
/* ---- CameraTracker::TrackFrame (core/system/camera_tracker.cpp:42-71), device-resident -------
 * The reference loops on the host: RunStep (kernel + finalize + sync + 120-byte copy) -> 6x6 ldlt().solve -> retract, per
 * iteration.  Here the whole coarse-to-fine schedule is enqueued at once, one launch per iteration: the pose stays in device memory, and
 * every workgroup of an iteration first folds the previous evaluation's partial sums (double, fixed order), solves the 6x6 system (LDL^T
 * in double) and applies the update (t += dt, R = exp(dw) R) -- all workgroups compute the same bits, so nothing is handed over across the
 * grid --, then evaluates its rows at the new pose.  The last update is stored into mapped host memory; the call returns after one stream wait.
 * levels[0] is the finest level; levels are processed from n_levels-1 down to 0 with levels[l].iterations steps each. */
typedef struct dfx_track_level {
  dfx_cam cam;
  dfx_img img0, img1, dpt0, grad1; /* keyframe image, live image, keyframe depth, live gradient at this level */
  int32_t iterations;
} dfx_track_level;
typedef struct dfx_track_result {
  dfx_se3 pose_ck;       /* updated estimate */
  float inliers_frac;    /* inliers / area of the last level-0 iteration (camera_tracker.cpp:67) */
  float error;           /* residual / inliers of that iteration, +inf if no inliers (camera_tracker.cpp:68) */
  float residual;
  uint64_t inliers;
  int32_t iterations;        /* total Gauss-Newton iterations run */
  int32_t solver_failures;   /* iterations skipped because the 6x6 system was singular / had no inliers */
} dfx_track_result;
DFX_API int dfx_track_frame(dfx_ctx* ctx, const dfx_se3* pose_ck_init, const dfx_track_level* levels, int n_levels,
                            float huber_delta, dfx_track_result* out);
/* N independent trackers in one schedule of launches (grid.y = candidate): DeepFactors::Relocalize tracks the live frame
 * against every keyframe and keeps the smallest error (core/deepfactors.cpp:713-743), the loop detector's geometry check does
 * the same over its candidates (core/system/loop_detector.cpp:146-167).  levels[k * n_levels + l] = level l of candidate k;
 * all candidates share the iteration schedule and the image sizes of candidate 0.  out[k] as for dfx_track_frame. */
DFX_API int dfx_track_frame_batch(dfx_ctx* ctx, int n, const dfx_se3* pose_ck_init, const dfx_track_level* levels, int n_levels,
                                  float huber_delta, dfx_track_result* out);

/* ---- SfmAligner<float,CS> (cuda/cu_sfmaligner.h:50-97) -------------------------------------- */
/* RunStep (cu_sfmaligner.cpp:149-185).  cs in {16, 32, 64}.  std0 may be NULL (dead input in the reference,
 * dense_sfm.h:58-67); valid0 may be NULL (otherwise written 1.0 where a pixel is an inlier, never cleared,
 * dense_sfm.h:161).  out_item = JTJJrReductionItem<float,12+cs> on the HOST. */
DFX_API int dfx_sfm_step(dfx_ctx* ctx, int cs, const dfx_se3* pose0, const dfx_se3* pose1, const dfx_cam* cam,
                         const dfx_sfm_params* params, const dfx_img* img0, const dfx_img* img1, const dfx_img* dpt0,
                         const dfx_img* std0, const dfx_img* valid0, const dfx_img* prx0_jac, const dfx_img* grad1,
                         void* out_item);
/* EvaluateError (cu_sfmaligner.cpp:120-147): border 1, min_dpt 0 (FindCorrespondence defaults, dense_sfm.h:91). */
DFX_API int dfx_sfm_error(dfx_ctx* ctx, const dfx_se3* pose0, const dfx_se3* pose1, const dfx_cam* cam,
                          const dfx_sfm_params* params, const dfx_img* img0, const dfx_img* img1, const dfx_img* dpt0,
                          const dfx_img* std0, const dfx_img* grad1, dfx_corr_item* out);

/* n independent SE3Aligner::RunStep of one image size in ONE launch (new): item p (120 bytes, JTJJrReductionItem<float,6>) is written to
 * (char*)out_items + p * dfx_item_size(6).  Same per-pair arithmetic and reduction order as dfx_se3_step launched with the same number of
 * workgroups; the reference steps one frame against one keyframe per blocking call (cu_se3aligner.cpp:153-176). */
typedef struct dfx_se3_pair {
  dfx_se3 pose_10;
  dfx_cam cam;
  dfx_img img0, img1, dpt0, grad1;
} dfx_se3_pair;
DFX_API int dfx_se3_step_batch_async(dfx_ctx* ctx, const dfx_se3_pair* pairs, int n, float huber_delta, void* out_items_dev);
DFX_API int dfx_se3_step_batch(dfx_ctx* ctx, const dfx_se3_pair* pairs, int n, float huber_delta, void* out_items_host);

/* One keyframe->frame pair of a batch: the argument list of SfmAligner::RunStep as a POD. */
typedef struct dfx_sfm_pair {
  dfx_se3 pose0, pose1;
  dfx_cam cam;
  dfx_img img0, img1, dpt0, valid0 /* ptr may be NULL */, prx0_jac, grad1;
} dfx_sfm_pair;

/* Batched RunStep over n independent pairs in ONE launch (new; the reference evaluates pairs one by one
 * from PhotometricFactor::linearize, photometric_factor.cpp:267-274).  Enqueues on the context's stream and
 * returns immediately; item p is written to DEVICE memory at (char*)out_items_dev + p*dfx_item_size(12+cs).
 * The pairs of a batch may differ in image size -- e.g. all pyramid levels of a factor set in ONE launch, as the reference's
 * relinearisation walks them (core/mapping/df_work.cpp:118-136, tools/kernel_benchmark.cpp:192-203): every pair then gets workgroups in
 * proportion to its pixel count (a 160x120 level 1/16 of a 640x480 one), the large pairs are dispatched first and the small levels
 * fill the tail of the launch.  Each pair's images must agree with its own img0 size.  (The dynamic schedule needs one size.) */
DFX_API int dfx_sfm_step_batch_async(dfx_ctx* ctx, int cs, const dfx_sfm_params* params, const dfx_sfm_pair* pairs,
                                     int n, void* out_items_dev);
/* Same, then copies the n items to `out_items_host` and waits. */
DFX_API int dfx_sfm_step_batch(dfx_ctx* ctx, int cs, const dfx_sfm_params* params, const dfx_sfm_pair* pairs, int n,
                               void* out_items_host);

/* Batched EvaluateError (new): PhotometricFactor::error -> RunWarping is one blocking SfmAligner::EvaluateError per factor in the reference
 * (core/gtsam/photometric_factor.cpp:61-81,197-216; cu_sfmaligner.cpp:120-147).  Here n pairs of one image size in ONE launch; only
 * pose0, pose1, cam, img0, img1, dpt0 of a dfx_sfm_pair are read (valid0 / prx0_jac / grad1 may be zeroed).  border 1, min_dpt 0 as in
 * dfx_sfm_error.  out_items[p] = CorrespondenceReductionItem of pair p (device memory for _async, host memory for the blocking form). */
DFX_API int dfx_sfm_error_batch_async(dfx_ctx* ctx, const dfx_sfm_params* params, const dfx_sfm_pair* pairs, int n, dfx_corr_item* out_items_dev);
DFX_API int dfx_sfm_error_batch(dfx_ctx* ctx, const dfx_sfm_params* params, const dfx_sfm_pair* pairs, int n, dfx_corr_item* out_items_host);

/* PhotometricFactor::RunAlignmentStep (core/gtsam/photometric_factor.cpp:225-293) over a batch: UpdateDepthMaps (:332-341: dpt0 =
 * decode(code0) with the keyframe's prx_orig / prx_jac) followed by SfmAligner::RunStep, for n pairs in two launches and no host
 * round trip in between.  pairs[p].dpt0 is the keyframe's depth map (written, then read); prx0_orig[p] its zero-code proximity;
 * codes0[p*cs ..) (HOST) its code; params->avg_dpt the decoder's scale.  Every DISTINCT depth map of the batch is decoded once
 * -- the reference decodes it again for every factor that shares the keyframe -- so pairs sharing dpt0 must carry the same code
 * and decoder images (checked).  Why the decode is not folded into the step kernel itself is in DESIGN.md section 3.3. */
DFX_API int dfx_sfm_linearize_batch_async(dfx_ctx* ctx, int cs, const dfx_sfm_params* params, const dfx_sfm_pair* pairs,
                                          const dfx_img* prx0_orig, const float* codes0, int n, void* out_items_dev);
DFX_API int dfx_sfm_linearize_batch(dfx_ctx* ctx, int cs, const dfx_sfm_params* params, const dfx_sfm_pair* pairs,
                                    const dfx_img* prx0_orig, const float* codes0, int n, void* out_items_host);

/* ---- Gauss-Newton normal equations of a keyframe graph (new: SURVEY section 8e) ---------------------------------------
 * The reference hands each pair's 44x44 system to its own gtsam::HessianFactor keyed by (pose0, pose1, code0)
 * (core/gtsam/photometric_factor.cpp:105-180) and lets iSAM2 add the factors; its graph links arbitrary keyframe -> frame
 * pairs, in both directions (core/mapping/mapper.cpp:308-311).  A dfx_graph fixes that structure once: n_nodes nodes
 * (keyframes / frames; node n owns D = 6 + cs unknowns: pose, code) and n_pairs pairs, pair p = (keyframe node, frame node) =
 * pair_nodes[2p], pair_nodes[2p+1] (a HOST array).  The assembled system is ONE flat float buffer -- the unit the multi-GPU
 * path reduces with RCCL --
 *     Hd [n_nodes][D][D]   diagonal blocks: G11, G13, G33 of the pairs out of a node, G22 of the pairs into it
 *     Ho [n_pairs][D][6]   off-diagonal block of pair p: rows = (pose | code) of its keyframe node, columns = pose of its frame
 *                          node (G12 over G23^T, photometric_factor.cpp:135-161)
 *     g  [n_nodes][D]      Jtr blocks g1 / g3 and g2 (not negated; the factor negates, photometric_factor.cpp:106)
 * dfx_graph_assemble_async overwrites the WHOLE buffer with the contribution of the pairs [first_pair, first_pair + n_local)
 * whose items (JTJJrReductionItem<float,12+cs>, device memory, item l at l * dfx_item_size(12+cs)) this rank holds: every
 * node's incident pairs are summed in ascending pair order in double and written once (no atomics: bit-reproducible; with
 * all items on one rank -- gather mode -- independent of the world size).  Enqueue only (on the tail stream when one is set). */
typedef struct dfx_graph dfx_graph;
DFX_API int dfx_graph_create(dfx_ctx* ctx, int cs, int n_nodes, int n_pairs, const int32_t* pair_nodes, dfx_graph** out);
DFX_API void dfx_graph_destroy(dfx_graph* graph);
DFX_API size_t dfx_graph_system_floats(const dfx_graph* graph);
DFX_API int dfx_graph_assemble_async(dfx_ctx* ctx, const dfx_graph* graph, const void* items_dev, int first_pair, int n_local,
                                     float* sys_dev);
/* dfx_sfm_step_batch_async + dfx_graph_assemble_async of the same pairs (pairs[l] = pair first_pair + l of the graph) as ONE call:
 * same items, same system, bit for bit -- but the reduction tail of the launch is one kernel instead of two: the workgroup that sums
 * a pair's partials writes its item and its off-diagonal block, and the last of a node's pairs to arrive gathers that node's diagonal
 * block and gradient (ascending pair order, double, as above).  128 pairs of 640x480, CS = 32: 15 + 6.5 us of kernels and one launch
 * boundary become one kernel.  Launches without that tail kernel (a single pair; the fp32 chain) run the two
 * kernels one after the other.  Enqueue only. */
DFX_API int dfx_sfm_step_batch_assemble_async(dfx_ctx* ctx, int cs, const dfx_sfm_params* params, const dfx_sfm_pair* pairs, int n,
                                              void* out_items_dev, const dfx_graph* graph, int first_pair, float* sys_dev);

/* ---- multi-GPU exchange (new: SURVEY section 8e; the reference is single-GPU) ---------------------------------------
 * One process per GPU; every rank evaluates a contiguous shard of the pair list (dfx_shard_range) with keyframe pyramids replicated,
 * then ONE collective over RCCL / xGMI brings the results together:
 *   reduce mode  dfx_graph_assemble_async (this rank's pairs) -> dfx_graph_reduce_async: the ranks' flat systems are summed in place onto
 *                `root` (ncclReduce), or onto every rank when root < 0 (ncclAllReduce).  1.3 MB for 64 keyframes / 1024 pairs at CS = 32.
 *   gather mode  dfx_items_all_gather_async: every rank contributes bytes_per_rank bytes of items (its shard, padded to the largest shard)
 *                and receives world * bytes_per_rank bytes in rank order -- the host can then emit one gtsam::HessianFactor per pair
 *                exactly as the reference does (core/gtsam/photometric_factor.cpp:180).
 * A communicator is created collectively: rank 0 calls dfx_comm_get_unique_id (DFX_COMM_ID_BYTES bytes), the host program hands the id to
 * the other ranks by its own means (MPI, a socket, a file), every rank calls dfx_comm_create with the context of its GPU.  The collectives
 * only enqueue, on the stream the step's results are complete on (the context's tail stream when one is set, else its stream).  librccl
 * is loaded at run time on first use (a copy the process already holds is reused; DFX_RCCL_LIB names a specific library); single-GPU
 * users never load it. */
typedef struct dfx_comm dfx_comm;
#define DFX_COMM_ID_BYTES 128
DFX_API int dfx_comm_get_unique_id(void* id_out);
DFX_API int dfx_comm_create(dfx_ctx* ctx, const void* id, int rank, int world, dfx_comm** out);
DFX_API void dfx_comm_destroy(dfx_comm* comm);
DFX_API int dfx_comm_rank(const dfx_comm* comm);
DFX_API int dfx_comm_world(const dfx_comm* comm);
/* The contiguous shard [first, first + count) of n pairs that `rank` of `world` evaluates (the first n % world ranks hold one more). */
DFX_API int dfx_shard_range(int n, int rank, int world, int* first, int* count);
DFX_API int dfx_graph_reduce_async(dfx_ctx* ctx, dfx_comm* comm, const dfx_graph* graph, float* sys_dev, int root);
/* The collective underneath: n floats summed in place over the ranks (what dfx_graph_reduce_async does with dfx_graph_system_floats(graph)). */
DFX_API int dfx_comm_reduce_f32_async(dfx_ctx* ctx, dfx_comm* comm, float* buf_dev, size_t n, int root);
DFX_API int dfx_items_all_gather_async(dfx_ctx* ctx, dfx_comm* comm, const void* items_local_dev, size_t bytes_per_rank, void* items_all_dev);
/* Replication of keyframe buffers (the pyramids of core/mapping/keyframe.h:46-56: img, dpt, vld, stdev, prx_orig, jac [H][W*CS], plus code and
 * pose): `bytes` bytes at buf_dev of rank `root` overwrite the same bytes on every other rank (ncclBroadcast, in place).  Every rank calls it with
 * a buffer of the same size -- pitched images as pitch_bytes * h.  Ordered on the context's MAIN stream (it rewrites inputs of later launches),
 * not on the tail stream.  include/dfx_host.hpp `dfx::KeyframeBroadcast` walks a keyframe's buffers; one call per buffer, ~60 MB per keyframe
 * at 640x480x32. */
DFX_API int dfx_comm_broadcast_async(dfx_ctx* ctx, dfx_comm* comm, void* buf_dev, size_t bytes, int root);

/* ---- image-proc free functions (cuda/cu_image_proc.h:27-46) ---------------------------------- */
/* UpdateDepth (cu_image_proc.cpp:248-277): dpt = a/(prx_orig + prx_jac . code) - a; code is a HOST array of cs floats. */
DFX_API int dfx_update_depth(dfx_ctx* ctx, int cs, const float* code, const dfx_img* prx_orig, const dfx_img* prx_jac,
                             float avg_dpt, const dfx_img* dpt_out);
/* n independent UpdateDepth jobs of one image size in ONE launch (new; Mapper::UpdateMap re-decodes every changed keyframe level by
 * level, core/mapping/mapper.cpp:860-888): job k uses codes[k*cs .. k*cs+cs) (HOST), prx_orig[k], prx_jac[k], dpt_out[k].  Enqueue only. */
DFX_API int dfx_update_depth_batch_async(dfx_ctx* ctx, int cs, int n, const float* codes, const dfx_img* prx_orig, const dfx_img* prx_jac,
                                         float avg_dpt, const dfx_img* dpt_out);
/* SobelGradients (cu_image_proc.cpp:57-112): grad = (gx, gy)/8 with clamped borders. */
DFX_API int dfx_sobel_gradients(dfx_ctx* ctx, const dfx_img* img, const dfx_img* grad_out);
/* GaussianBlurDown (cu_image_proc.cpp:134-186): 5x5 binomial + decimate by 2; out is (w/2, h/2). */
DFX_API int dfx_gaussian_blur_down(dfx_ctx* ctx, const dfx_img* in, const dfx_img* out);
/* Frame::FillPyramids (core/mapping/frame.h:80-94: level i = GaussianBlurDown(level i-1), SobelGradients on every level) and
 * DeepFactors::UploadLiveFrame (core/deepfactors.cpp:616-630) for n frames in ONE enqueue (new): the reference builds L images + L gradients
 * per frame with 2L - 1 blocking single-image calls at camera rate.  Here ONE launch per pyramid level over all frames of the batch; a level is
 * read once -- its Sobel gradient and its blur-down to the next level come from the same LDS tile.  img[0] is the input (device memory, read),
 * img[1 .. levels-1] and grad[0 .. levels-1] are written; level i + 1 must be (w/2, h/2) of level i; a grad view with a null ptr skips that
 * level's gradient (UploadLiveFrame leaves level 0 out, deepfactors.cpp:620-625).  All frames of a batch share the sizes of frame 0.
 * Same bits as the per-level calls dfx_gaussian_blur_down / dfx_sobel_gradients.  _async only enqueues; dfx_build_pyramid blocks like the
 * reference's calls. */
/* Debug (host arithmetic only, no device needed): the shape dfx_build_pyramid_batch_async gives the row-streaming launch of a level of `n` frames of w x h on a
 * device of `cus` compute units -- rows per segment and workgroups of the launch (tests/test_pyramid_shape.py: the segments tile the height, and where the frames
 * allow it the workgroups tile the compute units exactly).  w must be even (the row-streaming kernel's precondition). */
DFX_API int dfx_debug_pyramid_launch_shape(int w, int h, int n, int cus, int* rows_per_segment, int* workgroups, int* waves_per_workgroup);
#define DFX_MAX_PYR_LEVELS 8
typedef struct dfx_pyramid {
  int32_t levels;
  dfx_img img[DFX_MAX_PYR_LEVELS];
  dfx_img grad[DFX_MAX_PYR_LEVELS];
} dfx_pyramid;
DFX_API int dfx_build_pyramid_batch_async(dfx_ctx* ctx, const dfx_pyramid* frames, int n);
DFX_API int dfx_build_pyramid(dfx_ctx* ctx, const dfx_pyramid* frame);
/* SquaredError (cu_image_proc.cpp:190-240): sum (a-b)^2. */
DFX_API int dfx_squared_error(dfx_ctx* ctx, const dfx_img* a, const dfx_img* b, float* out);

/* ---- SparseGeometricFactor<float,CS>::linearize (core/gtsam/sparse_geometric_factor.cpp:147-275) ----
 * The reference evaluates the N sampled points on the CPU (forcing a device->host sync of both keyframes' 39 MB Jacobians).
 * Here: CS / 4 lanes per point, everything read from device memory.  `points_xy` is a HOST array of N (x, y) int pairs
 * (uniform_sampler.h:28-32), code0/code1 are HOST arrays of cs floats, kf0 = {prx_orig, prx_jac}, kf1 = {prx_orig, prx_jac,
 * dpt_grad} with dpt_grad the Sobel gradient of kf1's depth (mapper.cpp:998-1000).  rows_host receives the N x (12 + 2 cs + 1)
 * row-major matrix [A0 | A1 | A2 | A3 | b] of the gtsam::JacobianFactor (all-zero rows for points without a correspondence).
 * avg_dpt is 2.0 in the reference (:170).  One factor per blocking call (the reference's pattern): a batch of one of the entry below. */
DFX_API int dfx_sparse_geometric_linearize(dfx_ctx* ctx, int cs, const dfx_se3* pose0, const dfx_se3* pose1, const float* code0,
                                           const float* code1, const dfx_cam* cam, const int32_t* points_xy, int n_points,
                                           const dfx_img* prx0_orig, const dfx_img* prx0_jac, const dfx_img* prx1_orig,
                                           const dfx_img* prx1_jac, const dfx_img* dpt1_grad, float huber_delta, float avg_dpt,
                                           float* rows_host);
/* ALL sparse geometric factors of a relinearisation round in ONE launch (new).  The reference linearises every factor of the graph inside one
 * ISAM2::update -- a SparseGeometricFactor per keyframe pair (core/mapping/mapper.cpp:308-311), each a CPU loop over its points
 * (sparse_geometric_factor.cpp:147-275) -- so a 16-keyframe window with geo_npoints = 500 is 120 factors x 500 points per round; one blocking
 * call per factor costs 65 us each (launch + copies), 7.8 ms per round, against 0.9 ms for the round's whole DENSE photometric part.
 * Factor k's rows ([n_points][12 + 2 cs + 1], as above) start at row sum_{j<k} factors[j].n_points of the output.  The rows of a factor are
 * the bytes the single-factor call returns (same kernel).
 *   _async   rows stay in DEVICE memory at rows_dev (enqueue only: descriptors and host-resident point lists ride one host-to-device copy)
 *   blocking rows_host (HOST) receives all rows in one device-to-host copy (18 MB for 120 x 500 points at cs = 32: the copy, not the kernel,
 *            is then the cost of the call).
 * points_xy of a factor may live in DEVICE memory (points_on_device != 0): the reference samples a factor's points once, in its constructor
 * (sparse_geometric_factor.cpp:50-53; again per linearize only in its stochastic mode, :153-157), so a caller can upload them once; the host cannot range-check those, the kernel clamps them into
 * the image. */
typedef struct dfx_sparse_geo_factor {
  dfx_se3 pose0, pose1;
  dfx_cam cam;
  const float* code0;        /* HOST, cs floats */
  const float* code1;        /* HOST, cs floats */
  const int32_t* points_xy;  /* n_points (x, y) pairs: HOST memory, or DEVICE memory when points_on_device != 0 */
  int32_t n_points;
  int32_t points_on_device;
  dfx_img prx0_orig, prx0_jac, prx1_orig, prx1_jac, dpt1_grad;
} dfx_sparse_geo_factor;
DFX_API int dfx_sparse_geometric_linearize_batch_async(dfx_ctx* ctx, int cs, const dfx_sparse_geo_factor* factors, int n, float huber_delta,
                                                       float avg_dpt, float* rows_dev);
DFX_API int dfx_sparse_geometric_linearize_batch(dfx_ctx* ctx, int cs, const dfx_sparse_geo_factor* factors, int n, float huber_delta,
                                                 float avg_dpt, float* rows_host);
/* The same round with the factors' NORMAL EQUATIONS as the result: per factor the upper triangle (row-major) of [A | b]^T [A | b], NC (NC + 1) / 2 floats with
 * NC = 12 + 2 CS + 1 -- entry (i, j), i <= j, at i NC - i (i - 1) / 2 + (j - i); columns 0..11 the two poses, 12..12+2CS-1 the two codes, column NC - 1 = b, so the last
 * column holds A^T b and b^T b.  It is what gtsam's Cholesky elimination forms from the JacobianFactor on the host (the reference hands it the rows,
 * sparse_geometric_factor.cpp:262-275); formed here on the device (fp32, rows added in ascending order: deterministic), a 1024-factor round returns 12 MB instead
 * of 157 MB.  The rows themselves stay in the context's scratch.  _async: gram_dev is device memory, enqueue only; the blocking form copies to the host. */
DFX_API int dfx_sparse_geometric_gram_batch_async(dfx_ctx* ctx, int cs, const dfx_sparse_geo_factor* factors, int n, float huber_delta, float avg_dpt, float* gram_dev);
DFX_API int dfx_sparse_geometric_gram_batch(dfx_ctx* ctx, int cs, const dfx_sparse_geo_factor* factors, int n, float huber_delta, float avg_dpt, float* gram_host);

/* ---- DepthAligner<float,CS>::RunStep (cuda/cu_depthaligner.cpp:32-110); avg_dpt is 2 in the reference. */
DFX_API int dfx_depth_aligner_step(dfx_ctx* ctx, int cs, const float* code, const dfx_img* target_dpt,
                                   const dfx_img* prx_orig, const dfx_img* prx_jac, float avg_dpt, void* out_item);

#ifdef __cplusplus
}
#endif
#endif /* DFX_H_ */
