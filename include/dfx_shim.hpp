// dfx_shim.hpp -- header-only C++17 host layer over the C ABI (include/dfx.h) that mirrors the reference's operator
// interface for the alignment hot path: same class / method names, argument order and error behaviour as
//   df::SfmAligner<float,CS>   sources/cuda/cu_sfmaligner.h:50-97
//   df::SE3Aligner<float>      sources/cuda/cu_se3aligner.h:40-88
//   df::DepthAligner<float,CS> sources/cuda/cu_depthaligner.h
//   df::UpdateDepth / SobelGradients / GaussianBlurDown / SquaredError   sources/cuda/cu_image_proc.h:27-46
//   df::JTJJrReductionItem / df::CorrespondenceReductionItem            sources/cuda/reduction_items.h:35-143
//
// The reference's argument types come from Sophus, Eigen and VisionCore, none of which exist in this build image.
// The shim is therefore written against three tiny *concepts* and ships POD models of them (namespace dfx::pod):
//   SE3-like     : .unit_quaternion().coeffs() -> x,y,z,w  and .translation() -> 3 floats      (Sophus::SE3f satisfies it)
//                  -- or a dfx::pod::SE3f {q[4], t[3]}
//   Camera-like  : fx() fy() u0() v0() width() height()                                        (df::PinholeCamera<float>)
//   Image-like   : ptr() pitch() width() height()                                              (vc::Image2DView<T,TargetDeviceCUDA>)
// When <Eigen/Core> is on the include path (it always is in the reference's build) the result items carry Eigen types exactly as
// the reference's do -- `JtJ.toDenseMatrix()` returns Eigen::Matrix<float,NP,NP>, `Jtr` is Eigen::Matrix<float,NP,1> -- so the two
// consumers that matter compile unchanged: `sys.JtJ.toDenseMatrix().template cast<double>()` / `-sys.Jtr.template cast<double>()`
// (core/gtsam/photometric_factor.cpp:105-106) and `-result.JtJ.toDenseMatrix().ldlt().solve(result.Jtr)`
// (core/system/camera_tracker.cpp:59).  Without Eigen the same members are std::array (row-major dense).  tests/cpp/shim_test.cpp
// reproduces those call sites against Sophus / Eigen / VisionCore-shaped types and runs them on the GPU.  INTEGRATION.md shows
// the CMake lines that swap libdf_cuda for this header + libdfx.so.
//
// Errors: every non-zero C-ABI status becomes a dfx::Error (std::runtime_error), mirroring vc::CUDAException thrown by
// CudaCheckLastError (sources/cuda/launch_utils.h:26-32); "no overlap" stays in-band as inliers == 0.
#pragma once
#include <array>
#include <cstdint>
#include <cstring>
#include <memory>
#include <stdexcept>
#include <string>
#include <type_traits>
#include <vector>

#include "dfx.h"

#if defined(__has_include)
#if __has_include(<Eigen/Core>)
#include <Eigen/Core>
#define DFX_SHIM_HAS_EIGEN 1
#endif
#endif

namespace dfx {

struct Error : std::runtime_error {
  int code;
  Error(int c, const std::string& m) : std::runtime_error("dfx error " + std::to_string(c) + ": " + m), code(c) {}
};
inline void check(int rc) {
  if (rc != DFX_OK) throw Error(rc, dfx_last_error());
}

// ---- POD models of the three concepts (used when Sophus / VisionCore are absent) -----------------------------------
namespace pod {
struct SE3f {
  float q[4] = { 0, 0, 0, 1 };   // x y z w
  float t[3] = { 0, 0, 0 };
};
struct PinholeCamera {
  float fx_, fy_, u0_, v0_, w_, h_;
  float fx() const { return fx_; } float fy() const { return fy_; }
  float u0() const { return u0_; } float v0() const { return v0_; }
  float width() const { return w_; } float height() const { return h_; }
};
template <typename T>
struct Image2DView {   // non-owning device view, like vc::Image2DView<T, TargetDeviceCUDA>
  T* ptr_ = nullptr;
  std::size_t pitch_ = 0, w_ = 0, h_ = 0;
  T* ptr() const { return ptr_; }
  std::size_t pitch() const { return pitch_; }
  std::size_t width() const { return w_; }
  std::size_t height() const { return h_; }
};
struct Grad2 { float gx, gy; };   // Eigen::Matrix<float,1,2>
}  // namespace pod

// ---- concept adapters ------------------------------------------------------------------------------------------------
namespace detail {
template <typename T, typename = void> struct has_unit_quaternion : std::false_type {};
template <typename T> struct has_unit_quaternion<T, std::void_t<decltype(std::declval<const T&>().unit_quaternion())>> : std::true_type {};

template <typename SE3>
inline dfx_se3 to_se3(const SE3& p) {
  dfx_se3 o;
  if constexpr (has_unit_quaternion<SE3>::value) {   // Sophus::SE3f
    const auto q = p.unit_quaternion();
    o.q[0] = q.x(); o.q[1] = q.y(); o.q[2] = q.z(); o.q[3] = q.w();
    const auto t = p.translation();
    o.t[0] = t[0]; o.t[1] = t[1]; o.t[2] = t[2];
  } else {
    std::memcpy(o.q, p.q, sizeof(o.q));
    std::memcpy(o.t, p.t, sizeof(o.t));
  }
  return o;
}
template <typename Cam>
inline dfx_cam to_cam(const Cam& c) {
  return dfx_cam{ (float)c.fx(), (float)c.fy(), (float)c.u0(), (float)c.v0(), (float)c.width(), (float)c.height() };
}
template <typename Img>
inline dfx_img to_img(const Img& im) {
  return dfx_img{ (void*)im.ptr(), (std::size_t)im.pitch(), (uint32_t)im.width(), (uint32_t)im.height() };
}
}  // namespace detail

// One context per host thread (the reference is single-threaded under slam_mutex_).
class Context {
 public:
  // device < 0: the calling thread's current HIP device; stream: a hipStream_t, nullptr = the default stream (the reference's)
  explicit Context(int device = -1, void* stream = nullptr) { check(dfx_ctx_create(device, stream, &ctx_)); }
  ~Context() { dfx_ctx_destroy(ctx_); }
  Context(const Context&) = delete;
  Context& operator=(const Context&) = delete;
  dfx_ctx* get() const { return ctx_; }
  int device() const { return dfx_ctx_device(ctx_); }
  void SetStream(void* stream) { check(dfx_ctx_set_stream(ctx_, stream)); }
  // no reference counterpart: how the step kernels evaluate their sums (DFX_MFMA_AUTO = the exact bf16 split is the default;
  // DFX_MFMA_F32_CHAIN pins the fmaf-chain bits) and which launch schedule batched steps use (DFX_SCHEDULE_*), see include/dfx.h
  void SetMfmaMode(int mode) { check(dfx_set_mfma_mode(ctx_, mode)); }
  void SetSchedule(int mode) { check(dfx_set_schedule(ctx_, mode)); }
  void SetResultWait(int mode) { check(dfx_set_result_wait(ctx_, mode)); }   // DFX_WAIT_POLL (default) / DFX_WAIT_STREAM
  // The thread's default context: created on first use on the thread's CURRENT device (one process per GPU: this rank's GPU),
  // default stream.  SetDefault installs another one (other device / stream) for the aligners and free functions constructed or
  // called afterwards on this thread.
  static std::shared_ptr<Context>& DefaultSlot() {
    static thread_local std::shared_ptr<Context> c;
    return c;
  }
  static std::shared_ptr<Context> Default() {
    std::shared_ptr<Context>& c = DefaultSlot();
    if (!c) c = std::make_shared<Context>(-1, nullptr);
    return c;
  }
  static void SetDefault(std::shared_ptr<Context> c) { DefaultSlot() = std::move(c); }

 private:
  dfx_ctx* ctx_ = nullptr;
};

}  // namespace dfx

namespace df {

// reduction_items.h:35-71
template <typename Scalar>
struct CorrespondenceReductionItem {
  Scalar residual = 0;
  std::size_t inliers = 0;
};

// reduction_items.h:77-143.  JtJ keeps VisionCore's packed row-major upper triangle; toDenseMatrix() mirrors it.
template <typename Scalar, int NP>
struct JTJJrReductionItem {
  static_assert(std::is_same<Scalar, float>::value, "the gfx950 kernels are fp32 (the reference instantiates float only)");
#ifdef DFX_SHIM_HAS_EIGEN
  typedef Eigen::Matrix<Scalar, NP, 1> JacobianType;
#else
  typedef std::array<Scalar, NP> JacobianType;
#endif
  struct HessianType {
#ifdef DFX_SHIM_HAS_EIGEN
    typedef Eigen::Matrix<Scalar, NP, NP> DenseMatrixType;
#else
    typedef std::array<Scalar, NP * NP> DenseMatrixType;   // row-major
#endif
    std::array<Scalar, NP*(NP + 1) / 2> coeff_{};
    const std::array<Scalar, NP*(NP + 1) / 2>& coeff() const { return coeff_; }
    Scalar operator()(int r, int c) const {
      if (r > c) std::swap(r, c);
      return coeff_[(std::size_t)r * NP - (std::size_t)r * (r - 1) / 2 + (c - r)];
    }
    DenseMatrixType toDenseMatrix() const {
      DenseMatrixType M;
#ifdef DFX_SHIM_HAS_EIGEN
      for (int r = 0; r < NP; ++r) for (int c = 0; c < NP; ++c) M(r, c) = (*this)(r, c);
#else
      for (int r = 0; r < NP; ++r) for (int c = 0; c < NP; ++c) M[(std::size_t)r * NP + c] = (*this)(r, c);
#endif
      return M;
    }
  };
  HessianType JtJ;
  JacobianType Jtr{};
  Scalar residual = 0;
  std::size_t inliers = 0;

  static JTJJrReductionItem FromRaw(const void* raw) {
    JTJJrReductionItem it;
    std::memcpy(it.JtJ.coeff_.data(), dfx_item_jtj(raw), sizeof(Scalar) * it.JtJ.coeff_.size());
    const float* g = dfx_item_jtr(raw, NP);
#ifdef DFX_SHIM_HAS_EIGEN
    for (int i = 0; i < NP; ++i) it.Jtr(i) = g[i];
#else
    for (int i = 0; i < NP; ++i) it.Jtr[i] = g[i];
#endif
    it.residual = dfx_item_residual(raw, NP);
    it.inliers = (std::size_t)dfx_item_inliers(raw, NP);
    return it;
  }
};

// dense_sfm.h:36-43
struct DenseSfmParams {
  float huber_delta = 0.1f;
  float ocl_th = 1000;   // unused by the reference kernels; kept for source compatibility
  float avg_dpt = 2.0f;
  float min_dpt = 0.0f;
  int valid_border = 2;
};

// cu_sfmaligner.h:41-48.  Threads are fixed at 256 (4 wave64) on gfx950; step_blocks = workgroups per pair, 0 = auto.
struct SfmAlignerParams {
  DenseSfmParams sfmparams;
  int step_threads = 256;
  int step_blocks = 0;
  int eval_threads = 256;
  int eval_blocks = 0;
};

template <typename Scalar, int CS>
class SfmAligner {
 public:
  typedef std::shared_ptr<SfmAligner<Scalar, CS>> Ptr;
  typedef JTJJrReductionItem<Scalar, 12 + CS> ReductionItem;
  typedef CorrespondenceReductionItem<Scalar> ErrorReductionItem;

  explicit SfmAligner(SfmAlignerParams params = SfmAlignerParams(), std::shared_ptr<dfx::Context> ctx = dfx::Context::Default())
      : params_(params), ctx_(std::move(ctx)) {
    SetEvalThreadsBlocks(params.eval_threads, params.eval_blocks);
    SetStepThreadsBlocks(params.step_threads, params.step_blocks);
  }
  virtual ~SfmAligner() {}

  // cu_sfmaligner.cpp:120-147
  // Every image argument is its own deduced type: the reference's callers mix vc::Image2DView values, references and
  // Image2DManaged buffers (which convert to views there) in one call.
  template <typename SE3, typename Cam, typename Img0, typename Img1, typename Dpt0, typename Std0, typename GradBuffer>
  ErrorReductionItem EvaluateError(const SE3& pose0, const SE3& pose1, const Cam& cam, const Img0& img0,
                                   const Img1& img1, const Dpt0& dpt0, const Std0& std0,
                                   const GradBuffer& grad1) {
    const dfx_se3 p0 = dfx::detail::to_se3(pose0), p1 = dfx::detail::to_se3(pose1);
    const dfx_cam c = dfx::detail::to_cam(cam);
    const dfx_sfm_params prm = c_params();
    const dfx_img i0 = dfx::detail::to_img(img0), i1 = dfx::detail::to_img(img1), d0 = dfx::detail::to_img(dpt0);
    const dfx_img s0 = dfx::detail::to_img(std0), g1 = dfx::detail::to_img(grad1);
    dfx_corr_item out;
    dfx::check(dfx_sfm_error(ctx_->get(), &p0, &p1, &c, &prm, &i0, &i1, &d0, &s0, &g1, &out));
    return ErrorReductionItem{ out.residual, (std::size_t)out.inliers };
  }

  // cu_sfmaligner.cpp:149-185.  code0 is unused by the kernel (depth is already decoded), as in the reference.
  template <typename SE3, typename CodeT, typename Cam, typename Img0, typename Img1, typename Dpt0, typename Std0, typename Vld0, typename Jac0,
            typename GradBuffer>
  ReductionItem RunStep(const SE3& pose0, const SE3& pose1, const CodeT& /*code0*/, const Cam& cam, const Img0& img0,
                        const Img1& img1, const Dpt0& dpt0, const Std0& std0, Vld0& valid0,
                        const Jac0& prx0_jac, const GradBuffer& grad1) {
    const dfx_se3 p0 = dfx::detail::to_se3(pose0), p1 = dfx::detail::to_se3(pose1);
    const dfx_cam c = dfx::detail::to_cam(cam);
    const dfx_sfm_params prm = c_params();
    const dfx_img i0 = dfx::detail::to_img(img0), i1 = dfx::detail::to_img(img1), d0 = dfx::detail::to_img(dpt0);
    const dfx_img s0 = dfx::detail::to_img(std0), v0 = dfx::detail::to_img(valid0), jc = dfx::detail::to_img(prx0_jac);
    const dfx_img g1 = dfx::detail::to_img(grad1);
    std::vector<unsigned char> raw(dfx_item_size(12 + CS));
    dfx::check(dfx_sfm_step(ctx_->get(), CS, &p0, &p1, &c, &prm, &i0, &i1, &d0, s0.ptr ? &s0 : nullptr, v0.ptr ? &v0 : nullptr, &jc,
                            &g1, raw.data()));
    return ReductionItem::FromRaw(raw.data());
  }

  // ---- batched extension (no reference counterpart): one launch over n pairs of one pyramid level.
  // MakePair packs the RunStep argument list into the C POD; RunStepBatch returns the n items in order.
  template <typename SE3, typename Cam, typename Img0, typename Img1, typename Dpt0, typename Vld0, typename Jac0, typename GradBuffer>
  static dfx_sfm_pair MakePair(const SE3& pose0, const SE3& pose1, const Cam& cam, const Img0& img0, const Img1& img1,
                               const Dpt0& dpt0, const Vld0& valid0, const Jac0& prx0_jac, const GradBuffer& grad1) {
    dfx_sfm_pair p;
    p.pose0 = dfx::detail::to_se3(pose0); p.pose1 = dfx::detail::to_se3(pose1);
    p.cam = dfx::detail::to_cam(cam);
    p.img0 = dfx::detail::to_img(img0); p.img1 = dfx::detail::to_img(img1); p.dpt0 = dfx::detail::to_img(dpt0);
    p.valid0 = dfx::detail::to_img(valid0); p.prx0_jac = dfx::detail::to_img(prx0_jac); p.grad1 = dfx::detail::to_img(grad1);
    return p;
  }
  std::vector<ReductionItem> RunStepBatch(const std::vector<dfx_sfm_pair>& pairs) {
    const dfx_sfm_params prm = c_params();
    const std::size_t isz = dfx_item_size(12 + CS);
    std::vector<unsigned char> raw(isz * pairs.size());
    dfx::check(dfx_sfm_step_batch(ctx_->get(), CS, &prm, pairs.data(), (int)pairs.size(), raw.data()));
    std::vector<ReductionItem> out;
    out.reserve(pairs.size());
    for (std::size_t k = 0; k < pairs.size(); ++k) out.push_back(ReductionItem::FromRaw(raw.data() + k * isz));
    return out;
  }

  // for host layers built on the C ABI (include/dfx_host.hpp): this aligner's dfx_sfm_params and its context
  dfx_sfm_params Params() const { return c_params(); }
  dfx_ctx* ContextHandle() const { return ctx_->get(); }

  // cu_sfmaligner.cpp:187-203 (glog CHECK there, exception here)
  void SetEvalThreadsBlocks(int threads, int blocks) {
    if (threads % 64) throw dfx::Error(DFX_E_INVALID, "threads must be a multiple of 64 (CDNA wavefront)");
    params_.eval_threads = threads; params_.eval_blocks = blocks;
  }
  void SetStepThreadsBlocks(int threads, int blocks) {
    if (threads % 64) throw dfx::Error(DFX_E_INVALID, "threads must be a multiple of 64 (CDNA wavefront)");
    if (blocks < 0 || blocks > 65535) throw dfx::Error(DFX_E_INVALID, "blocks out of range [0, 65535]");
    params_.step_threads = threads; params_.step_blocks = blocks;   // travels with every call of THIS aligner (dfx_sfm_params.step_blocks)
  }

 private:
  dfx_sfm_params c_params() const {
    return dfx_sfm_params{ params_.sfmparams.huber_delta, params_.sfmparams.avg_dpt, params_.sfmparams.min_dpt, params_.sfmparams.valid_border,
                           params_.step_blocks };
  }
  SfmAlignerParams params_;
  std::shared_ptr<dfx::Context> ctx_;
};

template <typename Scalar>
class SE3Aligner {
 public:
  typedef std::shared_ptr<SE3Aligner<Scalar>> Ptr;
  typedef JTJJrReductionItem<Scalar, 6> ReductionItem;
  typedef CorrespondenceReductionItem<Scalar> CorrespondenceItem;

  explicit SE3Aligner(std::shared_ptr<dfx::Context> ctx = dfx::Context::Default()) : ctx_(std::move(ctx)) {}
  virtual ~SE3Aligner() {}

  // cu_se3aligner.cpp:125-151: renders img1 into frame 0 (img2)
  template <typename SE3, typename Cam, typename Img0, typename Img1, typename Dpt0, typename Img2>
  CorrespondenceItem Warp(const SE3& se3, const Cam& cam, const Img0& img0, const Img1& img1, const Dpt0& dpt0,
                          Img2& img2) {
    const dfx_se3 p = dfx::detail::to_se3(se3);
    const dfx_cam c = dfx::detail::to_cam(cam);
    const dfx_img i0 = dfx::detail::to_img(img0), i1 = dfx::detail::to_img(img1), d0 = dfx::detail::to_img(dpt0), i2 = dfx::detail::to_img(img2);
    dfx_corr_item out;
    dfx::check(dfx_se3_warp(ctx_->get(), &p, &c, &i0, &i1, &d0, &i2, &out));
    return CorrespondenceItem{ out.residual, (std::size_t)out.inliers };
  }

  // cu_se3aligner.cpp:153-176
  template <typename SE3, typename Cam, typename Img0, typename Img1, typename Dpt0, typename GradBuffer>
  ReductionItem RunStep(const SE3& se3, const Cam& cam, const Img0& img0, const Img1& img1, const Dpt0& dpt0,
                        const GradBuffer& grad1) {
    const dfx_se3 p = dfx::detail::to_se3(se3);
    const dfx_cam c = dfx::detail::to_cam(cam);
    const dfx_img i0 = dfx::detail::to_img(img0), i1 = dfx::detail::to_img(img1), d0 = dfx::detail::to_img(dpt0), g1 = dfx::detail::to_img(grad1);
    unsigned char raw[120];
    dfx::check(dfx_se3_step(ctx_->get(), &p, &c, &i0, &i1, &d0, &g1, huber_delta_, raw));
    return ReductionItem::FromRaw(raw);
  }

  void SetHuberDelta(float val) { huber_delta_ = val; }

 private:
  float huber_delta_ = 0.1f;   // cu_se3aligner.h:87
  std::shared_ptr<dfx::Context> ctx_;
};

template <typename Scalar, int CS>
class DepthAligner {
 public:
  typedef JTJJrReductionItem<Scalar, CS> ReductionItem;
  explicit DepthAligner(std::shared_ptr<dfx::Context> ctx = dfx::Context::Default()) : ctx_(std::move(ctx)) {}

  // cu_depthaligner.cpp:78-110 (avg_dpt hard-coded to 2 there)
  template <typename CodeT, typename Tgt, typename Prx, typename Jac>
  ReductionItem RunStep(const CodeT& code, const Tgt& target_dpt, const Prx& prx_orig, const Jac& prx_jac) {
    const dfx_img tg = dfx::detail::to_img(target_dpt), po = dfx::detail::to_img(prx_orig), jc = dfx::detail::to_img(prx_jac);
    if (jc.w / tg.w != (uint32_t)CS) throw dfx::Error(DFX_E_INVALID, "DepthAligner used with a different code size than it was compiled for");
    float cd[CS];
    for (int i = 0; i < CS; ++i) cd[i] = (float)code[i];
    std::vector<unsigned char> raw(dfx_item_size(CS));
    dfx::check(dfx_depth_aligner_step(ctx_->get(), CS, cd, &tg, &po, &jc, 2.0f, raw.data()));
    return ReductionItem::FromRaw(raw.data());
  }

 private:
  std::shared_ptr<dfx::Context> ctx_;
};

// ---- cu_image_proc.h:27-46 -------------------------------------------------------------------------------------------
// The reference's signatures, plus an optional trailing context (default: the thread's default context, i.e. the current
// device -- dfx::Context::SetDefault to use another device or stream).
template <typename T, int CS, typename ImageBuf, typename CodeT, typename PrxBuf, typename JacBuf>
void UpdateDepth(const CodeT& code, const PrxBuf& prx_orig, const JacBuf& prx_jac, T avg_dpt, ImageBuf& dpt_out,
                 const std::shared_ptr<dfx::Context>& ctx = dfx::Context::Default()) {
  float cd[CS];
  for (int i = 0; i < CS; ++i) cd[i] = (float)code[i];
  const dfx_img po = dfx::detail::to_img(prx_orig), jc = dfx::detail::to_img(prx_jac), out = dfx::detail::to_img(dpt_out);
  dfx::check(dfx_update_depth(ctx->get(), CS, cd, &po, &jc, (float)avg_dpt, &out));
}
#ifdef DFX_SHIM_HAS_EIGEN
// The reference's own callers name no template argument: `df::UpdateDepth(cde0, prx_orig, prx_jac, 2.0f, dpt)` (photometric_factor.cpp:337,
// mapper.cpp:883-886, 986-990) deduces T and CS from `const Eigen::Matrix<T,CS,1>& code` (cu_image_proc.h:40-45) -- found by compiling
// photometric_factor.cpp unmodified against this header (tests/cpp/ref_callers_test.cpp).
template <typename T, int CS, typename PrxBuf, typename JacBuf, typename ImageBuf>
void UpdateDepth(const Eigen::Matrix<T, CS, 1>& code, const PrxBuf& prx_orig, const JacBuf& prx_jac, T avg_dpt, ImageBuf& dpt_out,
                 const std::shared_ptr<dfx::Context>& ctx = dfx::Context::Default()) {
  UpdateDepth<T, CS, ImageBuf, Eigen::Matrix<T, CS, 1>, PrxBuf, JacBuf>(code, prx_orig, prx_jac, avg_dpt, dpt_out, ctx);
}
#endif
template <typename ImgBuf, typename GradBuf>
void SobelGradients(const ImgBuf& img, GradBuf& grad, const std::shared_ptr<dfx::Context>& ctx = dfx::Context::Default()) {
  const dfx_img i = dfx::detail::to_img(img), g = dfx::detail::to_img(grad);
  dfx::check(dfx_sobel_gradients(ctx->get(), &i, &g));
}
template <typename InBuf, typename OutBuf>
void GaussianBlurDown(const InBuf& in, OutBuf& out, const std::shared_ptr<dfx::Context>& ctx = dfx::Context::Default()) {
  const dfx_img i = dfx::detail::to_img(in), o = dfx::detail::to_img(out);
  dfx::check(dfx_gaussian_blur_down(ctx->get(), &i, &o));
}
template <typename Buf1, typename Buf2>
float SquaredError(const Buf1& buf1, const Buf2& buf2, const std::shared_ptr<dfx::Context>& ctx = dfx::Context::Default()) {
  const dfx_img a = dfx::detail::to_img(buf1), b = dfx::detail::to_img(buf2);
  float out = 0;
  dfx::check(dfx_squared_error(ctx->get(), &a, &b, &out));
  return out;
}

}  // namespace df
