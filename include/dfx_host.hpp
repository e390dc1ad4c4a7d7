// dfx_host.hpp -- header-only C++17 host layer above include/dfx_shim.hpp: the data model and the batching seam a maintainer
// links into the reference's mapper.  Everything is device-resident and goes through the C ABI; no
// HIP header is needed to compile it.
//
//   dfx::DeviceImage<T>, dfx::BufferPyramid<T>   replace the device side of cuda/synced_pyramid.h:30-217 (SyncedBufferPyramid:
//                                                a lazily mirrored CPU <-> GPU pyramid with dirty flags).  On an MI355X every level
//                                                of every keyframe stays in HBM (64 keyframes x 60 MB of 288 GB); host copies are
//                                                explicit (Download), nothing syncs behind the caller's back.
//   dfx::Frame, dfx::Keyframe<CS>                core/mapping/frame.h:36-120, keyframe.h:34-100: the buffers the path reads
//                                                (pyr_img, pyr_grad | pyr_dpt, pyr_vld, pyr_stdev, pyr_prx_orig, pyr_jac, dpt_grad,
//                                                code) with FillPyramids (frame.h:80-94) and the decoder hand-over /
//                                                UpdateDepthMaps of Mapper::BuildKeyframe (mapper.cpp:935-1000).
//   dfx::PhotometricFactor<CS>                   core/gtsam/photometric_factor.{h,cpp} minus GTSAM: error(), the linearisation
//                                                cache of GetJacobiansIfNeeded (:296-327), the residual rescaling (:209-216,
//                                                :275-282), and Hessian() = the G11..G33 / g1..g3 / f of linearize (:105-180).
//   dfx::LinearizeAll                            one relinearisation round over MANY factors: per pyramid level ONE
//                                                dfx_sfm_linearize_batch (UpdateDepthMaps once per keyframe + one batched RunStep),
//                                                then every factor's cache is seeded -- the per-factor linearize() calls iSAM2
//                                                makes afterwards launch nothing.
#pragma once
#include <array>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <map>
#include <memory>
#include <utility>
#include <vector>

#include "dfx_shim.hpp"

namespace dfx {

// ---- owning device image (vc::Image2DManaged<T, TargetDeviceCUDA>); converts to the view concept of the shim -------------------
template <typename T>
class DeviceImage {
  static_assert(sizeof(T) == 4 || sizeof(T) == 8, "float images and (gx, gy) gradient images");
 public:
  DeviceImage() = default;
  DeviceImage(std::size_t w, std::size_t h, std::shared_ptr<Context> ctx = Context::Default()) : ctx_(std::move(ctx)) {
    check(dfx_img_alloc(ctx_->get(), (uint32_t)w, (uint32_t)h, sizeof(T), &img_));
  }
  ~DeviceImage() { if (img_.ptr && ctx_) (void)dfx_img_free(ctx_->get(), &img_); }
  DeviceImage(const DeviceImage&) = delete;
  DeviceImage& operator=(const DeviceImage&) = delete;
  DeviceImage(DeviceImage&& o) noexcept : ctx_(std::move(o.ctx_)), img_(o.img_) { o.img_ = dfx_img{ nullptr, 0, 0, 0 }; }
  DeviceImage& operator=(DeviceImage&& o) noexcept { std::swap(ctx_, o.ctx_); std::swap(img_, o.img_); return *this; }

  // the Image-like concept of dfx_shim.hpp (vc::Image2DView accessors)
  T* ptr() const { return static_cast<T*>(img_.ptr); }
  std::size_t pitch() const { return img_.pitch_bytes; }
  std::size_t width() const { return img_.w; }
  std::size_t height() const { return img_.h; }
  std::size_t area() const { return (std::size_t)img_.w * img_.h; }
  const dfx_img& c_img() const { return img_; }

  void Upload(const T* host, std::size_t host_pitch_bytes = 0) {   // blocking, like copyFrom
    check(dfx_img_upload(ctx_->get(), &img_, host, host_pitch_bytes ? host_pitch_bytes : (std::size_t)img_.w * sizeof(T), sizeof(T)));
  }
  void Upload(const std::vector<T>& host) {
    if (host.size() != area()) throw Error(DFX_E_INVALID, "DeviceImage::Upload: size mismatch");
    Upload(host.data());
  }
  std::vector<T> Download() const {
    std::vector<T> host(area());
    check(dfx_img_download(ctx_->get(), &img_, host.data(), (std::size_t)img_.w * sizeof(T), sizeof(T)));
    return host;
  }
  void Fill(float v) { check(dfx_img_fill_f32(ctx_->get(), &img_, v)); }   // every float of the image (both components of a gradient)

 private:
  std::shared_ptr<Context> ctx_;
  dfx_img img_{ nullptr, 0, 0, 0 };
};

// ---- pyramid: level i is (w >> i) x (h >> i) elements of `elems_per_px` T each; GetGpuLevel is the reference's accessor name -----
template <typename T>
class BufferPyramid {
 public:
  BufferPyramid() = default;
  BufferPyramid(std::size_t levels, std::size_t w, std::size_t h, std::size_t elems_per_px = 1, std::shared_ptr<Context> ctx = Context::Default()) {
    for (std::size_t i = 0; i < levels; ++i) lv_.emplace_back((w >> i) * elems_per_px, h >> i, ctx);
  }
  std::size_t Levels() const { return lv_.size(); }
  DeviceImage<T>& GetGpuLevel(std::size_t i) { return lv_.at(i); }
  const DeviceImage<T>& GetGpuLevel(std::size_t i) const { return lv_.at(i); }
  DeviceImage<T>& operator[](std::size_t i) { return lv_.at(i); }
  const DeviceImage<T>& operator[](std::size_t i) const { return lv_.at(i); }
 private:
  std::vector<DeviceImage<T>> lv_;
};

struct Grad2f { float gx, gy; };   // Eigen::Matrix<float,1,2>

// ---- df::Frame (core/mapping/frame.h:36-120) ---------------------------------------------------------------------------------------
struct Frame {
  typedef std::shared_ptr<Frame> Ptr;
  Frame(std::size_t pyrlevels, std::size_t w, std::size_t h, std::shared_ptr<Context> ctx = Context::Default())
      : ctx(std::move(ctx)), width(w), height(h), pyr_img(pyrlevels, w, h, 1, this->ctx), pyr_grad(pyrlevels, w, h, 1, this->ctx) {}
  virtual ~Frame() {}
  virtual bool IsKeyframe() const { return false; }

  // frame.h:80-94: level 0 = img (HOST, w x h floats), level i = GaussianBlurDown(level i-1), Sobel / 8 gradient on every level
  // One call (dfx_build_pyramid: a launch per level, each level read once) instead of the reference's 2 L - 1 blocking single-image calls; same bits as
  // df::GaussianBlurDown / df::SobelGradients level by level.  BLOCKING like the reference, whose image operators each end in cudaDeviceSynchronize
  // (launch_utils.h:26-32): the pyramids are complete -- readable from any stream, any launch error raised HERE -- when it returns.
  void FillPyramids(const float* img_host, std::size_t pyrlevels) {
    pyr_img[0].Upload(img_host);
    const dfx_pyramid p = Describe(pyrlevels);
    check(dfx_build_pyramid(ctx->get(), &p));
  }
  // The enqueue-only form for a caller that orders its own work on the context's stream (a frame ring fed at camera rate): returns at once, the
  // pyramids are complete when the context's stream reaches this point; errors of the launches surface at the next blocking call.
  void FillPyramidsAsync(const float* img_host, std::size_t pyrlevels) {
    pyr_img[0].Upload(img_host);
    const dfx_pyramid p = Describe(pyrlevels);
    check(dfx_build_pyramid_batch_async(ctx->get(), &p, 1));
  }
  dfx_pyramid Describe(std::size_t pyrlevels, bool level0_gradient = true) const {
    if (pyrlevels < 1 || pyrlevels > pyr_img.Levels() || pyrlevels > DFX_MAX_PYR_LEVELS) throw Error(DFX_E_INVALID, "Frame: pyramid depth");
    dfx_pyramid p;
    std::memset(&p, 0, sizeof(p));
    p.levels = (int32_t)pyrlevels;
    for (std::size_t i = 0; i < pyrlevels; ++i) {
      p.img[i] = pyr_img[i].c_img();
      if (i > 0 || level0_gradient) p.grad[i] = pyr_grad[i].c_img();
    }
    return p;
  }

  std::shared_ptr<Context> ctx;
  std::size_t width, height;
  BufferPyramid<float> pyr_img;
  BufferPyramid<Grad2f> pyr_grad;
  dfx_se3 pose_wk{ { 0, 0, 0, 1 }, { 0, 0, 0 } };
  std::size_t id = 0;
  double timestamp = 0;
  bool marginalized = false;
};

// FillPyramids of many frames whose level 0 is already on the device (e.g. uploaded by the camera driver's stream): one launch per level over all frames.
// ENQUEUE ONLY (like Frame::FillPyramidsAsync): complete when the context's stream reaches this point
inline void FillPyramidsBatch(const std::vector<Frame*>& frames, std::size_t pyrlevels) {
  if (frames.empty()) return;
  std::vector<dfx_pyramid> d;
  for (const Frame* f : frames) d.push_back(f->Describe(pyrlevels));
  check(dfx_build_pyramid_batch_async(frames[0]->ctx->get(), d.data(), (int)d.size()));
}

// ---- df::Keyframe (core/mapping/keyframe.h:34-100) ---------------------------------------------------------------------------------
template <int CS>
struct Keyframe : Frame {
  typedef std::shared_ptr<Keyframe<CS>> Ptr;
  Keyframe(std::size_t pyrlevels, std::size_t w, std::size_t h, std::shared_ptr<Context> c = Context::Default())
      : Frame(pyrlevels, w, h, std::move(c)), pyr_dpt(pyrlevels, w, h, 1, ctx), pyr_vld(pyrlevels, w, h, 1, ctx), pyr_stdev(pyrlevels, w, h, 1, ctx),
        pyr_prx_orig(pyrlevels, w, h, 1, ctx), pyr_jac(pyrlevels, w, h, CS, ctx), dpt_grad(w, h, ctx) {
    for (std::size_t i = 0; i < pyrlevels; ++i) pyr_vld[i].Fill(1.0f);   // mapper.cpp:937 fillBuffer(pyr_vld, 1.0f)
    code.fill(0.0f);
  }
  bool IsKeyframe() const override { return true; }

  // the decoder's outputs at level i (decoder_network.cpp:126-136; HOST): zero-code proximity, log-uncertainty, code Jacobian [H][W*CS]
  void SetDecoderOutputs(std::size_t i, const float* prx_orig, const float* stdev, const float* jac) {
    pyr_prx_orig[i].Upload(prx_orig);
    pyr_stdev[i].Upload(stdev);
    pyr_jac[i].Upload(jac);
  }
  // mapper.cpp:984-1000 / 881-887: depth of every level from the code (one launch over all levels' jobs per image size is not
  // possible -- sizes differ -- so one enqueue per level, no host sync in between), then the level-0 depth gradient
  void UpdateDepthMaps(float avg_dpt = 2.0f, bool use_geometric = true) {
    for (std::size_t i = 0; i < pyr_dpt.Levels(); ++i)
      check(dfx_update_depth_batch_async(ctx->get(), CS, 1, code.data(), &pyr_prx_orig[i].c_img(), &pyr_jac[i].c_img(), avg_dpt, &pyr_dpt[i].c_img()));
    if (use_geometric) df::SobelGradients(pyr_dpt[0], dpt_grad, ctx);
  }

  BufferPyramid<float> pyr_dpt, pyr_vld, pyr_stdev, pyr_prx_orig, pyr_jac;
  DeviceImage<Grad2f> dpt_grad;
  std::array<float, CS> code;
};

// ---- replication of a keyframe over the ranks (the multi-GPU C ABI of include/dfx.h) ----------------------------------------------
// Every rank constructs the keyframe with the same sizes; rank `root` holds the content (FillPyramids, SetDecoderOutputs, code, pose).  One
// ncclBroadcast per buffer on the context's stream, the small host-side fields (code, pose, id, timestamp) through one device scratch image.
template <int CS>
void KeyframeBroadcast(Keyframe<CS>& kf, dfx_comm* comm, int root) {
  dfx_ctx* c = kf.ctx->get();
  auto bc = [&](const dfx_img& im) { check(dfx_comm_broadcast_async(c, comm, im.ptr, im.pitch_bytes * (std::size_t)im.h, root)); };
  for (std::size_t i = 0; i < kf.pyr_img.Levels(); ++i) {
    bc(kf.pyr_img[i].c_img()); bc(kf.pyr_grad[i].c_img()); bc(kf.pyr_dpt[i].c_img()); bc(kf.pyr_vld[i].c_img()); bc(kf.pyr_stdev[i].c_img());
    bc(kf.pyr_prx_orig[i].c_img()); bc(kf.pyr_jac[i].c_img());
  }
  bc(kf.dpt_grad.c_img());
  // code (CS), pose (7), id, timestamp, marginalized as one row of floats / raw words
  constexpr std::size_t kWords = CS + 7 + 4 + 1;   // ... + Frame::marginalized
  DeviceImage<float> small(kWords, 1, kf.ctx);
  std::vector<float> h(kWords, 0.f);
  for (int k = 0; k < CS; ++k) h[(std::size_t)k] = kf.code[(std::size_t)k];
  for (int k = 0; k < 4; ++k) h[CS + (std::size_t)k] = kf.pose_wk.q[k];
  for (int k = 0; k < 3; ++k) h[CS + 4 + (std::size_t)k] = kf.pose_wk.t[k];
  const std::uint64_t id64 = kf.id;
  std::memcpy(&h[CS + 7], &id64, 8);
  std::memcpy(&h[CS + 9], &kf.timestamp, 8);
  h[CS + 11] = kf.marginalized ? 1.0f : 0.0f;
  small.Upload(h);
  bc(small.c_img());
  h = small.Download();   // blocking: also the point where the broadcasts above are known to be complete on this rank
  for (int k = 0; k < CS; ++k) kf.code[(std::size_t)k] = h[(std::size_t)k];
  for (int k = 0; k < 4; ++k) kf.pose_wk.q[k] = h[CS + (std::size_t)k];
  for (int k = 0; k < 3; ++k) kf.pose_wk.t[k] = h[CS + 4 + (std::size_t)k];
  std::uint64_t idr;
  std::memcpy(&idr, &h[CS + 7], 8);
  kf.id = (std::size_t)idr;
  std::memcpy(&kf.timestamp, &h[CS + 9], 8);
  kf.marginalized = h[CS + 11] != 0.0f;
}   // (the valid0 shadows of the rewritten maps are reset by dfx_comm_broadcast_async itself: it is a writer the library sees)

// ---- gtsam::traits<Sophus::SE3f>::Local (core/gtsam/gtsam_traits.h:66-72): (t2 - t1, log(R2 R1^T)) -------------------------------------
namespace detail {
inline void quat_to_R(const float* q, double* R) {
  double x = q[0], y = q[1], z = q[2], w = q[3];
  const double n = std::sqrt(x * x + y * y + z * z + w * w);
  x /= n; y /= n; z /= n; w /= n;
  R[0] = 1 - 2 * (y * y + z * z); R[1] = 2 * (x * y - z * w);     R[2] = 2 * (x * z + y * w);
  R[3] = 2 * (x * y + z * w);     R[4] = 1 - 2 * (x * x + z * z); R[5] = 2 * (y * z - x * w);
  R[6] = 2 * (x * z - y * w);     R[7] = 2 * (y * z + x * w);     R[8] = 1 - 2 * (x * x + y * y);
}
inline double pose_local_norm(const dfx_se3& a, const dfx_se3& b) {
  double Ra[9], Rb[9], dR[9];
  quat_to_R(a.q, Ra); quat_to_R(b.q, Rb);
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) { double s = 0; for (int k = 0; k < 3; ++k) s += Rb[i * 3 + k] * Ra[j * 3 + k]; dR[i * 3 + j] = s; }
  const double c = std::fmin(1.0, std::fmax(-1.0, (dR[0] + dR[4] + dR[8] - 1.0) / 2.0)), ang = std::acos(c);
  double w[3] = { (dR[7] - dR[5]) / 2, (dR[2] - dR[6]) / 2, (dR[3] - dR[1]) / 2 };
  if (ang > 1e-12) for (double& v : w) v *= ang / std::sin(ang);
  double s = 0;
  for (int i = 0; i < 3; ++i) { const double d = (double)b.t[i] - a.t[i]; s += d * d + w[i] * w[i]; }
  return std::sqrt(s);
}
}  // namespace detail

// arguments of gtsam::HessianFactor(keys, Gs, gs, f) (photometric_factor.cpp:105-180), row-major double blocks
template <int CS>
struct HessianBlocks {
  std::array<double, 36> G11, G12, G22;
  std::array<double, 6 * CS> G13, G23;
  std::array<double, CS * CS> G33;
  std::array<double, 6> g1, g2;
  std::array<double, CS> g3;
  double f = 0;
};

// ---- df::PhotometricFactor<float,CS> without GTSAM: values are passed in directly ---------------------------------------------------------
template <int CS>
class PhotometricFactor {
 public:
  typedef df::SfmAligner<float, CS> AlignerT;
  typedef typename AlignerT::ReductionItem ReductionItem;
  PhotometricFactor(const dfx_cam& cam, std::shared_ptr<Keyframe<CS>> kf, std::shared_ptr<Frame> fr, int pyrlevel)
      : cam_(cam), kf_(std::move(kf)), fr_(std::move(fr)), pyrlevel_(pyrlevel) {}

  // GetJacobiansIfNeeded's test (:298-306): relinearise when a value moved by >= 1e-6 in its tangent space
  bool NeedsLinearization(const dfx_se3& pose0, const dfx_se3& pose1, const std::array<float, CS>& code0) const {
    if (first_) return true;
    const double eps = 1e-6;
    double dc = 0;
    for (int i = 0; i < CS; ++i) { const double d = (double)code0[i] - lin_code0_[i]; dc += d * d; }
    return !(detail::pose_local_norm(lin_pose0_, pose0) < eps) || !(detail::pose_local_norm(lin_pose1_, pose1) < eps) || !(std::sqrt(dc) < eps);
  }
  // RunAlignmentStep's post-processing (:275-282) + cache update; `item` is the raw result of this factor's RunStep at these values
  void Seed(const dfx_se3& pose0, const dfx_se3& pose1, const std::array<float, CS>& code0, ReductionItem item) {
    if (item.inliers > 0) item.residual = item.residual / item.inliers * cam_.w * cam_.h;
    else item.residual = std::numeric_limits<float>::infinity();
    lin_system_ = item; lin_pose0_ = pose0; lin_pose1_ = pose1; lin_code0_ = code0; first_ = false;
    ++linearizations_;
  }
  // linearize (:86-181) for ONE factor: UpdateDepthMaps + RunStep unless the cache holds these values
  const ReductionItem& GetJacobiansIfNeeded(AlignerT& aligner, const dfx_se3& pose0, const dfx_se3& pose1, const std::array<float, CS>& code0) {
    if (NeedsLinearization(pose0, pose1, code0)) {
      dfx_sfm_pair p = MakePair(pose0, pose1);
      const dfx_sfm_params prm = aligner.Params();
      std::vector<unsigned char> raw(dfx_item_size(12 + CS));
      check(dfx_sfm_linearize_batch(aligner.ContextHandle(), CS, &prm, &p, &kf_->pyr_prx_orig[pyrlevel_].c_img(), code0.data(), 1, raw.data()));
      Seed(pose0, pose1, code0, ReductionItem::FromRaw(raw.data()));
    }
    return lin_system_;
  }
  HessianBlocks<CS> Hessian() const {   // the slicing of linearize (:105-161): G = JtJ, g = -Jtr, f = rescaled residual
    HessianBlocks<CS> H;
    auto at = [&](int r, int c) { return (double)lin_system_.JtJ(r, c); };
    for (int r = 0; r < 6; ++r) for (int c = 0; c < 6; ++c) { H.G11[r * 6 + c] = at(r, c); H.G12[r * 6 + c] = at(r, 6 + c); H.G22[r * 6 + c] = at(6 + r, 6 + c); }
    for (int r = 0; r < 6; ++r) for (int c = 0; c < CS; ++c) { H.G13[r * CS + c] = at(r, 12 + c); H.G23[r * CS + c] = at(6 + r, 12 + c); }
    for (int r = 0; r < CS; ++r) for (int c = 0; c < CS; ++c) H.G33[r * CS + c] = at(12 + r, 12 + c);
    for (int i = 0; i < 6; ++i) { H.g1[i] = -(double)lin_system_.Jtr[i]; H.g2[i] = -(double)lin_system_.Jtr[6 + i]; }
    for (int i = 0; i < CS; ++i) H.g3[i] = -(double)lin_system_.Jtr[12 + i];
    H.f = lin_system_.residual;
    return H;
  }
  // error (:60-81): UpdateDepthMaps, EvaluateError, 0.5 * rescaled residual.  The decoder scale is the aligner's avg_dpt -- the one
  // dfx_sfm_linearize_batch decodes with, so error() and linearize() see the same depth map for the same code (the reference hard-codes
  // 2.0 in both paths, :332-341)
  double error(AlignerT& aligner, const dfx_se3& pose0, const dfx_se3& pose1, const std::array<float, CS>& code0) {
    const dfx_sfm_params prm = aligner.Params();
    check(dfx_update_depth_batch_async(aligner.ContextHandle(), CS, 1, code0.data(), &kf_->pyr_prx_orig[pyrlevel_].c_img(), &kf_->pyr_jac[pyrlevel_].c_img(),
                                       prm.avg_dpt, &kf_->pyr_dpt[pyrlevel_].c_img()));
    dfx_corr_item out;
    check(dfx_sfm_error(aligner.ContextHandle(), &pose0, &pose1, &cam_, &prm, &kf_->pyr_img[pyrlevel_].c_img(), &fr_->pyr_img[pyrlevel_].c_img(),
                        &kf_->pyr_dpt[pyrlevel_].c_img(), nullptr, nullptr, &out));
    return out.inliers > 0 ? 0.5 * (double)(out.residual / out.inliers * cam_.w * cam_.h) : std::numeric_limits<double>::infinity();
  }

  dfx_sfm_pair MakePair(const dfx_se3& pose0, const dfx_se3& pose1) const {
    dfx_sfm_pair p;
    p.pose0 = pose0; p.pose1 = pose1; p.cam = cam_;
    p.img0 = kf_->pyr_img[pyrlevel_].c_img(); p.img1 = fr_->pyr_img[pyrlevel_].c_img(); p.dpt0 = kf_->pyr_dpt[pyrlevel_].c_img();
    p.valid0 = kf_->pyr_vld[pyrlevel_].c_img(); p.prx0_jac = kf_->pyr_jac[pyrlevel_].c_img(); p.grad1 = fr_->pyr_grad[pyrlevel_].c_img();
    return p;
  }
  const std::shared_ptr<Keyframe<CS>>& keyframe() const { return kf_; }
  int pyrlevel() const { return pyrlevel_; }
  int linearizations() const { return linearizations_; }   // number of RunAlignmentSteps spent on this factor (observable effect of the cache)
  const ReductionItem& system() const { return lin_system_; }

 private:
  dfx_cam cam_;
  std::shared_ptr<Keyframe<CS>> kf_;
  std::shared_ptr<Frame> fr_;
  int pyrlevel_;
  bool first_ = true;
  dfx_se3 lin_pose0_{}, lin_pose1_{};
  std::array<float, CS> lin_code0_{};
  ReductionItem lin_system_{};
  int linearizations_ = 0;
};

// One relinearisation round (the batching seam): values[k] = (pose0, pose1, code0) of factors[k].  Factors whose values did not move
// keep their cache; the others run as ONE dfx_sfm_linearize_batch per pyramid level.  Returns the number of factors relinearised.
template <int CS>
struct FactorValues { dfx_se3 pose0, pose1; std::array<float, CS> code0; };

template <int CS>
int LinearizeAll(df::SfmAligner<float, CS>& aligner, const std::vector<PhotometricFactor<CS>*>& factors, const std::vector<FactorValues<CS>>& values) {
  if (factors.size() != values.size()) throw Error(DFX_E_INVALID, "LinearizeAll: one value triple per factor");
  std::map<int, std::vector<std::size_t>> todo;   // pyramid level -> factor indices (a batch needs one image size)
  for (std::size_t k = 0; k < factors.size(); ++k)
    if (factors[k]->NeedsLinearization(values[k].pose0, values[k].pose1, values[k].code0)) todo[factors[k]->pyrlevel()].push_back(k);
  int done = 0;
  const dfx_sfm_params prm = aligner.Params();
  const std::size_t isz = dfx_item_size(12 + CS);
  for (auto& lv : todo) {
    const std::vector<std::size_t>& idx = lv.second;
    std::vector<dfx_sfm_pair> pairs;
    std::vector<dfx_img> prx;
    std::vector<float> codes;
    for (std::size_t k : idx) {
      pairs.push_back(factors[k]->MakePair(values[k].pose0, values[k].pose1));
      prx.push_back(factors[k]->keyframe()->pyr_prx_orig[lv.first].c_img());
      codes.insert(codes.end(), values[k].code0.begin(), values[k].code0.end());
    }
    std::vector<unsigned char> raw(isz * idx.size());
    check(dfx_sfm_linearize_batch(aligner.ContextHandle(), CS, &prm, pairs.data(), prx.data(), codes.data(), (int)idx.size(), raw.data()));
    for (std::size_t q = 0; q < idx.size(); ++q) {
      const std::size_t k = idx[q];
      factors[k]->Seed(values[k].pose0, values[k].pose1, values[k].code0, PhotometricFactor<CS>::ReductionItem::FromRaw(raw.data() + q * isz));
      ++done;
    }
  }
  return done;
}

// PhotometricFactor::error over a factor set (what a Gauss-Newton / iSAM2 step evaluates to accept or reject an update; the reference
// calls error() factor by factor, each a blocking UpdateDepth + EvaluateError: photometric_factor.cpp:61-81,197-216).  Per pyramid level:
// ONE batched decode of the distinct keyframes and ONE batched EvaluateError (dfx_sfm_error_batch).  Returns sum_k error_k and, if asked,
// the per-factor values (0.5 * residual / inliers * w * h, +inf without overlap).
template <int CS>
double ErrorAll(df::SfmAligner<float, CS>& aligner, const std::vector<PhotometricFactor<CS>*>& factors, const std::vector<FactorValues<CS>>& values,
                std::vector<double>* per_factor = nullptr) {
  if (factors.size() != values.size()) throw Error(DFX_E_INVALID, "ErrorAll: one value triple per factor");
  std::map<int, std::vector<std::size_t>> by_level;
  for (std::size_t k = 0; k < factors.size(); ++k) by_level[factors[k]->pyrlevel()].push_back(k);
  const dfx_sfm_params prm = aligner.Params();
  std::vector<double> err(factors.size(), 0.0);
  for (auto& lv : by_level) {
    const std::vector<std::size_t>& idx = lv.second;
    std::vector<dfx_sfm_pair> pairs;
    std::vector<dfx_img> prx, jac, dpt;
    std::vector<float> codes;
    std::map<const void*, std::size_t> seen;   // depth map -> first factor that decodes it
    for (std::size_t k : idx) {
      pairs.push_back(factors[k]->MakePair(values[k].pose0, values[k].pose1));
      const auto& kf = factors[k]->keyframe();
      const void* key = kf->pyr_dpt[lv.first].c_img().ptr;
      auto hit = seen.find(key);
      if (hit == seen.end()) {
        seen.emplace(key, k);
        prx.push_back(kf->pyr_prx_orig[lv.first].c_img()); jac.push_back(kf->pyr_jac[lv.first].c_img()); dpt.push_back(kf->pyr_dpt[lv.first].c_img());
        codes.insert(codes.end(), values[k].code0.begin(), values[k].code0.end());
      } else if (values[hit->second].code0 != values[k].code0) {
        throw Error(DFX_E_INVALID, "ErrorAll: two factors of one keyframe carry different codes");
      }
    }
    check(dfx_update_depth_batch_async(aligner.ContextHandle(), CS, (int)prx.size(), codes.data(), prx.data(), jac.data(), prm.avg_dpt, dpt.data()));
    std::vector<dfx_corr_item> items(idx.size());
    check(dfx_sfm_error_batch(aligner.ContextHandle(), &prm, pairs.data(), (int)idx.size(), items.data()));
    for (std::size_t q = 0; q < idx.size(); ++q) {
      const dfx_sfm_pair& p = pairs[q];
      err[idx[q]] = items[q].inliers > 0 ? 0.5 * (double)(items[q].residual / items[q].inliers * p.cam.w * p.cam.h) : std::numeric_limits<double>::infinity();
    }
  }
  double total = 0.0;
  for (double e : err) total += e;
  if (per_factor) *per_factor = err;
  return total;
}

// ---- df::SparseGeometricFactor<float,CS> without GTSAM (core/gtsam/sparse_geometric_factor.{h,cpp}) -----------------------------------------------------
// The sampled points are fixed at construction (sparse_geometric_factor.cpp:50-53) and live in device memory; Linearize() is the reference's
// per-factor pattern (one blocking call), SparseGeometricLinearizeAll the round: every factor of the graph in ONE launch, rows left on the
// device (or fetched with one copy).  Rows: [n_points][12 + 2 CS + 1] = [A_pose0 | A_pose1 | A_code0 | A_code1 | b] of the JacobianFactor
// (sparse_geometric_factor.cpp:262-275).
template <int CS>
class SparseGeometricFactor {
 public:
  static constexpr int kCols = 12 + 2 * CS + 1;
  SparseGeometricFactor(const dfx_cam& cam, const std::vector<std::array<int32_t, 2>>& points, std::shared_ptr<Keyframe<CS>> kf0, std::shared_ptr<Keyframe<CS>> kf1,
                        float huber_delta, float avg_dpt = 2.0f)
      : cam_(cam), kf0_(std::move(kf0)), kf1_(std::move(kf1)), huber_delta_(huber_delta), avg_dpt_(avg_dpt), n_points_((int)points.size()),
        points_dev_((points.size() * 2), 1, kf0_->ctx) {
    static_assert(sizeof(std::array<int32_t, 2>) == 8, "packed (x, y) pairs");
    for (const auto& p : points)
      if (p[0] < 0 || p[0] >= (int)kf0_->width || p[1] < 0 || p[1] >= (int)kf0_->height) throw Error(DFX_E_INVALID, "SparseGeometricFactor: point outside the image");
    points_dev_.Upload(reinterpret_cast<const float*>(points.data()));   // raw words
  }
  int n_points() const { return n_points_; }
  float huber_delta() const { return huber_delta_; }
  float avg_dpt() const { return avg_dpt_; }
  dfx_ctx* ctx() const { return kf0_->ctx->get(); }
  // the descriptor of this factor at (pose0, pose1, code0, code1); the code arrays must outlive the call that consumes it
  dfx_sparse_geo_factor Describe(const dfx_se3& pose0, const dfx_se3& pose1, const std::array<float, CS>& code0, const std::array<float, CS>& code1) const {
    dfx_sparse_geo_factor f;
    std::memset(&f, 0, sizeof(f));
    f.pose0 = pose0; f.pose1 = pose1; f.cam = cam_; f.code0 = code0.data(); f.code1 = code1.data();
    f.points_xy = reinterpret_cast<const int32_t*>(points_dev_.ptr()); f.n_points = n_points_; f.points_on_device = 1;
    f.prx0_orig = kf0_->pyr_prx_orig[0].c_img(); f.prx0_jac = kf0_->pyr_jac[0].c_img();
    f.prx1_orig = kf1_->pyr_prx_orig[0].c_img(); f.prx1_jac = kf1_->pyr_jac[0].c_img(); f.dpt1_grad = kf1_->dpt_grad.c_img();
    return f;
  }
  std::vector<float> Linearize(const dfx_se3& pose0, const dfx_se3& pose1, const std::array<float, CS>& code0, const std::array<float, CS>& code1) const {
    std::vector<float> rows((std::size_t)n_points_ * kCols);
    const dfx_sparse_geo_factor f = Describe(pose0, pose1, code0, code1);
    check(dfx_sparse_geometric_linearize_batch(ctx(), CS, &f, 1, huber_delta_, avg_dpt_, rows.data()));
    return rows;
  }

 private:
  dfx_cam cam_;
  std::shared_ptr<Keyframe<CS>> kf0_, kf1_;
  float huber_delta_, avg_dpt_;
  int n_points_;
  DeviceImage<float> points_dev_;
};

template <int CS>
struct GeoValues { dfx_se3 pose0, pose1; std::array<float, CS> code0, code1; };

// All factors in ONE launch.  rows_dev != nullptr: enqueue only, factor k's rows start at row sum_{j<k} n_points_j of rows_dev (device memory of
// sum n_points x kCols floats).  Otherwise the rows come back in one host vector (one device-to-host copy).
template <int CS>
std::vector<float> SparseGeometricLinearizeAll(const std::vector<SparseGeometricFactor<CS>*>& factors, const std::vector<GeoValues<CS>>& values, float* rows_dev = nullptr) {
  if (factors.empty() || factors.size() != values.size()) throw Error(DFX_E_INVALID, "SparseGeometricLinearizeAll: one value tuple per factor");
  std::vector<dfx_sparse_geo_factor> d;
  std::size_t total = 0;
  for (std::size_t k = 0; k < factors.size(); ++k) {
    if (factors[k]->huber_delta() != factors[0]->huber_delta() || factors[k]->avg_dpt() != factors[0]->avg_dpt() || factors[k]->ctx() != factors[0]->ctx())
      throw Error(DFX_E_INVALID, "SparseGeometricLinearizeAll: the factors of a round share huber_delta, avg_dpt and the context");
    d.push_back(factors[k]->Describe(values[k].pose0, values[k].pose1, values[k].code0, values[k].code1));
    total += (std::size_t)factors[k]->n_points();
  }
  if (rows_dev) {
    check(dfx_sparse_geometric_linearize_batch_async(factors[0]->ctx(), CS, d.data(), (int)d.size(), factors[0]->huber_delta(), factors[0]->avg_dpt(), rows_dev));
    return {};
  }
  std::vector<float> rows(total * SparseGeometricFactor<CS>::kCols);
  check(dfx_sparse_geometric_linearize_batch(factors[0]->ctx(), CS, d.data(), (int)d.size(), factors[0]->huber_delta(), factors[0]->avg_dpt(), rows.data()));
  return rows;
}

// The round's NORMAL EQUATIONS instead of its rows (dfx_sparse_geometric_gram_batch[_async]): per factor the upper triangle (row-major) of [A | b]^T [A | b],
// kGram = kCols (kCols + 1) / 2 floats -- what gtsam's elimination forms from the JacobianFactor on the host (sparse_geometric_factor.cpp:262-275 hands it the
// rows), formed on the device: a 1024-factor round returns 12 MB instead of 157 MB.  gram_dev != nullptr: device memory [n][kGram], enqueue only, returns {}.
template <int CS>
std::vector<float> SparseGeometricGramAll(const std::vector<SparseGeometricFactor<CS>*>& factors, const std::vector<GeoValues<CS>>& values, float* gram_dev = nullptr) {
  constexpr std::size_t kGram = (std::size_t)SparseGeometricFactor<CS>::kCols * (SparseGeometricFactor<CS>::kCols + 1) / 2;
  if (factors.empty() || factors.size() != values.size()) throw Error(DFX_E_INVALID, "SparseGeometricGramAll: one value tuple per factor");
  std::vector<dfx_sparse_geo_factor> d;
  for (std::size_t k = 0; k < factors.size(); ++k) {
    if (factors[k]->huber_delta() != factors[0]->huber_delta() || factors[k]->avg_dpt() != factors[0]->avg_dpt() || factors[k]->ctx() != factors[0]->ctx())
      throw Error(DFX_E_INVALID, "SparseGeometricGramAll: the factors of a round share huber_delta, avg_dpt and the context");
    d.push_back(factors[k]->Describe(values[k].pose0, values[k].pose1, values[k].code0, values[k].code1));
  }
  if (gram_dev) {
    check(dfx_sparse_geometric_gram_batch_async(factors[0]->ctx(), CS, d.data(), (int)d.size(), factors[0]->huber_delta(), factors[0]->avg_dpt(), gram_dev));
    return {};
  }
  std::vector<float> g(factors.size() * kGram);
  check(dfx_sparse_geometric_gram_batch(factors[0]->ctx(), CS, d.data(), (int)d.size(), factors[0]->huber_delta(), factors[0]->avg_dpt(), g.data()));
  return g;
}

// Page-locked host memory (dfx_host_alloc) for large results that come back every round: the rows of a 1024-factor graph are 150 MB, and a fresh pageable
// std::vector per round costs more in page faults and a 5 GB/s copy than every kernel of the round.
template <typename T>
class PinnedBuffer {
 public:
  PinnedBuffer(std::size_t n, std::shared_ptr<Context> c) : ctx_(std::move(c)), n_(n) { void* p = nullptr; check(dfx_host_alloc(ctx_->get(), n * sizeof(T), &p)); p_ = static_cast<T*>(p); }
  ~PinnedBuffer() { (void)dfx_host_free(ctx_->get(), p_); }
  PinnedBuffer(const PinnedBuffer&) = delete;
  PinnedBuffer& operator=(const PinnedBuffer&) = delete;
  T* data() { return p_; }
  const T* data() const { return p_; }
  std::size_t size() const { return n_; }
  T& operator[](std::size_t i) { return p_[i]; }
  const T& operator[](std::size_t i) const { return p_[i]; }

 private:
  std::shared_ptr<Context> ctx_;
  T* p_ = nullptr;
  std::size_t n_;
};

// the round's rows into a buffer the caller keeps across rounds (factor k's rows start at (points of the factors before it) * kCols); returns the row count
template <int CS>
std::size_t SparseGeometricLinearizeAll(const std::vector<SparseGeometricFactor<CS>*>& factors, const std::vector<GeoValues<CS>>& values, PinnedBuffer<float>& rows_host) {
  if (factors.empty() || factors.size() != values.size()) throw Error(DFX_E_INVALID, "SparseGeometricLinearizeAll: one value tuple per factor");
  std::vector<dfx_sparse_geo_factor> d;
  std::size_t total = 0;
  for (std::size_t k = 0; k < factors.size(); ++k) {
    if (factors[k]->huber_delta() != factors[0]->huber_delta() || factors[k]->avg_dpt() != factors[0]->avg_dpt() || factors[k]->ctx() != factors[0]->ctx())
      throw Error(DFX_E_INVALID, "SparseGeometricLinearizeAll: the factors of a round share huber_delta, avg_dpt and the context");
    d.push_back(factors[k]->Describe(values[k].pose0, values[k].pose1, values[k].code0, values[k].code1));
    total += (std::size_t)factors[k]->n_points();
  }
  if (rows_host.size() < total * SparseGeometricFactor<CS>::kCols) throw Error(DFX_E_INVALID, "SparseGeometricLinearizeAll: row buffer too small");
  check(dfx_sparse_geometric_linearize_batch(factors[0]->ctx(), CS, d.data(), (int)d.size(), factors[0]->huber_delta(), factors[0]->avg_dpt(), rows_host.data()));
  return total;
}

}  // namespace dfx
