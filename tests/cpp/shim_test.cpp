// shim_test.cpp -- include/dfx_shim.hpp (the C++17 host mirror of the reference's operator interface) end to end on a GPU, written the
// way the reference's two callers write it, against Sophus- / Eigen- / VisionCore-shaped types (the stand-in headers of
// oracle/standins/, test infrastructure; the real libraries are absent from this image):
//   (1) CameraTracker::TrackFrame             core/system/camera_tracker.cpp:48-78   (SE3Aligner::RunStep, Eigen ldlt solve, Sophus retract, Warp)
//   (2) PhotometricFactor::RunAlignmentStep   core/gtsam/photometric_factor.cpp:225-293 (UpdateDepth + SfmAligner::RunStep, residual rescaling)
//       PhotometricFactor::linearize          :105-161  (toDenseMatrix().cast<double>(), -Jtr.cast<double>(), G11..G33 / g1..g3 blocks)
//       PhotometricFactor::RunWarping         :197-216  (EvaluateError)
//   (3) the rest of the interface: RunStepBatch, DepthAligner, GaussianBlurDown, SquaredError, per-aligner SetStepThreadsBlocks
// Every result is asserted against the same call made directly through the C ABI (bit for bit: the kernels are deterministic).
// Exit code 0 = pass.  Built by tests/cpp/Makefile (hipcc), run by tests/test_gpu_cpp_shim.py.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstring>
#include <limits>
#include <memory>
#include <vector>

#include <Eigen/Core>
#include <sophus/se3.hpp>
#include <VisionCore/Buffers/Image2D.hpp>

#include "../../include/dfx_shim.hpp"

#ifndef DFX_SHIM_HAS_EIGEN
#error "this test must see <Eigen/Core> (tests/cpp/Makefile adds -I oracle/standins)"
#endif

#define HIPOK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); std::exit(2); } } while (0)
#define REQUIRE(c) do { if (!(c)) { std::printf("FAILED %s at %s:%d\n", #c, __FILE__, __LINE__); return 1; } } while (0)

// df::PinholeCamera<float> as far as the callers use it (common/algorithm/pinhole_camera.h:44-128)
template <typename Scalar>
class PinholeCamera {
 public:
  PinholeCamera(Scalar fx, Scalar fy, Scalar u0, Scalar v0, Scalar w, Scalar h) : fx_(fx), fy_(fy), u0_(u0), v0_(v0), w_(w), h_(h) {}
  const Scalar& fx() const { return fx_; } const Scalar& fy() const { return fy_; }
  const Scalar& u0() const { return u0_; } const Scalar& v0() const { return v0_; }
  const Scalar& width() const { return w_; } const Scalar& height() const { return h_; }
 private:
  Scalar fx_, fy_, u0_, v0_, w_, h_;
};

typedef vc::Image2DView<float, vc::TargetDeviceCUDA> ImgView;
typedef Eigen::Matrix<float, 1, 2> GradT;
typedef vc::Image2DView<GradT, vc::TargetDeviceCUDA> GradView;

// vc::Image2DManaged<T, TargetDeviceCUDA>: an owning device image that IS a view (the reference passes it where views are expected)
template <typename T>
class DeviceImage : public vc::Image2DView<T, vc::TargetDeviceCUDA> {
 public:
  DeviceImage(std::size_t w, std::size_t h) : vc::Image2DView<T, vc::TargetDeviceCUDA>(alloc(w, h), w, h, w * sizeof(T)) {}
  ~DeviceImage() { (void)hipFree(this->ptr()); }
  DeviceImage(const DeviceImage&) = delete;
  void upload(const std::vector<T>& v) { HIPOK(hipMemcpy(this->ptr(), v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice)); }
  std::vector<T> download() const { std::vector<T> v(this->width() * this->height()); HIPOK(hipMemcpy(v.data(), this->ptr(), v.size() * sizeof(T), hipMemcpyDeviceToHost)); return v; }
 private:
  static T* alloc(std::size_t w, std::size_t h) { T* p = nullptr; HIPOK(hipMalloc(&p, w * h * sizeof(T))); HIPOK(hipMemset(p, 0, w * h * sizeof(T))); return p; }
};

// SyncedBufferPyramid as the callers see it (cuda/synced_pyramid.h): GetGpuLevel(i) -> view
template <typename T>
struct Pyramid {
  std::vector<std::unique_ptr<DeviceImage<T>>> lv;
  vc::Image2DView<T, vc::TargetDeviceCUDA> GetGpuLevel(int i) const { return *lv[i]; }
};
struct Frame { Pyramid<float> pyr_img; Pyramid<GradT> pyr_grad; };
struct Keyframe : Frame { Pyramid<float> pyr_dpt, pyr_vld, pyr_stdev, pyr_prx_orig, pyr_jac; };

static double tex(double u, double v) {
  return 0.5 + 0.2 * std::sin(0.081 * u + 0.047 * v) + 0.15 * std::sin(0.033 * u - 0.112 * v + 1.0) + 0.1 * std::sin(0.15 * u + 0.09 * v + 2.0);
}

static dfx_img cimg(const void* p, std::size_t pitch, std::size_t w, std::size_t h) { return dfx_img{ const_cast<void*>(p), pitch, (uint32_t)w, (uint32_t)h }; }
template <typename V> static dfx_img cimg(const V& v) { return cimg(v.ptr(), v.pitch(), v.width(), v.height()); }
static dfx_se3 cse3(const Sophus::SE3f& p) {
  const auto q = p.unit_quaternion();
  return dfx_se3{ { q.x(), q.y(), q.z(), q.w() }, { p.translation()[0], p.translation()[1], p.translation()[2] } };
}

int main() {
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) { std::printf("no HIP device\n"); return 3; }
  constexpr int CS = 32;
  const int W = 320, H = 240;
  const PinholeCamera<float> cam_(277.128f, 289.706f, 160.f, 120.f, (float)W, (float)H);
  const Eigen::Matrix<float, 3, 1> w_gt(0.01f, -0.015f, 0.008f), t_gt(0.04f, -0.03f, 0.02f);
  const Sophus::SE3f pose10_gt(Sophus::SO3f::exp(w_gt), t_gt);
  const Eigen::Matrix<float, 3, 3> Rg = pose10_gt.so3().matrix();

  // ---- synthetic keyframe + frame: linear decoder prx = prx_orig + jac . code, depth = a / prx - a
  const float avg_dpt = 2.0f;
  std::vector<float> img0(W * H), img1(W * H), prx_orig(W * H), jac((size_t)W * H * CS), dpt_true(W * H);
  Eigen::Matrix<float, CS, 1> code_true;
  for (int k = 0; k < CS; ++k) code_true(k) = 0.3f * std::sin(1.7f * k + 0.3f);
  for (int y = 0; y < H; ++y) for (int x = 0; x < W; ++x) {
    const double d = 2.5 + 0.3 * (x - 160.0) / W - 0.3 * (y - 120.0) / H + 0.2 * std::sin(0.02 * x) * std::cos(0.03 * y);
    double jc = 0;
    for (int k = 0; k < CS; ++k) {
      const float j = 0.004f * (float)std::sin(0.013 * (k + 1) * x / 8.0 + 0.7 * k) * (float)std::cos(0.011 * (k + 2) * y / 8.0 - 0.3 * k);
      jac[((size_t)y * W + x) * CS + k] = j;
      jc += (double)j * code_true(k);
    }
    prx_orig[y * W + x] = (float)(avg_dpt / (avg_dpt + d) - jc);
    const double dd = avg_dpt / ((double)prx_orig[y * W + x] + jc) - avg_dpt;
    dpt_true[y * W + x] = (float)dd;
    const double X = (x - cam_.u0()) / cam_.fx() * dd, Y = (y - cam_.v0()) / cam_.fy() * dd, Z = dd;
    const double qx = Rg(0, 0) * X + Rg(0, 1) * Y + Rg(0, 2) * Z + t_gt[0], qy = Rg(1, 0) * X + Rg(1, 1) * Y + Rg(1, 2) * Z + t_gt[1],
                 qz = Rg(2, 0) * X + Rg(2, 1) * Y + Rg(2, 2) * Z + t_gt[2];
    img1[y * W + x] = (float)tex(x, y);
    img0[y * W + x] = (float)tex(cam_.fx() * qx / qz + cam_.u0(), cam_.fy() * qy / qz + cam_.v0());
  }
  auto kf_ = std::make_shared<Keyframe>();
  auto fr_ = std::make_shared<Frame>();
  auto add = [&](Pyramid<float>& p, std::size_t w, const std::vector<float>* v) { p.lv.emplace_back(new DeviceImage<float>(w, H)); if (v) p.lv.back()->upload(*v); };
  add(kf_->pyr_img, W, &img0); add(fr_->pyr_img, W, &img1); add(kf_->pyr_dpt, W, nullptr); add(kf_->pyr_vld, W, nullptr); add(kf_->pyr_stdev, W, nullptr);
  add(kf_->pyr_prx_orig, W, &prx_orig); add(kf_->pyr_jac, (std::size_t)W * CS, &jac);
  fr_->pyr_grad.lv.emplace_back(new DeviceImage<GradT>(W, H));

  try {
    dfx_ctx* cabi = dfx::Context::Default()->get();
    const int i = 0;   // pyrlevel_
    {   // Frame::FillPyramids (mapping/frame.h:84-90)
      GradView g = fr_->pyr_grad.GetGpuLevel(i);
      df::SobelGradients(fr_->pyr_img.GetGpuLevel(i), g);
    }

    // ================= (2) PhotometricFactor::RunAlignmentStep, photometric_factor.cpp:225-293 =================
    typedef df::SfmAligner<float, CS> AlignerT;
    auto aligner_ = std::make_shared<AlignerT>(df::SfmAlignerParams());
    const Sophus::SE3f pose0, pose1 = pose10_gt.inverse();   // pose_10 = pose1^-1 * pose0
    const Eigen::Matrix<double, CS, 1> code0 = code_true.cast<double>();   // gtsam::Vector is double
    {   // UpdateDepthMaps (:332-341)
      const Eigen::Matrix<float, CS, 1> cde = code0.template cast<float>();
      ImgView dpt = kf_->pyr_dpt.GetGpuLevel(i);
      df::UpdateDepth<float, CS, ImgView>(cde, kf_->pyr_prx_orig.GetGpuLevel(i), kf_->pyr_jac.GetGpuLevel(i), avg_dpt, dpt);
      const std::vector<float> got = kf_->pyr_dpt.lv[0]->download();
      double e = 0;
      for (int k = 0; k < W * H; ++k) e = std::max(e, (double)std::fabs(got[k] - dpt_true[k]));
      std::printf("UpdateDepth: max |dpt - truth| = %.3e\n", e);
      REQUIRE(e < 2e-5);
    }
    Eigen::Matrix<float, CS, 1> cde = code0.template cast<float>();
    vc::Image2DView<float, vc::TargetDeviceCUDA> vld = kf_->pyr_vld.GetGpuLevel(i);
    auto result = aligner_->RunStep(pose0, pose1, cde, cam_,
                                    kf_->pyr_img.GetGpuLevel(i),
                                    fr_->pyr_img.GetGpuLevel(i),
                                    kf_->pyr_dpt.GetGpuLevel(i),
                                    kf_->pyr_stdev.GetGpuLevel(i),
                                    vld,
                                    kf_->pyr_jac.GetGpuLevel(i),
                                    fr_->pyr_grad.GetGpuLevel(i));
    const float raw_residual = result.residual;
    if (result.inliers > 0)
      result.residual = result.residual / result.inliers * cam_.width() * cam_.height();
    else
      result.residual = std::numeric_limits<float>::infinity();
    REQUIRE(result.inliers > (std::size_t)(0.9 * W * H));

    // the same call through the C ABI
    std::vector<unsigned char> raw(dfx_item_size(12 + CS));
    {
      const dfx_se3 p0 = cse3(pose0), p1 = cse3(pose1);
      const dfx_cam c{ cam_.fx(), cam_.fy(), cam_.u0(), cam_.v0(), cam_.width(), cam_.height() };
      const dfx_sfm_params prm{ 0.1f, 2.0f, 0.0f, 2, 0 };
      const dfx_img i0 = cimg(*kf_->pyr_img.lv[0]), i1 = cimg(*fr_->pyr_img.lv[0]), d0 = cimg(*kf_->pyr_dpt.lv[0]), s0 = cimg(*kf_->pyr_stdev.lv[0]),
                    v0 = cimg(*kf_->pyr_vld.lv[0]), jc = cimg(*kf_->pyr_jac.lv[0]), g1 = cimg(*fr_->pyr_grad.lv[0]);
      dfx::check(dfx_sfm_step(cabi, CS, &p0, &p1, &c, &prm, &i0, &i1, &d0, &s0, &v0, &jc, &g1, raw.data()));
    }
    REQUIRE(result.inliers == (std::size_t)dfx_item_inliers(raw.data(), 12 + CS));
    REQUIRE(raw_residual == dfx_item_residual(raw.data(), 12 + CS));
    REQUIRE(std::memcmp(result.JtJ.coeff().data(), dfx_item_jtj(raw.data()), sizeof(float) * dfx_item_jtj_len(12 + CS)) == 0);
    for (int k = 0; k < 12 + CS; ++k) REQUIRE(result.Jtr(k) == dfx_item_jtr(raw.data(), 12 + CS)[k]);

    // ---- PhotometricFactor::linearize, :105-106 and :135-161
    auto& sys = result;
    auto JtJ = sys.JtJ.toDenseMatrix().template cast<double>();
    auto Jtr = -sys.Jtr.template cast<double>();
    const Eigen::Matrix<double, 6, 6> G11 = JtJ.template block<6, 6>(0, 0);
    const Eigen::Matrix<double, 6, 6> G12 = JtJ.template block<6, 6>(0, 6);
    const Eigen::Matrix<double, 6, CS> G13 = JtJ.template block<6, CS>(0, 12);
    const Eigen::Matrix<double, 6, 6> G22 = JtJ.template block<6, 6>(6, 6);
    const Eigen::Matrix<double, 6, CS> G23 = JtJ.template block<6, CS>(6, 12);
    const Eigen::Matrix<double, CS, CS> G33 = JtJ.template block<CS, CS>(12, 12);
    const Eigen::Matrix<double, 6, 1> g1 = Jtr.template block<6, 1>(0, 0);
    const Eigen::Matrix<double, 6, 1> g2 = Jtr.template block<6, 1>(6, 0);
    const Eigen::Matrix<double, CS, 1> g3 = Jtr.template block<CS, 1>(12, 0);
    const float* pk = dfx_item_jtj(raw.data());   // packed row-major upper triangle
    auto packed = [&](int r, int c) { if (r > c) std::swap(r, c); return (double)pk[(std::size_t)r * (12 + CS) - (std::size_t)r * (r - 1) / 2 + (c - r)]; };
    REQUIRE(G11(2, 5) == packed(2, 5) && G11(5, 2) == packed(2, 5) && G12(1, 4) == packed(1, 10) && G13(3, 7) == packed(3, 19));
    REQUIRE(G22(0, 0) == packed(6, 6) && G23(5, 31) == packed(11, 43) && G33(31, 0) == packed(12, 43) && G33(4, 4) == packed(16, 16));
    REQUIRE(g1(0) == -(double)dfx_item_jtr(raw.data(), 12 + CS)[0] && g2(5) == -(double)dfx_item_jtr(raw.data(), 12 + CS)[11] && g3(31) == -(double)dfx_item_jtr(raw.data(), 12 + CS)[43]);
    REQUIRE(G33(3, 3) > 0 && G11(0, 0) > 0);

    // ---- PhotometricFactor::RunWarping, :197-216
    {
      auto res = aligner_->EvaluateError(pose0, pose1, cam_, kf_->pyr_img.GetGpuLevel(i), fr_->pyr_img.GetGpuLevel(i), kf_->pyr_dpt.GetGpuLevel(i),
                                         kf_->pyr_stdev.GetGpuLevel(i), fr_->pyr_grad.GetGpuLevel(i));
      dfx_corr_item ci;
      const dfx_se3 p0 = cse3(pose0), p1 = cse3(pose1);
      const dfx_cam c{ cam_.fx(), cam_.fy(), cam_.u0(), cam_.v0(), cam_.width(), cam_.height() };
      const dfx_sfm_params prm{ 0.1f, 2.0f, 0.0f, 2, 0 };
      const dfx_img i0 = cimg(*kf_->pyr_img.lv[0]), i1 = cimg(*fr_->pyr_img.lv[0]), d0 = cimg(*kf_->pyr_dpt.lv[0]);
      dfx::check(dfx_sfm_error(cabi, &p0, &p1, &c, &prm, &i0, &i1, &d0, nullptr, nullptr, &ci));
      REQUIRE(res.inliers == (std::size_t)ci.inliers && res.residual == ci.residual && res.inliers > (std::size_t)(0.9 * W * H));
    }

    // ---- batched extension == single calls; a second aligner with its own launch shape does not disturb the first
    {
      AlignerT other;
      other.SetStepThreadsBlocks(256, 7);
      auto alt = other.RunStep(pose0, pose1, cde, cam_, kf_->pyr_img.GetGpuLevel(i), fr_->pyr_img.GetGpuLevel(i), kf_->pyr_dpt.GetGpuLevel(i),
                               kf_->pyr_stdev.GetGpuLevel(i), vld, kf_->pyr_jac.GetGpuLevel(i), fr_->pyr_grad.GetGpuLevel(i));
      REQUIRE(alt.inliers == result.inliers && std::fabs(alt.JtJ(0, 0) - result.JtJ(0, 0)) <= 1e-4f * result.JtJ(0, 0));
      std::vector<dfx_sfm_pair> pairs(3, AlignerT::MakePair(pose0, pose1, cam_, kf_->pyr_img.GetGpuLevel(i), fr_->pyr_img.GetGpuLevel(i), kf_->pyr_dpt.GetGpuLevel(i),
                                                             vld, kf_->pyr_jac.GetGpuLevel(i), fr_->pyr_grad.GetGpuLevel(i)));
      auto items = aligner_->RunStepBatch(pairs);
      auto again = aligner_->RunStep(pose0, pose1, cde, cam_, kf_->pyr_img.GetGpuLevel(i), fr_->pyr_img.GetGpuLevel(i), kf_->pyr_dpt.GetGpuLevel(i),
                                     kf_->pyr_stdev.GetGpuLevel(i), vld, kf_->pyr_jac.GetGpuLevel(i), fr_->pyr_grad.GetGpuLevel(i));
      REQUIRE(std::memcmp(again.JtJ.coeff().data(), result.JtJ.coeff().data(), sizeof(float) * dfx_item_jtj_len(12 + CS)) == 0);   // unchanged by `other`
      REQUIRE(items.size() == 3 && items[0].inliers == result.inliers && items[2].inliers == result.inliers);
      REQUIRE(std::memcmp(items[0].JtJ.coeff().data(), items[2].JtJ.coeff().data(), sizeof(float) * dfx_item_jtj_len(12 + CS)) == 0);
      REQUIRE(std::fabs(items[1].JtJ(20, 20) - result.JtJ(20, 20)) <= 1e-5f * result.JtJ(20, 20));
    }

    // ---- DepthAligner<float,CS>::RunStep (cu_depthaligner.cpp:78-110) vs the C ABI
    {
      df::DepthAligner<float, CS> da;
      const Eigen::Matrix<float, CS, 1> c0 = Eigen::Matrix<float, CS, 1>::Zero();
      DeviceImage<float> target(W, H);
      target.upload(dpt_true);
      auto it = da.RunStep(c0, target, kf_->pyr_prx_orig.GetGpuLevel(i), kf_->pyr_jac.GetGpuLevel(i));
      std::vector<unsigned char> r2(dfx_item_size(CS));
      float cz[CS] = { 0 };
      const dfx_img tg = cimg(target), po = cimg(*kf_->pyr_prx_orig.lv[0]), jc = cimg(*kf_->pyr_jac.lv[0]);
      dfx::check(dfx_depth_aligner_step(cabi, CS, cz, &tg, &po, &jc, 2.0f, r2.data()));
      REQUIRE(it.inliers == (std::size_t)(W * H) && it.inliers == (std::size_t)dfx_item_inliers(r2.data(), CS));
      REQUIRE(std::memcmp(it.JtJ.coeff().data(), dfx_item_jtj(r2.data()), sizeof(float) * dfx_item_jtj_len(CS)) == 0 && it.residual == dfx_item_residual(r2.data(), CS));
      REQUIRE(it.residual > 0);
    }

    // ---- GaussianBlurDown / SquaredError (cu_image_proc.h:27-38)
    {
      DeviceImage<float> half(W / 2, H / 2), half2(W / 2, H / 2);
      df::GaussianBlurDown(kf_->pyr_img.GetGpuLevel(i), half);
      df::GaussianBlurDown(fr_->pyr_img.GetGpuLevel(i), half2);
      const float se = df::SquaredError(half, half2), zero = df::SquaredError(half, half);
      float ref = 0;
      const dfx_img a = cimg(half), b = cimg(half2);
      dfx::check(dfx_squared_error(cabi, &a, &b, &ref));
      REQUIRE(se == ref && se > 0 && zero == 0.0f);
    }

    // ================= (1) CameraTracker::TrackFrame, camera_tracker.cpp:48-78 =================
    {
      df::SE3Aligner<float> se3aligner_;
      se3aligner_.SetHuberDelta(0.1f);
      Sophus::SE3f pose_ck_;   // CameraTracker::Reset
      std::vector<PinholeCamera<float>> camera_pyr_(1, cam_);
      std::vector<ImgView> pyr_img1(1, fr_->pyr_img.GetGpuLevel(0));
      std::vector<GradView> pyr_grad1(1, fr_->pyr_grad.GetGpuLevel(0));
      struct { int pyramid_levels = 1; std::vector<int> iterations_per_level = { 15 }; } config_;
      float inliers_ = 0, error_ = 0;
      for (int level = config_.pyramid_levels - 1; level >= 0; --level) {
        for (int iter = 0; iter < config_.iterations_per_level[level]; ++iter) {
          auto result = se3aligner_.RunStep(pose_ck_, camera_pyr_[level],
                                            kf_->pyr_img.GetGpuLevel(level),
                                            pyr_img1[level],
                                            kf_->pyr_dpt.GetGpuLevel(level),
                                            pyr_grad1[level]);
          if (iter == 0) {   // the first step equals the C ABI's
            unsigned char r6[120];
            const dfx_se3 p = cse3(pose_ck_);
            const dfx_cam c{ cam_.fx(), cam_.fy(), cam_.u0(), cam_.v0(), cam_.width(), cam_.height() };
            const dfx_img i0 = cimg(*kf_->pyr_img.lv[0]), i1 = cimg(*fr_->pyr_img.lv[0]), d0 = cimg(*kf_->pyr_dpt.lv[0]), g1 = cimg(*fr_->pyr_grad.lv[0]);
            dfx::check(dfx_se3_step(cabi, &p, &c, &i0, &i1, &d0, &g1, 0.1f, r6));
            REQUIRE(std::memcmp(result.JtJ.coeff().data(), dfx_item_jtj(r6), 21 * sizeof(float)) == 0 && result.inliers == (std::size_t)dfx_item_inliers(r6, 6));
          }
          // update estimate
          Eigen::Matrix<float, 6, 1> update = -result.JtJ.toDenseMatrix().ldlt().solve(result.Jtr);
          Eigen::Matrix<float, 3, 1> trs_update = update.head<3>();
          Eigen::Matrix<float, 3, 1> rot_update = update.tail<3>();
          pose_ck_.translation() += trs_update;
          pose_ck_.so3() = Sophus::SO3f::exp(rot_update) * pose_ck_.so3();

          if (level == 0 && iter == config_.iterations_per_level[level] - 1) {
            inliers_ = result.inliers / (float)pyr_img1[level].area();
            error_ = result.inliers != 0 ? result.residual / result.inliers : std::numeric_limits<float>::infinity();
          }
        }
      }
      int level = 0;
      DeviceImage<float> warped(pyr_img1[level].width(), pyr_img1[level].height());   // vc::Image2DManaged<float, TargetDeviceCUDA>
      auto wr = se3aligner_.Warp(pose_ck_, camera_pyr_[level], kf_->pyr_img.GetGpuLevel(level), pyr_img1[level], kf_->pyr_dpt.GetGpuLevel(level), warped);
      const float dt = (pose_ck_.translation() - t_gt).norm();
      const float dq = (pose_ck_.so3().log() - w_gt).norm();
      std::printf("shim tracker: err=%.3e inliers=%.3f dt=%.3e dw=%.3e warp_inliers=%zu\n", error_, inliers_, dt, dq, wr.inliers);
      REQUIRE(error_ < 1e-4f && dt < 2e-3f && dq < 1e-3f && inliers_ > 0.9f && wr.inliers > (std::size_t)(0.8 * W * H));

      // error behaviour: a size mismatch must throw (the reference aborts / throws vc::CUDAException)
      ImgView bad(fr_->pyr_img.lv[0]->ptr(), W - 1, H, (std::size_t)W * 4);
      bool threw = false;
      try { se3aligner_.RunStep(pose_ck_, cam_, kf_->pyr_img.GetGpuLevel(0), bad, kf_->pyr_dpt.GetGpuLevel(0), pyr_grad1[0]); } catch (const dfx::Error&) { threw = true; }
      REQUIRE(threw);
    }
  } catch (const std::exception& e) {
    std::printf("exception: %s\n", e.what());
    return 1;
  }
  std::printf("shim_test OK\n");
  std::fflush(stdout);   // the verdict reaches the pipe before the process tears the GPU runtime down
  return 0;
}
