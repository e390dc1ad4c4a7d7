// shim_test.cpp -- exercises include/dfx_shim.hpp (the C++17 host mirror of the reference's operator interface) end to end
// on a GPU: the coarse Gauss-Newton loop of CameraTracker::TrackFrame (reference sources/core/system/camera_tracker.cpp:42-71)
// written against df::SE3Aligner<float> exactly as the reference writes it, on a synthetic pair with a known motion.
// Exit code 0 = pass.  Built by tests/cpp/Makefile (hipcc), run by tests/test_gpu_cpp_shim.py.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <vector>

#include "../../include/dfx_shim.hpp"

using dfx::pod::Grad2;
using dfx::pod::Image2DView;
using dfx::pod::PinholeCamera;
using dfx::pod::SE3f;

#define HIPOK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 2; } } while (0)

static double tex(double u, double v) {
  return 0.5 + 0.2 * std::sin(0.081 * u + 0.047 * v) + 0.15 * std::sin(0.033 * u - 0.112 * v + 1.0) + 0.1 * std::sin(0.15 * u + 0.09 * v + 2.0);
}

static void so3_exp(const double* w, double* R) {
  const double th = std::sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
  const double A = th < 1e-9 ? 1.0 : std::sin(th) / th, B = th < 1e-9 ? 0.5 : (1 - std::cos(th)) / (th * th);
  const double K[9] = { 0, -w[2], w[1], w[2], 0, -w[0], -w[1], w[0], 0 };
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) {
    double k2 = 0; for (int k = 0; k < 3; ++k) k2 += K[i * 3 + k] * K[k * 3 + j];
    R[i * 3 + j] = (i == j) + A * K[i * 3 + j] + B * k2;
  }
}
static void R_to_q(const double* R, float* q) {   // trace > 0 for the small rotations used here
  const double s = std::sqrt(R[0] + R[4] + R[8] + 1.0) * 2;
  q[3] = (float)(0.25 * s); q[0] = (float)((R[7] - R[5]) / s); q[1] = (float)((R[2] - R[6]) / s); q[2] = (float)((R[3] - R[1]) / s);
}
static void q_to_R(const float* q, double* R) {
  const double x = q[0], y = q[1], z = q[2], w = q[3];
  R[0] = 1 - 2 * (y * y + z * z); R[1] = 2 * (x * y - z * w); R[2] = 2 * (x * z + y * w);
  R[3] = 2 * (x * y + z * w); R[4] = 1 - 2 * (x * x + z * z); R[5] = 2 * (y * z - x * w);
  R[6] = 2 * (x * z - y * w); R[7] = 2 * (y * z + x * w); R[8] = 1 - 2 * (x * x + y * y);
}

// SE3SolveAndUpdate (lucas_kanade_se3.h:85-95): update = -LDLT(JtJ)^-1 Jtr; t += dt; R = exp(dw) R
static bool solve_update(const df::SE3Aligner<float>::ReductionItem& it, SE3f& pose) {
  double A[36], b[6], L[36] = { 0 }, D[6], y[6], x[6];
  const auto M = it.JtJ.toDenseMatrix();
  for (int i = 0; i < 36; ++i) A[i] = M[i];
  for (int i = 0; i < 6; ++i) b[i] = it.Jtr[i];
  for (int j = 0; j < 6; ++j) {
    double d = A[j * 6 + j];
    for (int p = 0; p < j; ++p) d -= L[j * 6 + p] * L[j * 6 + p] * D[p];
    if (d == 0) return false;
    D[j] = d; L[j * 6 + j] = 1;
    for (int i = j + 1; i < 6; ++i) {
      double s = A[i * 6 + j];
      for (int p = 0; p < j; ++p) s -= L[i * 6 + p] * L[j * 6 + p] * D[p];
      L[i * 6 + j] = s / d;
    }
  }
  for (int i = 0; i < 6; ++i) { double s = b[i]; for (int p = 0; p < i; ++p) s -= L[i * 6 + p] * y[p]; y[i] = s; }
  for (int i = 5; i >= 0; --i) { double s = y[i] / D[i]; for (int p = i + 1; p < 6; ++p) s -= L[p * 6 + i] * x[p]; x[i] = s; }
  double w[3] = { -x[3], -x[4], -x[5] }, E[9], R[9], Rn[9];
  so3_exp(w, E);
  q_to_R(pose.q, R);
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) { double s = 0; for (int k = 0; k < 3; ++k) s += E[i * 3 + k] * R[k * 3 + j]; Rn[i * 3 + j] = s; }
  R_to_q(Rn, pose.q);
  for (int i = 0; i < 3; ++i) pose.t[i] -= (float)x[i];
  return true;
}

int main() {
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) { std::printf("no HIP device\n"); return 3; }
  const int W = 320, H = 240;
  const PinholeCamera cam{ 277.128f, 289.706f, 160.f, 120.f, (float)W, (float)H };
  const double w_gt[3] = { 0.01, -0.015, 0.008 }, t_gt[3] = { 0.04, -0.03, 0.02 };
  double Rg[9];
  so3_exp(w_gt, Rg);
  std::vector<float> img0(W * H), img1(W * H), dpt0(W * H);
  for (int y = 0; y < H; ++y) for (int x = 0; x < W; ++x) {
    const double d = 2.5 + 0.3 * (x - 160.0) / W - 0.3 * (y - 120.0) / H + 0.2 * std::sin(0.02 * x) * std::cos(0.03 * y);
    const double X = (x - cam.u0()) / cam.fx() * d, Y = (y - cam.v0()) / cam.fy() * d, Z = d;
    const double qx = Rg[0] * X + Rg[1] * Y + Rg[2] * Z + t_gt[0], qy = Rg[3] * X + Rg[4] * Y + Rg[5] * Z + t_gt[1], qz = Rg[6] * X + Rg[7] * Y + Rg[8] * Z + t_gt[2];
    img1[y * W + x] = (float)tex(x, y);
    img0[y * W + x] = (float)tex(cam.fx() * qx / qz + cam.u0(), cam.fy() * qy / qz + cam.v0());
    dpt0[y * W + x] = (float)d;
  }
  float *d_img0, *d_img1, *d_dpt0, *d_img2;
  Grad2* d_grad1;
  HIPOK(hipMalloc(&d_img0, W * H * 4)); HIPOK(hipMalloc(&d_img1, W * H * 4)); HIPOK(hipMalloc(&d_dpt0, W * H * 4));
  HIPOK(hipMalloc(&d_img2, W * H * 4)); HIPOK(hipMalloc(&d_grad1, W * H * 8));
  HIPOK(hipMemcpy(d_img0, img0.data(), W * H * 4, hipMemcpyHostToDevice));
  HIPOK(hipMemcpy(d_img1, img1.data(), W * H * 4, hipMemcpyHostToDevice));
  HIPOK(hipMemcpy(d_dpt0, dpt0.data(), W * H * 4, hipMemcpyHostToDevice));
  Image2DView<float> v0{ d_img0, (size_t)W * 4, (size_t)W, (size_t)H }, v1{ d_img1, (size_t)W * 4, (size_t)W, (size_t)H };
  Image2DView<float> vd{ d_dpt0, (size_t)W * 4, (size_t)W, (size_t)H }, v2{ d_img2, (size_t)W * 4, (size_t)W, (size_t)H };
  Image2DView<Grad2> vg{ d_grad1, (size_t)W * 8, (size_t)W, (size_t)H };

  try {
    df::SobelGradients(v1, vg);
    df::SE3Aligner<float> aligner;
    aligner.SetHuberDelta(0.1f);
    SE3f pose;   // identity, like CameraTracker::Reset
    float err = 0;
    for (int it = 0; it < 15; ++it) {
      auto r = aligner.RunStep(pose, cam, v0, v1, vd, vg);
      if (r.inliers == 0) { std::printf("no overlap\n"); return 1; }
      err = r.residual / r.inliers;
      if (!solve_update(r, pose)) { std::printf("singular\n"); return 1; }
    }
    double Rq[9], qg[4]; float qgf[4];
    R_to_q(Rg, qgf); for (int i = 0; i < 4; ++i) qg[i] = qgf[i];
    q_to_R(pose.q, Rq);
    const double dt = std::sqrt(std::pow(pose.t[0] - t_gt[0], 2) + std::pow(pose.t[1] - t_gt[1], 2) + std::pow(pose.t[2] - t_gt[2], 2));
    const double dq = std::sqrt(std::pow(pose.q[0] - qg[0], 2) + std::pow(pose.q[1] - qg[1], 2) + std::pow(pose.q[2] - qg[2], 2));
    auto wr = aligner.Warp(pose, cam, v0, v1, vd, v2);
    std::printf("shim tracker: err=%.3e dt=%.3e dq=%.3e warp_inliers=%zu\n", err, dt, dq, wr.inliers);
    if (!(err < 1e-4 && dt < 2e-3 && dq < 1e-3 && wr.inliers > (size_t)(0.8 * W * H))) return 1;

    // error behaviour: a size mismatch must throw (the reference aborts / throws vc::CUDAException)
    Image2DView<float> bad{ d_img1, (size_t)W * 4, (size_t)W - 1, (size_t)H };
    bool threw = false;
    try { aligner.RunStep(pose, cam, v0, bad, vd, vg); } catch (const dfx::Error&) { threw = true; }
    if (!threw) { std::printf("size mismatch did not throw\n"); return 1; }
  } catch (const std::exception& e) {
    std::printf("exception: %s\n", e.what());
    return 1;
  }
  std::printf("shim_test OK\n");
  return 0;
}
