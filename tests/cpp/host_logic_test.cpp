// host_logic_test.cpp -- the parts of include/dfx_host.hpp / dfx_shim.hpp that are pure host logic, run WITHOUT a GPU (plain g++; the
// library is linked but no device entry point is called): the relinearisation test of GetJacobiansIfNeeded (photometric_factor.cpp:296-306:
// relinearise when pose0, pose1 or code0 moved by >= 1e-6 in its tangent space, gtsam_traits.h:66-72), the residual rescaling of
// RunAlignmentStep (:275-282), the G11..G33 / g1..g3 / f slicing of linearize (:105-161), the packed upper-triangular item accessors
// (reduction_items.h:77-143) and the item layout helpers of include/dfx.h.  Exit code 0 = pass.
#include <cmath>
#include <cstdio>
#include <cstring>
#include <limits>
#include <vector>

#include "../../include/dfx_host.hpp"

#define REQUIRE(c) do { if (!(c)) { std::printf("FAILED %s at %s:%d\n", #c, __FILE__, __LINE__); return 1; } } while (0)

constexpr int CS = 32;
constexpr int NP = 12 + CS;

static dfx_se3 rotated(const dfx_se3& p, double ax, double ay, double az) {   // R <- exp(w) R  (left perturbation, testing_utils.h:73-88)
  const double th = std::sqrt(ax * ax + ay * ay + az * az);
  double q[4] = { 0, 0, 0, 1 };
  if (th > 0) { const double s = std::sin(th / 2) / th; q[0] = ax * s; q[1] = ay * s; q[2] = az * s; q[3] = std::cos(th / 2); }
  const double x = p.q[0], y = p.q[1], z = p.q[2], w = p.q[3];
  dfx_se3 o = p;   // o.q = q (x) p.q  (Hamilton, xyzw)
  o.q[0] = (float)(q[3] * x + q[0] * w + q[1] * z - q[2] * y);
  o.q[1] = (float)(q[3] * y - q[0] * z + q[1] * w + q[2] * x);
  o.q[2] = (float)(q[3] * z + q[0] * y - q[1] * x + q[2] * w);
  o.q[3] = (float)(q[3] * w - q[0] * x - q[1] * y - q[2] * z);
  return o;
}

int main() {
  typedef dfx::PhotometricFactor<CS> Factor;
  typedef Factor::ReductionItem Item;
  const dfx_cam cam{ 277.128f, 289.706f, 160.f, 120.f, 320.f, 240.f };
  Factor f(cam, nullptr, nullptr, 0);

  // ---- raw item -> shim item: packed row-major upper triangle, Jtr, residual, inliers (include/dfx.h helpers)
  std::vector<unsigned char> raw(dfx_item_size(NP), 0);
  float* jtj = reinterpret_cast<float*>(raw.data());
  int t = 0;
  for (int r = 0; r < NP; ++r) for (int c = r; c < NP; ++c) jtj[t++] = 1000.f * r + c;     // entry (r, c), r <= c
  REQUIRE(t == NP * (NP + 1) / 2);
  float* jtr = jtj + t;
  for (int i = 0; i < NP; ++i) jtr[i] = -1.f - i;
  jtr[NP] = 7.5f;                                                                              // residual
  REQUIRE(dfx_item_jtr(raw.data(), NP) == jtr && dfx_item_residual(raw.data(), NP) == 7.5f);
  const std::size_t ioff = (((std::size_t)(t + NP + 1)) * 4 + 7) & ~(std::size_t)7;
  REQUIRE(ioff + 8 == dfx_item_size(NP));
  const unsigned long long inl = 60000ull;
  std::memcpy(raw.data() + ioff, &inl, 8);
  REQUIRE(dfx_item_inliers(raw.data(), NP) == inl);
  const Item it = Item::FromRaw(raw.data());
  REQUIRE(it.JtJ(3, 17) == 3017.f && it.JtJ(17, 3) == 3017.f && it.JtJ(43, 43) == 43043.f && it.inliers == 60000u && it.residual == 7.5f);

  // ---- relinearisation cache
  dfx_se3 p0{ { 0, 0, 0, 1 }, { 0.1f, -0.2f, 0.3f } }, p1{ { 0.05f, -0.02f, 0.01f, 0.998f }, { 1.f, 2.f, 3.f } };
  std::array<float, CS> code{};
  for (int i = 0; i < CS; ++i) code[i] = 0.1f * i;
  REQUIRE(f.NeedsLinearization(p0, p1, code));                  // never linearised
  f.Seed(p0, p1, code, it);
  REQUIRE(!f.NeedsLinearization(p0, p1, code) && f.linearizations() == 1);
  {  // moves below / above the 1e-6 threshold: translation (exact in float around 1.0), rotation, code
    dfx_se3 q = p1; q.t[0] = std::nextafter(q.t[0], 2.f);      // 1.19e-7 < 1e-6
    REQUIRE(!f.NeedsLinearization(p0, q, code));
    q.t[0] = 1.f + 2e-6f;
    REQUIRE(f.NeedsLinearization(p0, q, code));
    REQUIRE(f.NeedsLinearization(rotated(p0, 0, 0, 5e-4), p1, code));   // float quaternions resolve ~1e-7 rad: 5e-4 rad is far above
    REQUIRE(!f.NeedsLinearization(rotated(p0, 0, 0, 0), p1, code));
    std::array<float, CS> c2 = code; c2[5] += 1e-5f;
    REQUIRE(f.NeedsLinearization(p0, p1, c2));
    c2 = code; c2[0] += 1e-7f;                                  // code[0] = 0: representable, below the threshold
    REQUIRE(!f.NeedsLinearization(p0, p1, c2));
  }

  // ---- residual rescaling (w * h / inliers) and the HessianFactor blocks
  const Item& sys = f.system();
  REQUIRE(std::fabs(sys.residual - 7.5f / 60000.f * 320.f * 240.f) < 1e-5f);
  const dfx::HessianBlocks<CS> H = f.Hessian();
  REQUIRE(H.f == (double)sys.residual);
  for (int r = 0; r < 6; ++r) for (int c = 0; c < 6; ++c) {
    REQUIRE(H.G11[r * 6 + c] == (double)it.JtJ(r, c) && H.G12[r * 6 + c] == (double)it.JtJ(r, 6 + c) && H.G22[r * 6 + c] == (double)it.JtJ(6 + r, 6 + c));
  }
  for (int r = 0; r < 6; ++r) for (int c = 0; c < CS; ++c) REQUIRE(H.G13[r * CS + c] == (double)it.JtJ(r, 12 + c) && H.G23[r * CS + c] == (double)it.JtJ(6 + r, 12 + c));
  for (int r = 0; r < CS; ++r) for (int c = 0; c < CS; ++c) REQUIRE(H.G33[r * CS + c] == (double)it.JtJ(12 + r, 12 + c));
  for (int i = 0; i < 6; ++i) REQUIRE(H.g1[i] == 1.0 + i && H.g2[i] == 7.0 + i);      // g = -Jtr
  for (int i = 0; i < CS; ++i) REQUIRE(H.g3[i] == 13.0 + i);
  {  // no overlap: inliers == 0 -> f = +inf (photometric_factor.cpp:279-282)
    Factor g(cam, nullptr, nullptr, 0);
    Item z = it; z.inliers = 0;
    g.Seed(p0, p1, code, z);
    REQUIRE(std::isinf(g.system().residual) && g.system().residual > 0);
  }
  // ---- pose distance of gtsam::traits<SE3f>::Local: (t2 - t1, log(R2 R1^T))
  REQUIRE(std::fabs(dfx::detail::pose_local_norm(p0, rotated(p0, 0.003, -0.004, 0.0)) - 0.005) < 2e-6);
  dfx_se3 pt = p0; pt.t[1] += 0.25f;
  REQUIRE(std::fabs(dfx::detail::pose_local_norm(p0, pt) - 0.25) < 1e-6);
  std::printf("host_logic_test OK\n");
  return 0;
}
