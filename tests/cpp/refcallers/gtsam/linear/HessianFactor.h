// Stand-in for gtsam::HessianFactor as photometric_factor.cpp:180 builds it: (keys, {G11, G12, G13, G22, G23, G33}, {g1, g2, g3}, f).
// TEST INFRASTRUCTURE: keeps what it is given so that the test can compare it with the C ABI's item.
#pragma once
#include <gtsam/nonlinear/NonlinearFactor.h>

namespace gtsam {
class HessianFactor : public GaussianFactor {
 public:
  HessianFactor(const std::vector<Key>& keys, const std::vector<Matrix>& Gs, const std::vector<Vector>& gs, double f) : keys_(keys), Gs_(Gs), gs_(gs), f_(f) {}
  const std::vector<Key>& keys() const { return keys_; }
  const std::vector<Matrix>& Gs() const { return Gs_; }
  const std::vector<Vector>& gs() const { return gs_; }
  double constantTerm() const { return f_; }
 private:
  std::vector<Key> keys_;
  std::vector<Matrix> Gs_;
  std::vector<Vector> gs_;
  double f_;
};
}  // namespace gtsam
