// Stand-in for vc::types::SquareUpperTriangularMatrix (jczarnowski/vision_core, absent).  TEST INFRASTRUCTURE (oracle/_ref).
// N (N + 1) / 2 coefficients, row-major over the upper triangle ((0,0), (0,1) .. (0,N-1), (1,1) ..); constructed from an
// N-vector v it holds v_r * v_c for c >= r (the packing order is observable only through our own ABI).
#pragma once
#include <Eigen/Core>

namespace vc {
namespace types {

template <typename T, int N>
class SquareUpperTriangularMatrix {
 public:
  typedef Eigen::Matrix<T, N*(N + 1) / 2, 1> CoeffType;
  typedef Eigen::Matrix<T, N, N> DenseMatrixType;
  SquareUpperTriangularMatrix() {}
  template <typename D>
  explicit SquareUpperTriangularMatrix(const Eigen::MatrixBase<D>& v) {
    int k = 0;
    for (int r = 0; r < N; ++r) for (int c = r; c < N; ++c) coeff_(k++) = v(r) * v(c);
  }
  static SquareUpperTriangularMatrix Zero() { return SquareUpperTriangularMatrix(); }
  SquareUpperTriangularMatrix& operator+=(const SquareUpperTriangularMatrix& o) { coeff_ += o.coeff_; return *this; }
  CoeffType& coeff() { return coeff_; }
  const CoeffType& coeff() const { return coeff_; }
  DenseMatrixType toDenseMatrix() const {
    DenseMatrixType M;
    int k = 0;
    for (int r = 0; r < N; ++r) for (int c = r; c < N; ++c) { M(r, c) = coeff_(k); M(c, r) = coeff_(k); ++k; }
    return M;
  }
 private:
  CoeffType coeff_;
};

}  // namespace types
}  // namespace vc
