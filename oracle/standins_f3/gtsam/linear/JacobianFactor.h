// Stand-in (see gtsam/nonlinear/NonlinearFactor.h): VerticalBlockMatrix = column blocks of one dense row-major double matrix,
// JacobianFactor = (keys, Ab) carrier.  `Ab(b).template block<1,N>(i,0) = row; Ab(b).block<1,N>(i,0).setZero(); Ab(b)(i,0) = x;`
#pragma once
#include <vector>

#include <Eigen/Core>
#include <gtsam/nonlinear/NonlinearFactor.h>

namespace gtsam {

class VerticalBlockMatrix {
 public:
  class BlockView {
   public:
    BlockView(VerticalBlockMatrix& m, int col0) : m_(m), col0_(col0) {}
    template <int R, int C> struct Sub {
      BlockView& b; int i, j;
      template <typename O> Sub& operator=(const Eigen::MatrixBase<O>& o) {
        static_assert((int)Eigen::traits<O>::Rows == R && (int)Eigen::traits<O>::Cols == C, "block size differs");
        for (int r = 0; r < R; ++r) for (int c = 0; c < C; ++c) b(i + r, j + c) = static_cast<double>(o(r, c));
        return *this;
      }
      Sub& setZero() { for (int r = 0; r < R; ++r) for (int c = 0; c < C; ++c) b(i + r, j + c) = 0.0; return *this; }
    };
    template <int R, int C> Sub<R, C> block(int i, int j) { return Sub<R, C>{ *this, i, j }; }
    double& operator()(int i, int j) { return m_.data_[(size_t)i * m_.cols_ + col0_ + j]; }
   private:
    VerticalBlockMatrix& m_; int col0_;
  };
  template <typename Dims> VerticalBlockMatrix(const Dims& dims, size_t rows) : rows_((int)rows) {
    int c = 0;
    for (auto d : dims) { starts_.push_back(c); c += (int)d; }
    cols_ = c;
    data_.assign((size_t)rows_ * cols_, 0.0);
  }
  BlockView operator()(int b) { return BlockView(*this, starts_[(size_t)b]); }
  int rows() const { return rows_; }
  int cols() const { return cols_; }
  const std::vector<double>& data() const { return data_; }   // row-major [rows][cols]
 private:
  int rows_, cols_;
  std::vector<int> starts_;
  std::vector<double> data_;
};

class JacobianFactor : public GaussianFactor {
 public:
  JacobianFactor() : Ab_(std::vector<int>{ 1 }, 0) {}
  JacobianFactor(const std::vector<Key>& keys, const VerticalBlockMatrix& Ab) : keys_(keys), Ab_(Ab) {}
  const VerticalBlockMatrix& matrixObject() const { return Ab_; }
  const std::vector<Key>& keys() const { return keys_; }
 private:
  std::vector<Key> keys_;
  VerticalBlockMatrix Ab_;
};

}  // namespace gtsam
