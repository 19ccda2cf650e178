// blockcg/small_matrix.hpp -- fixed-size complex matrices for the header-level drop-in API.
//
// The reference defines block_fermion<N_rhs> (3 x N_rhs) and block_matrix<N_rhs> (N_rhs x N_rhs) as
// Eigen::Matrix<std::complex<double>, ...> (inc/fields.hpp:18-23).  Eigen is vendored only inside the
// reference tree, so these are small self-contained types with the same storage (column-major, element
// (i,j) at j*Rows+i) and the handful of members the reference's drivers use.
#ifndef BLOCKCG_SMALL_MATRIX_HPP
#define BLOCKCG_SMALL_MATRIX_HPP
#include <cmath>
#include <complex>
#include <cstdlib>

namespace blockcg {

using cplx = std::complex<double>;

// Eigen's setRandom for real scalars: -1 + 2*rand()/RAND_MAX
// (inc/Eigen3/Eigen/src/Core/MathFunctions.h:618-627).  For complex the two draws are constructor
// arguments of std::complex (:723-727); g++ evaluates them right to left, so the IMAGINARY part is
// drawn first.  Kept identical so that srand(k) reproduces the reference's lattices.
inline double eigen_random_real() { return -1.0 + 2.0 * static_cast<double>(std::rand()) / static_cast<double>(RAND_MAX); }
inline cplx eigen_random_complex() {
  const double im = eigen_random_real();
  const double re = eigen_random_real();
  return cplx(re, im);
}

// Coefficient-wise real array / real vector of fixed length: what Eigen's .real(), .array(), .rowwise().norm() yield in
// the reference's residual expressions (benchmark.cpp:100-101, inc/block_solvers.hpp:20,42,62,83,130,155,170-172):
//   (r2.diagonal().real().array() / b2.diagonal().array().real()).maxCoeff()
//   (delta.rowwise().norm().array() / b_norm).maxCoeff(),   r2.diagonal().real().array().sqrt()
template <int N>
struct rarray {
  double v[N];
  rarray() {
    for (int i = 0; i < N; ++i) v[i] = 0.0;
  }
  double& operator()(int i) { return v[i]; }
  double operator()(int i) const { return v[i]; }
  double& operator[](int i) { return v[i]; }
  double operator[](int i) const { return v[i]; }
  static constexpr int size() { return N; }
  const rarray& array() const { return *this; }
  const rarray& matrix() const { return *this; }
  const rarray& real() const { return *this; }
  const rarray& eval() const { return *this; }
  rarray sqrt() const {
    rarray r;
    for (int i = 0; i < N; ++i) r.v[i] = std::sqrt(v[i]);
    return r;
  }
  double maxCoeff() const {
    double m = v[0];
    for (int i = 1; i < N; ++i) m = v[i] > m ? v[i] : m;
    return m;
  }
  double minCoeff() const {
    double m = v[0];
    for (int i = 1; i < N; ++i) m = v[i] < m ? v[i] : m;
    return m;
  }
  double sum() const {
    double s = 0.0;
    for (int i = 0; i < N; ++i) s += v[i];
    return s;
  }
};
template <int N>
inline rarray<N> operator/(const rarray<N>& a, const rarray<N>& b) {
  rarray<N> r;
  for (int i = 0; i < N; ++i) r.v[i] = a.v[i] / b.v[i];
  return r;
}
template <int N>
inline rarray<N> operator*(const rarray<N>& a, const rarray<N>& b) {
  rarray<N> r;
  for (int i = 0; i < N; ++i) r.v[i] = a.v[i] * b.v[i];
  return r;
}
template <int N>
inline rarray<N> sqrt(const rarray<N>& a) { return a.sqrt(); }

// complex counterpart: diagonal() of a matrix
template <int N>
struct carray {
  cplx v[N];
  cplx& operator()(int i) { return v[i]; }
  const cplx& operator()(int i) const { return v[i]; }
  const carray& array() const { return *this; }
  const carray& matrix() const { return *this; }
  rarray<N> real() const {
    rarray<N> r;
    for (int i = 0; i < N; ++i) r.v[i] = v[i].real();
    return r;
  }
  rarray<N> imag() const {
    rarray<N> r;
    for (int i = 0; i < N; ++i) r.v[i] = v[i].imag();
    return r;
  }
};

template <int Rows, int Cols>
class cmatrix {
 public:
  cplx v[Rows * Cols];
  cmatrix() {
    for (int k = 0; k < Rows * Cols; ++k) v[k] = cplx(0.0, 0.0);
  }
  static cmatrix Zero() { return cmatrix(); }
  static cmatrix Identity() {
    cmatrix r;
    for (int i = 0; i < (Rows < Cols ? Rows : Cols); ++i) r(i, i) = cplx(1.0, 0.0);
    return r;
  }
  cplx& operator()(int i, int j) { return v[j * Rows + i]; }
  const cplx& operator()(int i, int j) const { return v[j * Rows + i]; }
  cplx* data() { return v; }
  const cplx* data() const { return v; }
  static constexpr int rows() { return Rows; }
  static constexpr int cols() { return Cols; }
  void setZero() {
    for (int k = 0; k < Rows * Cols; ++k) v[k] = cplx(0.0, 0.0);
  }
  void setRandom() {  // column-major fill order, like Eigen's
    for (int k = 0; k < Rows * Cols; ++k) v[k] = eigen_random_complex();
  }
  const cmatrix& eval() const { return *this; }
  cmatrix<Cols, Rows> adjoint() const {
    cmatrix<Cols, Rows> r;
    for (int j = 0; j < Cols; ++j)
      for (int i = 0; i < Rows; ++i) r(j, i) = std::conj((*this)(i, j));
    return r;
  }
  cmatrix<Rows, 1> col(int j) const {
    cmatrix<Rows, 1> r;
    for (int i = 0; i < Rows; ++i) r(i, 0) = (*this)(i, j);
    return r;
  }
  void set_col(int j, const cmatrix<Rows, 1>& c) {
    for (int i = 0; i < Rows; ++i) (*this)(i, j) = c(i, 0);
  }
  cmatrix& operator+=(const cmatrix& o) {
    for (int k = 0; k < Rows * Cols; ++k) v[k] += o.v[k];
    return *this;
  }
  cmatrix& operator-=(const cmatrix& o) {
    for (int k = 0; k < Rows * Cols; ++k) v[k] -= o.v[k];
    return *this;
  }
  cmatrix operator-() const {
    cmatrix r;
    for (int k = 0; k < Rows * Cols; ++k) r.v[k] = -v[k];
    return r;
  }
  carray<(Rows < Cols ? Rows : Cols)> diagonal() const {
    carray<(Rows < Cols ? Rows : Cols)> d;
    for (int i = 0; i < (Rows < Cols ? Rows : Cols); ++i) d.v[i] = (*this)(i, i);
    return d;
  }
  struct rowwise_proxy {
    const cmatrix& a;
    rarray<Rows> norm() const {
      rarray<Rows> r;
      a.rowwise_norm(r.v);
      return r;
    }
  };
  rowwise_proxy rowwise() const { return rowwise_proxy{*this}; }
  // sqrt(sum_j |a_ij|^2) for every row i (Eigen: rowwise().norm())
  void rowwise_norm(double* out) const {
    for (int i = 0; i < Rows; ++i) {
      double s = 0.0;
      for (int j = 0; j < Cols; ++j) s += std::norm((*this)(i, j));
      out[i] = std::sqrt(s);
    }
  }
};

template <int R, int C>
inline cmatrix<R, C> operator+(cmatrix<R, C> a, const cmatrix<R, C>& b) { return a += b; }
template <int R, int C>
inline cmatrix<R, C> operator-(cmatrix<R, C> a, const cmatrix<R, C>& b) { return a -= b; }
template <int R, int C>
inline cmatrix<R, C> operator*(double s, cmatrix<R, C> a) {
  for (int k = 0; k < R * C; ++k) a.v[k] *= s;
  return a;
}
template <int R, int K, int C>
inline cmatrix<R, C> operator*(const cmatrix<R, K>& a, const cmatrix<K, C>& b) {
  cmatrix<R, C> r;
  for (int j = 0; j < C; ++j)
    for (int k = 0; k < K; ++k)
      for (int i = 0; i < R; ++i) r(i, j) += a(i, k) * b(k, j);
  return r;
}

}  // namespace blockcg
#endif
