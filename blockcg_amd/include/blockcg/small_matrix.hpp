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
class cmatrix;

// What A.fullPivLu() returns (Eigen's FullPivLU, complete pivoting): the two members the reference's solver templates use,
// .solve(B) (inc/block_solvers.hpp:31,36,73,142,166) and .inverse().  Own code; the elimination is the one the library
// itself runs for alpha and beta_s (csrc/small_matrix.hpp: pivot = the entry of largest modulus of the remaining block,
// found in column-major order), so a solver written against these headers in the reference's style gets the coefficients
// the library's own SBCGrQ computes.
template <int N>
class full_piv_lu {
 public:
  explicit full_piv_lu(const cmatrix<N, N>& A);
  template <int C>
  cmatrix<N, C> solve(const cmatrix<N, C>& B) const;
  cmatrix<N, N> inverse() const;

 private:
  cplx w_[N * N];  // L (unit diagonal, below) and U (on and above the diagonal) of the row- and column-permuted matrix
  int rp_[N], cp_[N];
  cplx& w(int i, int j) { return w_[j * N + i]; }
  const cplx& w(int i, int j) const { return w_[j * N + i]; }
};

// What A.llt() returns (Eigen's LLT): .matrixL() is the lower-triangular Cholesky factor, A = L L^dagger, and the
// reference takes .matrixL().adjoint() as the R of thinQR (inc/fields.hpp:142).  A non-positive pivot gives NaN, as in Eigen.
template <int N>
class llt_of {
 public:
  explicit llt_of(const cmatrix<N, N>& A);
  cmatrix<N, N> matrixL() const;
  cmatrix<N, N> matrixU() const;

 private:
  cplx l_[N * N];
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
  // Eigen's decompositions as the reference's templates call them (square matrices)
  full_piv_lu<Rows> fullPivLu() const {
    static_assert(Rows == Cols, "fullPivLu: square matrices");
    return full_piv_lu<Rows>(*this);
  }
  llt_of<Rows> llt() const {
    static_assert(Rows == Cols, "llt: square matrices");
    return llt_of<Rows>(*this);
  }
  cmatrix inverse() const { return fullPivLu().inverse(); }
  double norm() const {  // Frobenius norm (a column's 2-norm)
    double s2 = 0.0;
    for (int k = 0; k < Rows * Cols; ++k) s2 += std::norm(v[k]);
    return std::sqrt(s2);
  }
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

template <int R, int C>
inline cmatrix<R, C> operator*(cmatrix<R, C> a, double s) { return s * a; }
template <int R, int C>
inline cmatrix<R, C> operator*(cplx s, cmatrix<R, C> a) {
  for (int k = 0; k < R * C; ++k) a.v[k] *= s;
  return a;
}

template <int N>
full_piv_lu<N>::full_piv_lu(const cmatrix<N, N>& A) {
  for (int k = 0; k < N * N; ++k) w_[k] = A.v[k];
  for (int i = 0; i < N; ++i) rp_[i] = cp_[i] = i;
  for (int k = 0; k < N; ++k) {
    int bi = k, bj = k;
    double best = -1.0;
    for (int j = k; j < N; ++j)
      for (int i = k; i < N; ++i) {
        const double a = std::norm(w(i, j));
        if (a > best) {
          best = a;
          bi = i;
          bj = j;
        }
      }
    if (bi != k) {
      for (int j = 0; j < N; ++j) {
        const cplx t = w(k, j);
        w(k, j) = w(bi, j);
        w(bi, j) = t;
      }
      const int t = rp_[k];
      rp_[k] = rp_[bi];
      rp_[bi] = t;
    }
    if (bj != k) {
      for (int i = 0; i < N; ++i) {
        const cplx t = w(i, k);
        w(i, k) = w(i, bj);
        w(i, bj) = t;
      }
      const int t = cp_[k];
      cp_[k] = cp_[bj];
      cp_[bj] = t;
    }
    const cplx inv_p = cplx(1.0) / w(k, k);
    for (int i = k + 1; i < N; ++i) w(i, k) *= inv_p;
    for (int j = k + 1; j < N; ++j) {
      const cplx u = w(k, j);
      for (int i = k + 1; i < N; ++i) w(i, j) -= w(i, k) * u;
    }
  }
}
template <int N>
template <int C>
cmatrix<N, C> full_piv_lu<N>::solve(const cmatrix<N, C>& B) const {
  cmatrix<N, C> X;
  cplx y[N];
  for (int c = 0; c < C; ++c) {
    for (int i = 0; i < N; ++i) y[i] = B(rp_[i], c);
    for (int i = 1; i < N; ++i)
      for (int p = 0; p < i; ++p) y[i] -= w(i, p) * y[p];
    for (int i = N - 1; i >= 0; --i) {
      for (int p = i + 1; p < N; ++p) y[i] -= w(i, p) * y[p];
      y[i] /= w(i, i);
    }
    for (int i = 0; i < N; ++i) X(cp_[i], c) = y[i];
  }
  return X;
}
template <int N>
cmatrix<N, N> full_piv_lu<N>::inverse() const {
  return solve(cmatrix<N, N>::Identity());
}

template <int N>
llt_of<N>::llt_of(const cmatrix<N, N>& A) {
  for (int k = 0; k < N * N; ++k) l_[k] = cplx(0.0, 0.0);
  for (int j = 0; j < N; ++j) {  // column j of L: L(j,j), then L(i > j, j)
    double d = A(j, j).real();
    for (int p = 0; p < j; ++p) d -= std::norm(l_[p * N + j]);
    const double ljj = std::sqrt(d);
    l_[j * N + j] = ljj;
    for (int i = j + 1; i < N; ++i) {
      cplx sacc = A(i, j);
      for (int p = 0; p < j; ++p) sacc -= l_[p * N + i] * std::conj(l_[p * N + j]);
      l_[j * N + i] = sacc / ljj;
    }
  }
}
template <int N>
cmatrix<N, N> llt_of<N>::matrixL() const {
  cmatrix<N, N> L;
  for (int k = 0; k < N * N; ++k) L.v[k] = l_[k];
  return L;
}
template <int N>
cmatrix<N, N> llt_of<N>::matrixU() const {
  return matrixL().adjoint();
}

}  // namespace blockcg
#endif
