// oracle/small_matrix.hpp -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
//
// Dense m x m complex<double> algebra used by the CPU oracle (oracle/oracle.hpp).
// The reference keeps this work in Eigen (vendored only under /root/reference, so it
// cannot travel); this file restates the few Eigen calls SBCGrQ makes:
//   .llt().matrixL().adjoint()   inc/fields.hpp:142          -> cholesky_upper()
//   .fullPivLu().solve(Identity) inc/block_solvers.hpp:142,166 -> inverse_full_piv_lu()
//   operator*, .adjoint()        inc/block_solvers.hpp:145,153,158,163-177
//   .rowwise().norm()            inc/block_solvers.hpp:130,155,169-172
// Storage is column-major like Eigen's default (inc/Eigen3/Eigen/src/Core/util/Macros.h:334-337):
// element (i,j) at a[j*m+i].
#ifndef BLOCKCG_ORACLE_SMALL_MATRIX_HPP
#define BLOCKCG_ORACLE_SMALL_MATRIX_HPP
#include <cmath>
#include <complex>
#include <vector>

namespace oracle {

using cplx = std::complex<double>;

struct Mat {
  int m;
  std::vector<cplx> a;
  explicit Mat(int m_ = 0) : m(m_), a(static_cast<size_t>(m_) * m_, cplx(0, 0)) {}
  cplx& operator()(int i, int j) { return a[static_cast<size_t>(j) * m + i]; }
  const cplx& operator()(int i, int j) const { return a[static_cast<size_t>(j) * m + i]; }
  static Mat Identity(int m) {
    Mat r(m);
    for (int i = 0; i < m; ++i) r(i, i) = cplx(1, 0);
    return r;
  }
  Mat adjoint() const {
    Mat r(m);
    for (int i = 0; i < m; ++i)
      for (int j = 0; j < m; ++j) r(j, i) = std::conj((*this)(i, j));
    return r;
  }
};

inline Mat operator*(const Mat& x, const Mat& y) {
  const int m = x.m;
  Mat r(m);
  for (int j = 0; j < m; ++j)
    for (int k = 0; k < m; ++k) {
      const cplx ykj = y(k, j);
      for (int i = 0; i < m; ++i) r(i, j) += x(i, k) * ykj;
    }
  return r;
}
inline Mat operator*(double s, const Mat& x) {
  Mat r(x);
  for (auto& v : r.a) v *= s;
  return r;
}
inline Mat operator+(const Mat& x, const Mat& y) {
  Mat r(x);
  for (size_t i = 0; i < r.a.size(); ++i) r.a[i] += y.a[i];
  return r;
}
inline Mat operator-(const Mat& x, const Mat& y) {
  Mat r(x);
  for (size_t i = 0; i < r.a.size(); ++i) r.a[i] -= y.a[i];
  return r;
}
inline Mat operator-(const Mat& x) {
  Mat r(x);
  for (auto& v : r.a) v = -v;
  return r;
}

// R = chol(G)^dagger, upper triangular, G = R^dagger R.  Restates
// G.llt().matrixL().adjoint() (inc/fields.hpp:142).  Like Eigen's LLT it does not stop on a
// non-positive pivot: sqrt of a negative number yields NaN and the caller sees NaN.
inline Mat cholesky_upper(const Mat& G) {
  const int m = G.m;
  Mat L(m);
  for (int k = 0; k < m; ++k) {
    double x = G(k, k).real();
    for (int p = 0; p < k; ++p) x -= std::norm(L(k, p));
    const double lkk = std::sqrt(x);
    L(k, k) = cplx(lkk, 0.0);
    for (int i = k + 1; i < m; ++i) {
      cplx s = G(i, k);
      for (int p = 0; p < k; ++p) s -= L(i, p) * std::conj(L(k, p));
      L(i, k) = s / lkk;
    }
  }
  return L.adjoint();
}

// A^{-1} by LU with full (row and column) pivoting; restates
// A.fullPivLu().solve(Identity) (inc/block_solvers.hpp:142,166).
inline Mat inverse_full_piv_lu(const Mat& A) {
  const int m = A.m;
  Mat lu(A);
  std::vector<int> rowp(m), colp(m);
  for (int i = 0; i < m; ++i) rowp[i] = colp[i] = i;
  for (int k = 0; k < m; ++k) {
    int pi = k, pj = k;
    double best = -1.0;
    for (int j = k; j < m; ++j)
      for (int i = k; i < m; ++i) {
        const double v = std::norm(lu(i, j));
        if (v > best) { best = v; pi = i; pj = j; }
      }
    if (pi != k) {
      for (int j = 0; j < m; ++j) std::swap(lu(k, j), lu(pi, j));
      std::swap(rowp[k], rowp[pi]);
    }
    if (pj != k) {
      for (int i = 0; i < m; ++i) std::swap(lu(i, k), lu(i, pj));
      std::swap(colp[k], colp[pj]);
    }
    const cplx piv = lu(k, k);
    for (int i = k + 1; i < m; ++i) lu(i, k) /= piv;
    for (int j = k + 1; j < m; ++j) {
      const cplx ukj = lu(k, j);
      for (int i = k + 1; i < m; ++i) lu(i, j) -= lu(i, k) * ukj;
    }
  }
  // Solve P A Q = L U  =>  A^{-1} = Q U^{-1} L^{-1} P, column by column of the identity.
  Mat inv(m);
  std::vector<cplx> y(m);
  for (int c = 0; c < m; ++c) {
    for (int i = 0; i < m; ++i) y[i] = (rowp[i] == c) ? cplx(1, 0) : cplx(0, 0);
    for (int i = 0; i < m; ++i)
      for (int p = 0; p < i; ++p) y[i] -= lu(i, p) * y[p];
    for (int i = m - 1; i >= 0; --i) {
      for (int p = i + 1; p < m; ++p) y[i] -= lu(i, p) * y[p];
      y[i] /= lu(i, i);
    }
    for (int i = 0; i < m; ++i) inv(colp[i], c) = y[i];
  }
  return inv;
}

// ||row_i(A)||_2 for every i; restates A.rowwise().norm() (inc/block_solvers.hpp:130).
inline std::vector<double> rowwise_norm(const Mat& A) {
  std::vector<double> r(A.m, 0.0);
  for (int i = 0; i < A.m; ++i) {
    double s = 0.0;
    for (int j = 0; j < A.m; ++j) s += std::norm(A(i, j));
    r[i] = std::sqrt(s);
  }
  return r;
}

}  // namespace oracle
#endif
