// Host-side dense m x m complex<double> algebra for the SBCGrQ coefficient updates.
//
// The reference does this work with Eigen fixed-size matrices (inc/block_solvers.hpp:142,153,155,
// 163-172; inc/fields.hpp:142).  Eigen is vendored only inside the reference tree, so the product
// carries its own few routines; m <= 32, cost is microseconds per iteration.
// Column-major storage, element (i,j) at j*m+i, the layout of block_matrix<N_rhs>
// (inc/fields.hpp:22-23).
#pragma once
#include <cmath>
#include <complex>
#include <utility>
#include <vector>

namespace bcg {

using cd = std::complex<double>;

class CMat {
 public:
  CMat() : m_(0) {}
  explicit CMat(int m) : m_(m), v_(static_cast<size_t>(m) * m) {}
  CMat(int m, const double* interleaved) : m_(m), v_(static_cast<size_t>(m) * m) {
    for (size_t k = 0; k < v_.size(); ++k) v_[k] = cd(interleaved[2 * k], interleaved[2 * k + 1]);
  }
  static CMat identity(int m) {
    CMat r(m);
    for (int i = 0; i < m; ++i) r(i, i) = 1.0;
    return r;
  }
  int dim() const { return m_; }
  cd& operator()(int i, int j) { return v_[static_cast<size_t>(j) * m_ + i]; }
  const cd& operator()(int i, int j) const { return v_[static_cast<size_t>(j) * m_ + i]; }
  const cd* data() const { return v_.data(); }
  cd* data() { return v_.data(); }
  void store(double* interleaved) const {
    for (size_t k = 0; k < v_.size(); ++k) {
      interleaved[2 * k] = v_[k].real();
      interleaved[2 * k + 1] = v_[k].imag();
    }
  }
  bool all_finite() const {
    for (const cd& z : v_)
      if (!std::isfinite(z.real()) || !std::isfinite(z.imag())) return false;
    return true;
  }
  CMat adjoint() const {
    CMat r(m_);
    for (int j = 0; j < m_; ++j)
      for (int i = 0; i < m_; ++i) r(j, i) = std::conj((*this)(i, j));
    return r;
  }
  CMat& operator+=(const CMat& o) {
    for (size_t k = 0; k < v_.size(); ++k) v_[k] += o.v_[k];
    return *this;
  }
  CMat& operator-=(const CMat& o) {
    for (size_t k = 0; k < v_.size(); ++k) v_[k] -= o.v_[k];
    return *this;
  }
  CMat& operator*=(double s) {
    for (cd& z : v_) z *= s;
    return *this;
  }
  CMat operator-() const {
    CMat r(*this);
    for (cd& z : r.v_) z = -z;
    return r;
  }
  // max_i ||row_i|| / denom_i   (delta.rowwise().norm().array() / b_norm).maxCoeff()
  // (inc/block_solvers.hpp:155,169-172)
  std::vector<double> row_norms() const {
    std::vector<double> r(m_);
    for (int i = 0; i < m_; ++i) {
      double s = 0;
      for (int j = 0; j < m_; ++j) s += std::norm((*this)(i, j));
      r[i] = std::sqrt(s);
    }
    return r;
  }

 private:
  int m_;
  std::vector<cd> v_;
};

inline CMat operator+(CMat a, const CMat& b) { return a += b; }
inline CMat operator-(CMat a, const CMat& b) { return a -= b; }
inline CMat operator*(double s, CMat a) { return a *= s; }

inline CMat operator*(const CMat& a, const CMat& b) {
  const int m = a.dim();
  CMat c(m);
  for (int j = 0; j < m; ++j)
    for (int k = 0; k < m; ++k) {
      const cd bkj = b(k, j);
      const double br = bkj.real(), bi = bkj.imag();
      for (int i = 0; i < m; ++i) {
        const cd aik = a(i, k);
        c(i, j) += cd(aik.real() * br - aik.imag() * bi, aik.real() * bi + aik.imag() * br);
      }
    }
  return c;
}

// Upper-triangular R with G = R^dagger R (G Hermitian positive definite): the reference's
// G.llt().matrixL().adjoint() (inc/fields.hpp:142).  Returns false when a pivot is not positive
// and finite (the reference would carry NaN forward).
inline bool cholesky_upper(const CMat& G, CMat& R) {
  const int m = G.dim();
  R = CMat(m);
  bool ok = true;
  for (int j = 0; j < m; ++j) {
    // row j of R: R(j,j) then R(j, i>j)
    double d = G(j, j).real();
    for (int p = 0; p < j; ++p) d -= std::norm(R(p, j));
    if (!(d > 0.0) || !std::isfinite(d)) ok = false;
    const double rjj = std::sqrt(d);
    R(j, j) = rjj;
    for (int i = j + 1; i < m; ++i) {
      cd s = G(j, i);  // = conj(G(i,j))
      for (int p = 0; p < j; ++p) s -= std::conj(R(p, j)) * R(p, i);
      R(j, i) = s / rjj;
    }
  }
  return ok;
}

// A^{-1} via Gaussian elimination with full pivoting: A.fullPivLu().solve(Identity)
// (inc/block_solvers.hpp:142,166).
inline CMat inverse_full_pivot(const CMat& A) {
  const int m = A.dim();
  CMat w(A);
  std::vector<int> rp(m), cp(m);
  for (int i = 0; i < m; ++i) rp[i] = cp[i] = i;
  for (int k = 0; k < m; ++k) {
    int bi = k, bj = k;
    double best = -1;
    for (int j = k; j < m; ++j)
      for (int i = k; i < m; ++i) {
        const double a = std::norm(w(i, j));
        if (a > best) { best = a; bi = i; bj = j; }
      }
    if (bi != k) {
      for (int j = 0; j < m; ++j) std::swap(w(k, j), w(bi, j));
      std::swap(rp[k], rp[bi]);
    }
    if (bj != k) {
      for (int i = 0; i < m; ++i) std::swap(w(i, k), w(i, bj));
      std::swap(cp[k], cp[bj]);
    }
    const cd inv_p = cd(1.0) / w(k, k);
    for (int i = k + 1; i < m; ++i) w(i, k) *= inv_p;
    for (int j = k + 1; j < m; ++j) {
      const cd u = w(k, j);
      for (int i = k + 1; i < m; ++i) w(i, j) -= w(i, k) * u;
    }
  }
  CMat inv(m);
  std::vector<cd> y(m);
  for (int c = 0; c < m; ++c) {
    for (int i = 0; i < m; ++i) y[i] = (rp[i] == c) ? cd(1.0) : cd(0.0);
    for (int i = 1; i < m; ++i)
      for (int p = 0; p < i; ++p) y[i] -= w(i, p) * y[p];
    for (int i = m - 1; i >= 0; --i) {
      for (int p = i + 1; p < m; ++p) y[i] -= w(i, p) * y[p];
      y[i] /= w(i, i);
    }
    for (int i = 0; i < m; ++i) inv(cp[i], c) = y[i];
  }
  return inv;
}

// Inverse of an upper-triangular matrix by back substitution in extended precision, rounded once
// to double.  Used by the MFMA fast path, which applies Q <- Q R^{-1} as a dense product instead
// of the reference's column-by-column substitution (inc/fields.hpp:125-136).
inline CMat upper_triangular_inverse(const CMat& R) {
  const int m = R.dim();
  using cl = std::complex<long double>;
  std::vector<cl> x(static_cast<size_t>(m) * m, cl(0));
  auto X = [&](int i, int j) -> cl& { return x[static_cast<size_t>(j) * m + i]; };
  for (int j = 0; j < m; ++j) {
    X(j, j) = cl(1) / cl(R(j, j));
    for (int i = j - 1; i >= 0; --i) {
      cl s = 0;
      for (int p = i + 1; p <= j; ++p) s += cl(R(i, p)) * X(p, j);
      X(i, j) = -s / cl(R(i, i));
    }
  }
  CMat out(m);
  for (int j = 0; j < m; ++j)
    for (int i = 0; i < m; ++i) out(i, j) = cd(static_cast<double>(X(i, j).real()), static_cast<double>(X(i, j).imag()));
  return out;
}

}  // namespace bcg
