// oracle/ref_harness.cpp -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
//
// extern "C" wrappers around the UNMODIFIED reference headers under /root/reference/inc, so that
// tests and tests/golden/generate.py can run the real reference from Python (ctypes) in the build
// container.  Compiled by oracle/Makefile from the sources where they lie; outputs go only to
// oracle/_ref/ (git-ignored).  No reference source is copied into this repository.
//
// Two builds of this one file:
//   libref1d.so  (default)           : the reference's own dirac_op (inc/dirac_op.hpp), 1-D.
//   libref4d.so  (-DREF_SUBSTITUTE_OP): the reference's SBCGrQ and field primitives, unmodified,
//                 driven by a substitute class named dirac_op (the include guard
//                 LKEEGAN_BLOCKCG_DIRAC_OP_H of inc/dirac_op.hpp:1-2 is pre-defined) that applies
//                 this repository's n-D staggered operator with Eigen arithmetic.  This gives the
//                 n-D solver a reference-arithmetic oracle.
//
// Layouts at this boundary equal the reference's in-memory layouts.
#include <algorithm>
#include <cassert>
#include <chrono>
#include <complex>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "fields.hpp"

#ifdef REF_SUBSTITUTE_OP
#define LKEEGAN_BLOCKCG_DIRAC_OP_H
// Substitute operator with the reference's class name and public shape (inc/dirac_op.hpp:8-44):
// public V, mass, template op<N_rhs>(lhs, rhs) const.
class dirac_op {
 public:
  using gauge = Eigen::Matrix<std::complex<double>, N_f, N_f>;
  std::vector<gauge, Eigen::aligned_allocator<gauge>> U;  // [site][mu]
  int ndim;
  int dims[4];
  int V;
  double mass;
  dirac_op(int ndim_, const int* dims_, double mass_) : ndim(ndim_), V(1), mass(mass_) {
    for (int mu = 0; mu < 4; ++mu) {
      dims[mu] = mu < ndim ? dims_[mu] : 1;
      V *= dims[mu];
    }
    U.resize(static_cast<size_t>(V) * ndim);
  }
  int neighbour(int ix, int mu, int sign) const {
    int x[4], r = ix;
    for (int nu = 0; nu < 4; ++nu) {
      x[nu] = r % dims[nu];
      r /= dims[nu];
    }
    x[mu] = (x[mu] + sign + dims[mu]) % dims[mu];
    return ((x[3] * dims[2] + x[2]) * dims[1] + x[1]) * dims[0] + x[0];
  }
  double eta(int ix, int mu) const {
    int s = 0, r = ix;
    for (int nu = 0; nu < mu; ++nu) {
      s += r % dims[nu];
      r /= dims[nu];
    }
    return (s & 1) ? -1.0 : 1.0;
  }
  template <int N_rhs>
  void D(block_fermion_field<N_rhs>& lhs, const block_fermion_field<N_rhs>& rhs) const {
    for (int ix = 0; ix < V; ++ix) {
      block_fermion<N_rhs> acc;
      acc.setZero();
      for (int mu = 0; mu < ndim; ++mu) {
        const int xp = neighbour(ix, mu, +1), xm = neighbour(ix, mu, -1);
        acc += eta(ix, mu) * (U[ix * ndim + mu] * rhs[xp] - U[xm * ndim + mu].adjoint() * rhs[xm]);
      }
      lhs[ix] = 0.5 * acc;
    }
  }
  template <int N_rhs>
  void op(block_fermion_field<N_rhs>& lhs, const block_fermion_field<N_rhs>& rhs) const {
    block_fermion_field<N_rhs> tmp(lhs.V);
    D(tmp, rhs);
    D(lhs, tmp);
    lhs.rescale_add(-1.0, rhs, mass * mass);
  }
};
#else
#include "dirac_op.hpp"
#endif

#include "block_solvers.hpp"
#ifndef REF_SUBSTITUTE_OP
#include "standard_solvers.hpp"  // CG, SCG: compiled from src/standard_solvers.cpp where it lies (oracle/Makefile)
#endif

namespace {
using cd = std::complex<double>;

template <int N>
block_fermion_field<N> load_field(int V, const double* p) {
  block_fermion_field<N> f(V);
  for (int ix = 0; ix < V; ++ix) std::memcpy(f[ix].data(), p + static_cast<size_t>(ix) * N * N_f * 2, sizeof(cd) * N * N_f);
  return f;
}
template <int N>
void store_field(const block_fermion_field<N>& f, double* p) {
  for (int ix = 0; ix < f.V; ++ix) std::memcpy(p + static_cast<size_t>(ix) * N * N_f * 2, f[ix].data(), sizeof(cd) * N * N_f);
}
template <int N>
block_matrix<N> load_mat(const double* p) {
  block_matrix<N> m;
  std::memcpy(m.data(), p, sizeof(cd) * N * N);
  return m;
}
template <int N>
void store_mat(const block_matrix<N>& m, double* p) {
  std::memcpy(p, m.data(), sizeof(cd) * N * N);
}

// (every width instantiates the reference's templates over Eigen fixed-size matrices: the widths with fixtures only)
#define REF_DISPATCH(m, CALL)                      \
  switch (m) {                                     \
    case 1: { constexpr int N = 1; CALL; } break; \
    case 2: { constexpr int N = 2; CALL; } break; \
    case 3: { constexpr int N = 3; CALL; } break; \
    case 4: { constexpr int N = 4; CALL; } break; \
    case 5: { constexpr int N = 5; CALL; } break; \
    case 6: { constexpr int N = 6; CALL; } break; \
    case 7: { constexpr int N = 7; CALL; } break; \
    case 8: { constexpr int N = 8; CALL; } break; \
    case 12: { constexpr int N = 12; CALL; } break; \
    case 16: { constexpr int N = 16; CALL; } break; \
    case 32: { constexpr int N = 32; CALL; } break; \
    default: return -1;                            \
  }
}  // namespace

extern "C" {

void ref_srand(unsigned seed) { std::srand(seed); }

#ifdef REF_SUBSTITUTE_OP
// U: [site][mu][3x3 column-major] values supplied by the caller.
void* ref_dirac_create(int ndim, const int* dims, double mass, const double* U) {
  dirac_op* D = new dirac_op(ndim, dims, mass);
  for (size_t k = 0; k < D->U.size(); ++k) std::memcpy(D->U[k].data(), U + k * 18, sizeof(cd) * 9);
  return D;
}
#else
// The reference's gauge links are private and drawn from std::rand() in the constructor
// (inc/dirac_op.hpp:27-32).  Seeding, constructing, re-seeding and replaying V calls of
// Matrix<complex,3,3>::setRandom() recovers them without touching reference internals.
void* ref_dirac_create(int V, double mass, unsigned seed, double* U_out) {
  std::srand(seed);
  dirac_op* D = new dirac_op(V, mass);
  if (U_out) {
    std::srand(seed);
    Eigen::Matrix<cd, N_f, N_f> u;
    for (int ix = 0; ix < V; ++ix) {
      u.setRandom();
      std::memcpy(U_out + static_cast<size_t>(ix) * 18, u.data(), sizeof(cd) * 9);
    }
  }
  return D;
}
#endif
void ref_dirac_destroy(void* h) { delete static_cast<dirac_op*>(h); }
int ref_dirac_volume(void* h) { return static_cast<dirac_op*>(h)->V; }

// block_fermion_field<N>::setRandom (inc/fields.hpp:62-66), continuing the current rand() state.
int ref_field_random(int m, int V, double* out) {
  REF_DISPATCH(m, {
    block_fermion_field<N> f(V);
    f.setRandom();
    store_field(f, out);
  });
  return 0;
}

int ref_dirac_op(void* h, int m, const double* in, double* out) {
  const dirac_op& D = *static_cast<dirac_op*>(h);
  REF_DISPATCH(m, {
    block_fermion_field<N> fi = load_field<N>(D.V, in);
    block_fermion_field<N> fo(D.V);
    D.op(fo, fi);
    store_field(fo, out);
  });
  return 0;
}

int ref_add_scalar(int m, int V, double* y, const double* x, double a) {
  REF_DISPATCH(m, {
    auto fy = load_field<N>(V, y);
    auto fx = load_field<N>(V, x);
    fy.add(fx, a);
    store_field(fy, y);
  });
  return 0;
}
int ref_rescale_add_scalar(int m, int V, double* y, double a, const double* x, double b) {
  REF_DISPATCH(m, {
    auto fy = load_field<N>(V, y);
    auto fx = load_field<N>(V, x);
    fy.rescale_add(a, fx, b);
    store_field(fy, y);
  });
  return 0;
}
int ref_add_matrix(int m, int V, double* y, const double* x, const double* Mx) {
  REF_DISPATCH(m, {
    auto fy = load_field<N>(V, y);
    auto fx = load_field<N>(V, x);
    fy.add(fx, load_mat<N>(Mx));
    store_field(fy, y);
  });
  return 0;
}
int ref_rescale_add_matrix(int m, int V, double* y, const double* Mx, const double* x, double b) {
  REF_DISPATCH(m, {
    auto fy = load_field<N>(V, y);
    auto fx = load_field<N>(V, x);
    fy.rescale_add(load_mat<N>(Mx), fx, b);
    store_field(fy, y);
  });
  return 0;
}
int ref_hermitian_dot(int m, int V, const double* a, const double* b, double* out) {
  REF_DISPATCH(m, {
    auto fa = load_field<N>(V, a);
    auto fb = load_field<N>(V, b);
    store_mat<N>(fa.hermitian_dot(fb), out);
  });
  return 0;
}
int ref_tri_solve_rhs(int m, int V, double* y, const double* R) {
  REF_DISPATCH(m, {
    auto fy = load_field<N>(V, y);
    fy.multiply_upper_triangular_inverse_RHS(load_mat<N>(R));
    store_field(fy, y);
  });
  return 0;
}
int ref_thin_qr(int m, int V, double* y, double* R_out) {
  REF_DISPATCH(m, {
    auto fy = load_field<N>(V, y);
    block_matrix<N> R;
    fy.thinQR(R);
    store_field(fy, y);
    store_mat<N>(R, R_out);
  });
  return 0;
}
int ref_sub(int m, int V, double* y, const double* x) {
  REF_DISPATCH(m, {
    auto fy = load_field<N>(V, y);
    auto fx = load_field<N>(V, x);
    fy -= fx;
    store_field(fy, y);
  });
  return 0;
}
// The two Eigen factorizations SBCGrQ uses (inc/fields.hpp:142, inc/block_solvers.hpp:142).
int ref_cholesky_upper(int m, const double* G, double* R) {
  REF_DISPATCH(m, {
    block_matrix<N> g = load_mat<N>(G);
    block_matrix<N> r = g.llt().matrixL().adjoint();
    store_mat<N>(r, R);
  });
  return 0;
}
int ref_inverse(int m, const double* A, double* Ainv) {
  REF_DISPATCH(m, {
    block_matrix<N> a = load_mat<N>(A);
    block_matrix<N> r = a.fullPivLu().solve(block_matrix<N>::Identity());
    store_mat<N>(r, Ainv);
  });
  return 0;
}

// The unmodified SBCGrQ (inc/block_solvers.hpp:91-185).
int ref_sbcgrq(void* h, int m, const double* B, int nshift, const double* sigma, double eps, double eps_shifts,
               int max_iterations, double* X_out, int* iters_out, double* seconds_out) {
  const dirac_op& D = *static_cast<dirac_op*>(h);
  std::vector<double> sig(sigma, sigma + nshift);
  REF_DISPATCH(m, {
    block_fermion_field<N> fB = load_field<N>(D.V, B);
    std::vector<block_fermion_field<N>> X(nshift, fB);
    const auto t0 = std::chrono::steady_clock::now();
    const int it = SBCGrQ<N>(X, fB, D, sig, eps, eps_shifts, max_iterations);
    const auto t1 = std::chrono::steady_clock::now();
    if (seconds_out) *seconds_out = std::chrono::duration<double>(t1 - t0).count();
    if (iters_out) *iters_out = it;
    if (X_out)
      for (int s = 0; s < nshift; ++s) store_field(X[s], X_out + static_cast<size_t>(s) * D.V * N * N_f * 2);
  });
  return 0;
}

// The unmodified BCG / BCGrQ (inc/block_solvers.hpp:10-45, 50-86).
int ref_bcg(void* h, int m, const double* B, double eps, int max_iterations, int with_qr, double* X_out, int* iters_out) {
  const dirac_op& D = *static_cast<dirac_op*>(h);
  REF_DISPATCH(m, {
    block_fermion_field<N> fB = load_field<N>(D.V, B);
    block_fermion_field<N> X(D.V);
    const int it = with_qr ? BCGrQ<N>(X, fB, D, eps, max_iterations) : BCG<N>(X, fB, D, eps, max_iterations);
    if (iters_out) *iters_out = it;
    store_field(X, X_out);
  });
  return 0;
}

#ifndef REF_SUBSTITUTE_OP
// The unmodified CG / SCG (src/standard_solvers.cpp:3-32, 34-95).
int ref_cg(void* h, const double* b, double eps, int max_iterations, double* x_out, int* iters_out) {
  const dirac_op& D = *static_cast<dirac_op*>(h);
  fermion_field fb = load_field<1>(D.V, b);
  fermion_field x(D.V);
  const int it = CG(x, fb, D, eps, max_iterations);
  if (iters_out) *iters_out = it;
  store_field(x, x_out);
  return 0;
}
int ref_scg(void* h, const double* b, int nshift, const double* sigma, double eps, double eps_shifts, int max_iterations,
            double* x_out, int* iters_out) {
  const dirac_op& D = *static_cast<dirac_op*>(h);
  fermion_field fb = load_field<1>(D.V, b);
  std::vector<double> sig(sigma, sigma + nshift);
  std::vector<fermion_field> x(nshift, fb);
  const int it = SCG(x, fb, D, sig, eps, eps_shifts, max_iterations);
  if (iters_out) *iters_out = it;
  for (int s = 0; s < nshift; ++s) store_field(x[s], x_out + static_cast<size_t>(s) * D.V * N_f * 2);
  return 0;
}
#endif

// True residuals exactly as test/solvers.cpp:104-116 computes them. res_out[nshift][m].
int ref_true_residuals(void* h, int m, const double* B, int nshift, const double* sigma, const double* X, double* res_out) {
  const dirac_op& D = *static_cast<dirac_op*>(h);
  REF_DISPATCH(m, {
    block_fermion_field<N> fB = load_field<N>(D.V, B);
    block_fermion_field<N> AX(D.V);
    block_matrix<N> b2 = fB.hermitian_dot(fB);
    for (int s = 0; s < nshift; ++s) {
      auto fX = load_field<N>(D.V, X + static_cast<size_t>(s) * D.V * N * N_f * 2);
      D.op(AX, fX);
      AX.add(fX, sigma[s]);
      AX -= fB;
      block_matrix<N> r2 = AX.hermitian_dot(AX);
      for (int i = 0; i < N; ++i) res_out[s * N + i] = sqrt(r2(i, i).real() / b2(i, i).real());
    }
  });
  return 0;
}

}  // extern "C"
